"""Typed Python wrappers over the C ABI (include/loans_hip.h).

Device memory, streams and allocation come from PyTorch-ROCm (plumbing); all
arithmetic is in the HIP kernels.  Tensors handed to these wrappers are NHWC
float32 and contiguous unless a wrapper says otherwise.  Nothing here has a CPU
fallback: without the built library ``_lib.load()`` raises.
"""
import ctypes as C
import os
import threading
import time

import numpy as np

import torch

from . import _lib
from ._lib import (F_ADDEND, F_ADDEND_MASK, F_AFFINE_IN, F_BIAS, F_BNSUMS, F_DENSE, F_GY_BF16, F_MASK, F_OUT_BF16, F_RELU_IN, F_STATS,
                   IgemmDesc, check)

# Arithmetic of the convolution contractions: 'f32' (exact fp32 MFMA, the parity path) or 'bf16' (operands rounded
# to bf16 while staged into LDS, bf16 MFMA, fp32 accumulate; tensors stay fp32 in memory).  BASELINE configs 3 / 5.
COMPUTE = 'f32'


# Storage of the activations / gradients inside the localizer's residual stages: 'f32', or 'bf16' (needs COMPUTE ==
# 'bf16'): the stem's pool then writes bf16, every conv / BN pass of the stages reads and writes bf16 (their wrappers
# below dispatch on the tensor's dtype) and the pooled features return to fp32.  Parameters, their gradients
# ("fp32 grad accumulate"), BN statistics and the assessor stay fp32.
STORAGE = 'f32'
BF16 = torch.bfloat16


def set_compute_dtype(name):
    """the PROCESS DEFAULT: what a model without a precision of its own (``Link.set_precision``) runs in"""
    global COMPUTE, STORAGE
    if name not in ('f32', 'bf16'):
        raise ValueError("compute dtype must be 'f32' or 'bf16'")
    COMPUTE = name
    if name == 'f32':
        STORAGE = 'f32'


def set_storage_dtype(name):
    global STORAGE
    if name not in ('f32', 'bf16'):
        raise ValueError("storage dtype must be 'f32' or 'bf16'")
    if name == 'bf16' and COMPUTE != 'bf16':
        raise ValueError("bf16 storage needs set_compute_dtype('bf16') first")
    STORAGE = name


for _gone in ('LOANS_COMPUTE', 'LOANS_STORAGE'):         # removed in round 5 (the arithmetic is a property of the model): say so
    if os.environ.get(_gone):
        import warnings
        warnings.warn('%s is no longer read: use Link.set_precision / ops.precision' % _gone, stacklevel=1)


def check_precision(compute, storage):
    if compute not in ('f32', 'bf16') or storage not in ('f32', 'bf16'):
        raise ValueError("compute / storage dtype must be 'f32' or 'bf16'")
    if storage == 'bf16' and compute != 'bf16':
        raise ValueError('bf16 storage needs bf16 compute')
    return compute, storage


class precision:
    """``with ops.precision(compute, storage):`` -- the arithmetic of everything launched inside.  The dtype arm is a
    property of the MODEL (``Link.set_precision`` in runtime/core.py): a link that has one runs its ``__call__`` inside this
    scope, every Function remembers the scope it ran forward in and runs backward in it again, so an fp32 and a bf16 model
    live in one process (tests/test_gpu_bf16_storage.py::test_fp32_and_bf16_models_step_in_one_process).  COMPUTE / STORAGE
    above are the scope's current value: the process default outside any scope."""

    # The scope is PROCESS-wide (module globals, read by every launcher): one thread at a time may hold one.  A second thread
    # that asks for ANOTHER arithmetic while a scope is open would silently run -- or make the owner run -- in the wrong one
    # (ADVICE r5), so that is an error here; the same arithmetic, or no scope at all, is fine from any thread.
    _owner, _depth = None, 0

    def __init__(self, compute, storage=None):
        self.want = check_precision(compute, storage if storage is not None else ('f32' if compute == 'f32' else 'bf16'))

    def __enter__(self):
        global COMPUTE, STORAGE
        me = threading.get_ident()
        if precision._depth and precision._owner != me and self.want != (COMPUTE, STORAGE):
            raise RuntimeError('ops.precision%r entered on a second thread while another thread holds %r: the scope is '
                               'process-wide' % (self.want, (COMPUTE, STORAGE)))
        self.prev_owner = precision._owner
        precision._owner, precision._depth = me, precision._depth + 1
        self.old = (COMPUTE, STORAGE)
        COMPUTE, STORAGE = self.want
        return self

    def __exit__(self, *exc):
        global COMPUTE, STORAGE
        COMPUTE, STORAGE = self.old
        precision._depth -= 1
        precision._owner = self.prev_owner if precision._depth else None
        return False


def current_precision():
    return COMPUTE, STORAGE


def _is16(t):
    return t is not None and t.dtype == BF16


def cast_bf16(w):
    """bf16 operand copy of an fp32 parameter tensor (weights change every step: made per use, ~microseconds)."""
    out = _empty(w.shape, device=w.device, dtype=BF16)
    check(_lib.load().loans_cast_bf16(_ptr(w), _ptr(out), w.numel(), _stream()), 'loans_cast_bf16')
    return out


def _igemm_fn(lib):
    return lib.loans_igemm_bf16_f32 if COMPUTE == 'bf16' else lib.loans_igemm_f32


# When bench.py sets this to a list, conv_fprop brackets each launch with HIP events recorded on the
# launch stream and appends (tag, algorithmic_flops, start_event, end_event).
EVENT_LOG = None
# development probe (tools/host_lead.py): a list makes `probe(name)` record a HIP event on the current stream together with
# the host's clock at the moment it was enqueued -- where in a step the host runs ahead of the GPU, and where it does not
PROBE_LOG = None


def probe(name):
    log = PROBE_LOG
    if log is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        log.append((name, ev, time.perf_counter()))
# When bench.py sets this to a dict, every convolution launch (forward, data gradient, weight gradient) adds its ALGORITHMIC
# FLOP (2 per MAC, logical input channels, no padding / im2col redundancy) under 'fprop' / 'dgrad' / 'wgrad'.
FLOP_COUNT = None


def _addressed(size, out, k, stride, pad):
    """input rows (columns) that at least one tap of one output position addresses: all of them once k >= stride, every
    stride-th one for a strided 1x1"""
    if k >= stride:
        return size
    return len({o * stride + r - pad for o in range(out) for r in range(k)} & set(range(size)))


def _conv_bytes(geo, x, w, out):
    """ALGORITHMIC HBM bytes of one forward convolution as (read, written): the input pixels its taps address (a stride-2 1x1
    reads a quarter of its input tensor), its weights and its output once each, in the types they are stored in (x = None:
    the input is shared with a convolution already counted)"""
    n = 0
    if x is not None:
        n = x.numel() * x.element_size()
        if not geo.dense and geo.k < geo.stride:
            n = n * _addressed(geo.H, geo.Ho, geo.k, geo.stride, geo.pad) * _addressed(geo.W, geo.Wo, geo.k, geo.stride, geo.pad) \
                // (geo.H * geo.W)
    return n + w.numel() * w.element_size(), out.numel() * out.element_size()


def _sum_bytes(a, b):
    return a[0] + b[0], a[1] + b[1]


def _count_flops(kind, geo):
    if FLOP_COUNT is not None:
        FLOP_COUNT[kind] = FLOP_COUNT.get(kind, 0) + 2 * geo.B * geo.Ho * geo.Wo * geo.Cout * geo.k * geo.k * geo.cin_logical


# When bench.py sets this to a dict, every wrapper adds the ALGORITHMIC work of its launch(es) under a kernel class --
# [FLOP, bytes read, bytes written, seconds at the binding roofline summed per launch]: 2 FLOP per MAC of a contraction (logical channels, no padding), every tensor a pass must touch
# once, in its storage type (a BN backward that takes its own sums reads its operands twice: the sums must be complete before
# the first output) -- `roofline.whole_step.binding` prices the step's classes against max(bytes / HBM rate, FLOP / MFMA peak).
# Classes follow what a kernel trace can tell apart by name (tools/class_times.py): conv (forward + data gradient), wgrad,
# bn_fwd, bn_bwd, stem (pool forward, the stem's backward), crop (STN sampler + the crop gradient), heads, optimizer (+ the
# step's weight preparation and memsets).
CLASS_COUNT = None


def _nbytes(*tensors):
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


# what a launch is priced against (bench.py's constants; MI355X_MICROARCH.md): dense MFMA peak of the arm, achievable HBM rate
ROOFLINE_MFMA = {'f32': 157.3e12, 'bf16': 2.5e15}
ROOFLINE_HBM = 6.3e12


def _acct(cls, flop=0, rd=0, wr=0):
    c = CLASS_COUNT
    if c is not None:
        e = c.setdefault(cls, [0, 0, 0, 0.0])
        e[0] += flop
        e[1] += rd
        e[2] += wr
        e[3] += max(flop / ROOFLINE_MFMA[COMPUTE], (rd + wr) / ROOFLINE_HBM)       # this launch at the roofline that binds IT


def _conv_flop(geo):
    return 2 * geo.B * geo.Ho * geo.Wo * geo.Cout * geo.k * geo.k * geo.cin_logical

BN_EPS = 2e-5          # chainer.links.BatchNormalization default (sheep/resnet.py:44)
BN_DECAY = 0.9
# Chainer 4.1.0's CPU path folds eps into the running variance (see oracle/chainer_ops.py)
RUNNING_VAR_INCLUDES_EPS = 1


def _variant(geo, *what):
    """key of geo.tuned under which a wrapper keeps the tile it resolved for one call variant (which flags, which switches):
    later calls skip building the candidate lists, the mode string and the tuning closure -- pure host time, 600 convolution
    calls per ResNet-50 step.  Lives in geo.tuned, so whatever clears a shape's picks clears these too; never saved to a table."""
    return '~%r' % ((what, SPLITK, HALO, WGHALO, CLASS_LAUNCH, FINETAIL, STEM_DIRECT, COMPUTE, STORAGE, TUNE_POLICY, PW),)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)
_raw_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device


def _stream():
    """the current HIP stream of the current device as the `void* stream` of the C ABI.  (torch.cuda.current_stream().cuda_stream
    builds a Stream object and resolves the device index in Python on the way: 8 us, three thousand times per ResNet-50 step
    -- a fifth of the host's enqueue time, tools/host_profile.py; these two calls are the same lookup in C.)"""
    return _raw_stream(_raw_device())


def _memo(fn):
    """candidate lists / launch plans are pure functions of the geometry and of the module's switches, yet were rebuilt on EVERY
    call of a convolution wrapper, tuned or not (1.4 ms of host time per ResNet-50 step for the weight gradients alone)"""
    cache = {}

    def wrapped(*args, **kw):
        if kw:
            return fn(*args, **kw)
        key = (tuple(id(a) if isinstance(a, dict) else a for a in args), SPLITK, HALO, WGHALO, CLASS_LAUNCH, FINETAIL,
               STEM_DIRECT, COMPUTE, STORAGE, PW)
        try:
            return cache[key]
        except KeyError:
            if len(cache) >= 8192:      # (keys hold their geometries alive: a process that keeps making new ones starts over)
                cache.clear()
            v = cache[key] = fn(*args)
            return v
        except TypeError:               # an unhashable argument: not memoised
            return fn(*args)
    wrapped.__name__, wrapped.__doc__, wrapped.uncached = fn.__name__, fn.__doc__, fn
    return wrapped


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _chk(t, name='tensor'):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError('%s must be a contiguous float32 device tensor' % name)


# --------------------------------------------------------------------------- #
# convolution geometry -> descriptors
# --------------------------------------------------------------------------- #
def conv_outsize(size, k, s, p, cover_all=False):
    if cover_all:
        return (size + p * 2 - k + s - 1) // s + 1
    return (size + p * 2 - k) // s + 1


_TUNE_CACHE = {}


# ---- persisted tile table ---------------------------------------------------------------------------------------------------
# The autotuner decides by timing, so a run under a profiler (rocprofv3 --pmc serialises and slows every dispatch) would
# pick other tiles than the run whose numbers are being explained.  A normal run therefore WRITES its table
# (`save_tune_table`; bench.py --tune-file F when F does not exist yet) and the profiled runs of the same command READ it
# (`load_tune_table`; LOANS_TUNE_FILE=F or bench.py --tune-file F with F present): every shape found in the table skips
# the timing and launches exactly the kernels of the timed run; shapes that are not in it are tuned as usual.
def _tune_key_str(key):
    return ','.join(str(int(v)) for v in key)


# Tile ids are this library's: bump when an id changes meaning, so that tables written by an older build are not launched as-is
TUNE_SCHEMA = 3
_TUNE_LOADED = {}        # shape key -> {mode: tile} read from a file: a proposal, accepted per use if the tile is on offer (_tuned_tile)


def _tune_stamp():
    """what a table's timings were taken on: entries are only trusted on the same chip and tile-id schema"""
    if not torch.cuda.is_available():
        return {"arch": None, "compute_units": 0, "schema": TUNE_SCHEMA}
    p = torch.cuda.get_device_properties(torch.cuda.current_device())
    # the ISA name, not the marketing name (which some boxes of the pool report as an empty string)
    return {"arch": str(getattr(p, 'gcnArchName', '')).split(':')[0], "compute_units": int(p.multi_processor_count), "schema": TUNE_SCHEMA}


def save_tune_table(path):
    import json
    table = {_tune_key_str(k): {m: t for m, t in v.items() if not m.startswith('~')} for k, v in sorted(_TUNE_CACHE.items()) if v}
    table = {k: v for k, v in table.items() if v}
    with open(path, 'w') as f:
        json.dump({"what": "loans_amd tile table: (B,H,W,Cin,Cout,k,stride,pad,dense) -> {mode: tile id | splits << 8 | class launch << 16}",
                   "stamp": _tune_stamp(), "entries": table}, f, indent=1, sort_keys=True)
    return len(table)


def load_tune_table(path):
    """Read a saved table as PROPOSALS (entries tuned in this process win; a proposed tile is launched only if the current
    candidate list of its problem still offers it -- an ablation switch, another library revision or another chip retunes
    instead of launching a stale id).  A table stamped for another device / CU count / tile-id schema is ignored.  Returns
    the number of shapes read."""
    import json
    with open(path) as f:
        doc = json.load(f)
    stamp = doc.get("stamp")
    if stamp is not None and any(stamp.get(k) != v for k, v in _tune_stamp().items()):
        return 0
    if stamp is None and doc.get("entries") and TUNE_SCHEMA > 2:
        return 0                    # a table from before the stamps (rounds 1-2): mode keys and tile offers have changed since
    table = doc["entries"]
    for ks, modes in table.items():
        key = tuple(int(v) for v in ks.split(','))
        key = key[:-1] + (bool(key[-1]),)
        cur = _TUNE_LOADED.setdefault(key, {})
        for mode, tile in modes.items():
            cur.setdefault(mode, int(tile))
    return len(table)


if os.environ.get('LOANS_TUNE_FILE') and os.path.exists(os.environ['LOANS_TUNE_FILE']):
    load_tune_table(os.environ['LOANS_TUNE_FILE'])


class ConvGeometry:
    """Descriptors for one Convolution2D at a fixed input shape (cached by the
    caller): forward, weight-gradient and the per-stride-parity-class data
    gradient launches.  Channels are the PHYSICAL (multiple-of-4) counts."""

    def __init__(self, B, H, W, Cin, Cout, k, stride, pad, dense=False):
        self.B, self.H, self.W, self.Cin, self.Cout = B, H, W, Cin, Cout
        self.k, self.stride, self.pad = k, stride, pad
        self.Ho, self.Wo = conv_outsize(H, k, stride, pad), conv_outsize(W, k, stride, pad)
        self.dense = dense
        self.base_flags = 0
        self.cin_logical = 3 if Cin == 4 else Cin
        # tile choices are a property of the problem shape, not of the layer: equal convs share one table
        self.key = (B, H, W, Cin, Cout, k, stride, pad, dense)
        self.tuned = _TUNE_CACHE.setdefault(self.key, {})
        if dense:
            self._init_dense()
            return
        self.in_numel = B * H * W * Cin
        self.w_numel = Cout * k * k * Cin
        d = IgemmDesc()
        d.B, d.inH, d.inW, d.Cin = B, H, W, Cin
        d.outH, d.outW, d.Cout = self.Ho, self.Wo, Cout
        d.gridH, d.gridW = self.Ho, self.Wo
        d.osy = d.osx = 1
        d.oy0 = d.ox0 = 0
        d.isy = d.isx = stride
        d.ntaps = k * k
        t = 0
        for r in range(k):
            for s in range(k):
                d.dy[t], d.dx[t] = r - pad, s - pad
                t += 1
        self.fwd = d
        # data gradient: gathered tensor = gy (Ho x Wo x Cout), output = gx (H x W x Cin)
        self.dgrad = []          # list of (desc, tapsel, weight_offset_in_floats)
        self.dgrad_has_empty_class = False
        off = 0
        for cy in range(stride):
            for cx in range(stride):
                gh = (H - cy + stride - 1) // stride
                gw = (W - cx + stride - 1) // stride
                if gh <= 0 or gw <= 0:
                    continue
                taps = [(r, s) for r in range(k) for s in range(k)
                        if (cy + pad - r) % stride == 0 and (cx + pad - s) % stride == 0]
                if not taps:        # k < stride (1x1 / stride 2): this class of input pixels gets no gradient
                    self.dgrad_has_empty_class = True
                    continue
                g = IgemmDesc()
                g.B, g.inH, g.inW, g.Cin = B, self.Ho, self.Wo, Cout
                g.outH, g.outW, g.Cout = H, W, Cin
                g.gridH, g.gridW = gh, gw
                g.osy = g.osx = stride
                g.oy0, g.ox0 = cy, cx
                g.isy = g.isx = 1
                g.ntaps = len(taps)
                for t, (r, s) in enumerate(taps):
                    g.dy[t] = (cy + pad - r) // stride
                    g.dx[t] = (cx + pad - s) // stride
                tapsel = (C.c_int32 * len(taps))(*[r * k + s for r, s in taps])
                self.dgrad.append((g, tapsel, off))
                off += Cin * len(taps) * Cout
        self.dgrad_weight_floats = off
        self.flops_fwd = 2 * B * self.Ho * self.Wo * Cout * k * k * Cin

    def _init_dense(self):
        """LOANS_F_DENSE (include/loans_hip.h): the RGB stem reads packed 3-channel rows of a zero-padded frame, so
        its K is k rows x (kwp pixels x 3) = 7 x 24 = 168 instead of 49 taps x 4 padded channels = 196 and the
        loader needs no bounds masks.  kwp - k extra window pixels meet zero weights."""
        B, H, W, k, s, p = self.B, self.H, self.W, self.k, self.stride, self.pad
        assert self.Cin == 3
        self.kwp = dense_window(k)
        self.Hp = H + 2 * p
        self.Wp = max(W + 2 * p, s * (self.Wo - 1) + self.kwp)
        self.Wp += self.Wp & 1                  # even: every K row starts on an 8-byte boundary
        self.in_numel = B * self.Hp * self.Wp * 3
        self.w_numel = self.Cout * k * self.kwp * 3
        self.base_flags = F_DENSE
        d = IgemmDesc()
        d.B, d.inH, d.inW, d.Cin = B, self.Hp, self.Wp * 3, self.kwp * 3
        d.outH, d.outW, d.Cout = self.Ho, self.Wo, self.Cout
        d.gridH, d.gridW = self.Ho, self.Wo
        d.osy = d.osx = 1
        d.oy0 = d.ox0 = 0
        d.isy, d.isx = s, s * 3
        d.ntaps = k
        for t in range(k):
            d.dy[t], d.dx[t] = t, 0
        self.fwd = d
        self.dgrad, self.dgrad_has_empty_class, self.dgrad_weight_floats = [], False, 0      # frames get no gradient
        self.flops_fwd = 2 * B * self.Ho * self.Wo * self.Cout * k * k * 3
        self._wmask = None

    def wmask(self, device):
        """1 on the real weights, 0 on the window-padding columns (kx >= k) of the dense layout."""
        if self._wmask is None or self._wmask.device != device:
            m = np.zeros((self.Cout, self.k, self.kwp, 3), np.float32)
            m[:, :, :self.k] = 1.0
            self._wmask = torch.from_numpy(m).to(device)
        return self._wmask


def dense_window(k):
    """Pixels per K row of the dense RGB layout: the smallest window >= k whose 3-channel run is whole float4s."""
    kwp = k
    while (kwp * 3) % 4:
        kwp += 1
    return kwp


def _with_flags(desc, flags, tile=0):
    desc.flags = flags
    desc.tile = tile
    return desc


# --------------------------------------------------------------------------- #
# tile autotuning: the first time a geometry is used, every tile variant of the
# kernel is timed on it (HIP events, scratch outputs) and the fastest is kept.
# Results do not depend on the tile: K is accumulated in the same order by all.
# --------------------------------------------------------------------------- #
AUTOTUNE = True          # False: every launch on the library's default tile
# candidates.  LOANS_TILE_SPLIT (6) only pays where nothing else shares the machine: in backward the dgrad launches run
# beside the weight-gradient GEMMs of the side stream and the two-launch split measured slower there
# +16 (LOANS_TILE_DMA): the same tile shape with its operands staged by LDS-DMA (fp32 arm only)
_FPROP_TILES = (1, 2, 3, 4, 6, 17, 18, 19, 20, 22)
_IGEMM_TILES = (1, 2, 3, 17, 18, 19)
_WGRAD_TILES = (1, 3, 5)
_WGRAD16_TILES = (1, 3, 5)
_IGEMM16_TILES = (1, 2, 3, 4, 7)
# LOANS_TILE_HALO_* (csrc/halo_bf16.hip): stride-1 convolutions with the input tile staged once per 64-channel chunk
TILE_HALO_128, TILE_HALO_256x64, TILE_HALO_128x64, TILE_HALO_128x64S, TILE_WS64 = 11, 12, 13, 14, 15
TILE_HALO_256x128 = 36       # one 512-thread block per CU: 16 x 16 pixels x 128 output channels
TILE_HALO_256x256 = 42       # the same with 256 output channels (eight 128 x 64 wave tiles, two-pass epilogue): the res4 / res5 layers
TILE_WSW64 = 37              # weights stationary, every wave on its own 2 x 16 pixel unit (no block barriers)
HALO = True


TILE_PW = 40                 # LOANS_TILE_PW (csrc/pw_bf16.hip): short-K 1 x 1 convolutions, operands never in LDS
PW = True


@_memo
def _pw_tiles(geo, plain):
    """LOANS_TILE_PW where loans_pw16_covers holds: a 1 x 1 / 1 convolution, Cin 64 or 128 with Cout a multiple of 64 up to 512 or
    Cin 256 with Cout a multiple of 128 up to 1024, no epilogue beyond the BN statistics (plain = no ReLU on the input, no bias, no
    addend).  ResNet-50's res2 / res3 / res4 expansions."""
    if not PW or not plain or geo.dense or geo.k != 1 or geo.stride != 1 or geo.pad != 0:
        return ()
    if geo.Cin == 256:
        return (TILE_PW,) if geo.Cout % 128 == 0 and geo.Cout <= 1024 else ()
    if geo.Cin not in (64, 128) or geo.Cout % 64 or geo.Cout > 512:
        return ()
    return (TILE_PW,)


# A bottleneck's bn2 -> relu -> conv3 without the activation tensor between them (round 5): conv3's forward
# (LOANS_TILE_PW) and its weight gradient apply relu(x * scale + shift) while they load bn2's INPUT (LOANS_F_AFFINE_IN), the apply
# pass that read it and wrote the activation disappears.  False: the activation is materialised as before.
BN_ON_LOAD = True


def affine_in_ok(geo, x):
    """conv_fprop_affine / conv_wgrad(..., in_affine=) cover this convolution: bf16 storage, a shape LOANS_TILE_PW takes"""
    return BN_ON_LOAD and PW and WGRAD_SLABS and _is16(x) and x.is_contiguous() and bool(_pw_tiles(geo, True))


def conv_fprop_affine(x, st, w, geo, stats=None):
    """conv(relu(x * st.scale + st.shift)) on bf16 tensors: the BN + ReLU in front of a 1 x 1 convolution applied on load, bit for
    bit bn_apply(x, st, relu=True) followed by conv_fprop(..., tile=TILE_PW)"""
    assert affine_in_ok(geo, x) and x.numel() == geo.in_numel
    lib = _lib.load()
    out = _empty((geo.B, geo.Ho, geo.Wo, geo.Cout), device=x.device, dtype=BF16)
    w16 = w if _is16(w) else _bf16_shadow(w)
    if w16 is None:
        w16 = cast_bf16(w)
    flags = F_AFFINE_IN | (F_STATS if stats is not None else 0)
    _count_flops('fprop', geo)
    if CLASS_COUNT is not None:
        rd, wr = _conv_bytes(geo, x, w16, out)
        _acct('conv', _conv_flop(geo), rd, wr)
    log = EVENT_LOG
    if log is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    packs = PW_PACK_CALLS
    wp = _pw_packed(lib, w, w16, geo, _stream())
    packs = PW_PACK_CALLS - packs
    check(lib.loans_igemm_bf16s(_ptr(x), _ptr(wp), _ptr(out), _ptr(st.affine()), _ptr(stats), 0, 0,
                                C.byref(_with_flags(geo.fwd, flags, TILE_PW)), _stream()), 'loans_igemm_bf16s[fprop, bn on load]')
    if log is not None:
        ev1.record()
        log.append(('fprop_bn' if stats is not None else 'fprop',
                    2 * geo.B * geo.Ho * geo.Wo * geo.Cout * geo.k * geo.k * geo.cin_logical, ev0, ev1, 1 + packs, 1,
                    _conv_bytes(geo, x, w16, out)))
    return out


PW_PACK_CALLS = 0           # per-call packing launches (tests: a prepared step makes none)


def _pw_packed(lib, w, w16, geo, st):
    """the weights in LOANS_TILE_PW's fragment order.  Inside a step: the buffer this step's begin_step filled for (w, Cout, Cin) -- all
    such layers in one launch, from the fp32 masters (_WeightPrep.begin) -- or, the first time a layer asks, a buffer registered for
    the next steps and packed now; outside a step (tests, inference) one small launch per call."""
    global PW_PACK_CALLS
    wp = _weight_preps.get(w.device.index)
    packed = None
    # (Cin = 256 stays per call: its 512 KB matrix, packed at begin_step, has left the L2 by the time res4 runs, and the tile loop
    # of 2048 waves in step then pays an HBM miss per fragment group -- 0.082 ms in the step against 0.060 with the packing launch
    # right in front of it as the warm-up; the 32 / 128 KB matrices of res2 / res3 do not care: 0.148 -> 0.139, 0.088 -> 0.084)
    if wp is not None and wp.live and WEIGHT_PREP and w.dtype == torch.float32 and w.is_contiguous() and geo.Cin < 256:
        key = (w.data_ptr(), geo.Cout, geo.Cin)
        entry = wp.pw.get(key)
        if entry is not None:
            entry['used'] = wp.step
            if torch.cuda.is_current_stream_capturing():
                wp.captured_pw_keys.add(key)       # a graph being recorded has this buffer's address baked in: never evicted
            if key in wp.pw_prepared:
                return entry['buf']
        elif len(wp.pw_order) < _MAX_PREP_JOBS and not torch.cuda.is_current_stream_capturing():
            entry = wp.pw[key] = {'buf': torch.empty(geo.Cout * geo.Cin, device=w.device, dtype=BF16), 'w': w, 'used': wp.step}
            wp.pw_order.append(key)
            wp.pw_dirty = True
        if entry is not None:
            packed = entry['buf']
    if packed is None:
        packed = _empty(geo.Cout * geo.Cin, device=w16.device, dtype=BF16)
    PW_PACK_CALLS += 1
    check(lib.loans_pw_pack_bf16(_ptr(w16), _ptr(packed), geo.Cout, geo.Cin, st), 'loans_pw_pack_bf16')
    return packed


TILE_256x256 = 9        # LOANS_TILE_256x256 (loans_igemm_bf16s): eight 128 x 64 wave tiles, for GEMMs with >= 256 columns
TILE_256x256PP = 43     # LOANS_TILE_256x256PP: the same tile with a ping-pong K loop (csrc/igemm16_pp.h); not for the dense stem
TILE_256x256PP16 = 44   # ... on v_mfma_f32_16x16x32_bf16 (not bit-identical to the 32 x 32 x 16 tiles: 32 k values per MFMA)


TILE_DEEP = 32          # LOANS_TILE_DEEP: a longer LDS ring for grids of about one block per CU


@_memo
def _wide16_tiles(columns, rows=None):
    """further tile forms of loans_igemm_bf16s by GEMM shape: 256 x 256 where the columns fill it, the deep-ring forms of the
    small tiles where the grid is small (at most four 64 x 64 blocks per CU of an MI355X)"""
    t = (TILE_256x256, TILE_256x256PP, TILE_256x256PP16) if columns % 256 == 0 else ()
    if rows is not None and ((rows + 63) // 64) * ((columns + 63) // 64) <= 1024:
        t += (1 | TILE_DEEP, 2 | TILE_DEEP, 3 | TILE_DEEP)
    return t


@_memo
def _splitk16_candidates(rows, out_channels, ktot):
    """split-K forms of the bf16 implicit GEMM (loans_igemm_bf16s_splitk) for grids that cannot fill the machine: few tiles,
    long K (res6 / res7 at 512 px, everything at small batch).  tile id = base tile | (splits << 8)."""
    nchunks = (ktot + 63) // 64
    out = []
    c8 = out_channels // 8
    if not SPLITK or out_channels % 8 or c8 > 256 or 256 % c8 or nchunks < 16:
        return ()
    for base, bm, bn in ((3, 64, 64), (2, 128, 64)):
        tiles = ((rows + bm - 1) // bm) * ((out_channels + bn - 1) // bn)
        # below one tile per CU only: at 256 tiles (res7 of configs[2]) the memset, the atomics and the finalize pass cost what
        # the second resident block gains (measured: 0.036 - 0.039 against 0.041 ms per conv, nothing on the step)
        if tiles >= 256:
            continue
        out += [base | (sp << 8) for sp in (2, 4, 8) if tiles * sp <= 2048 and nchunks // sp >= 4]
    return tuple(out)


def _igemm16_splitk(lib, src, d_list, out, flags, tile, bias, stats, ref, addend, rows, Cout, st):
    """split-K convolution on bf16 storage: zeroed fp32 workspace, raw partial launches (one per descriptor: the parity classes
    of a strided data gradient share the workspace), one finalize pass with the epilogue flags.  d_list = [(desc, weights)]."""
    partial = _empty((rows, Cout), device=out.device, dtype=torch.float32).zero_()
    for d, wt in d_list:
        check(lib.loans_igemm_bf16s_splitk(_ptr(src), _ptr(wt), _ptr(partial),
                                           C.byref(_with_flags(d, flags & (F_RELU_IN | F_DENSE), tile & 0xFF)), tile >> 8, st),
              'loans_igemm_bf16s_splitk')
    check(lib.loans_igemm_finalize_bf16(_ptr(partial), _ptr(out), _ptr(bias), _ptr(stats), _ptr(ref), _ptr(addend),
                                        flags & (F_BIAS | F_STATS | F_MASK | F_ADDEND | F_ADDEND_MASK), rows, Cout, st),
          'loans_igemm_finalize_bf16')


@_memo
def _halo_tiles(geo, gathered_channels, out_channels, out_hw, relu_in=False):
    """halo-tile candidates of a bf16-storage convolution / data gradient (the conditions of loans_halo16_covers), offered
    where a 8 x 16 pixel tile is not mostly empty"""
    if not HALO or geo.dense or geo.stride != 1 or geo.k > 3 or gathered_channels % 64 or min(out_hw) < 6 or out_hw[1] < 12:
        return ()
    tiles = (TILE_HALO_128, TILE_HALO_128x64) if out_channels > 64 else (TILE_HALO_128x64,)
    if out_channels > 64 and min(out_hw) >= 12:
        tiles += (TILE_HALO_256x128,)
    if out_channels >= 256 and min(out_hw) >= 12:
        tiles += (TILE_HALO_256x256,)
    if gathered_channels == 64 and out_channels <= 64:
        tiles += (TILE_HALO_128x64S,) + ((TILE_HALO_256x64,) if min(out_hw) >= 12 else ())
        if geo.k == 3 and geo.pad == 1 and not relu_in and min(out_hw) >= 12:
            tiles += (TILE_WS64, TILE_WSW64)
    return tiles


# cold timing (shapes that offer LOANS_TILE_PW): a 512 MB fill between the timed launches of the autotuner, so that every candidate
# finds its operands in HBM, where the step's previous kernel left them, and not in the 256 MB Infinity Cache, where the
# candidate's own previous repetition did.  (For every shape -- LOANS_TUNE_COLD=1 of round 3 -- it was worth 0.6-1 % at ResNet-50
# and nothing elsewhere: removed.)
_cold = {}


def _time_call(fn, reps=5, cold=False):
    fn()
    best = float('inf')
    for _ in range(reps):
        if cold:
            dev = torch.cuda.current_device()
            if dev not in _cold:
                _cold[dev] = torch.empty(512 << 20, device='cuda', dtype=torch.uint8)
            _cold[dev].zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


# How a problem shape gets its tile when neither this process nor a loaded table has one:
#   'time'  (default; bench.py, the trainer): every candidate is timed on the shape, the fastest is kept -- the pick depends on
#           the box, on what else the GPU is doing and on chance where two tiles are within noise of each other;
#   'fixed' (the parity test session, tests/conftest.py): candidate number crc32(shape, mode, LOANS_TUNE_SALT) % n -- the same
#           kernels on every box and in every run, and over the layer shapes of a network nearly every tile form on offer.
#           A test session never times: TIMED_PICKS counts the picks that timing decided and conftest asserts it stays put.
TUNE_POLICY = os.environ.get('LOANS_TUNE_POLICY', 'time')
TUNE_SALT = os.environ.get('LOANS_TUNE_SALT', '0')
TUNE_VERBOSE = False        # development tools set this: one line per pick
TIMED_PICKS = 0
assert TUNE_POLICY in ('time', 'fixed'), TUNE_POLICY


def _fixed_pick(geo, mode, candidates):
    import zlib
    return candidates[zlib.crc32(('%s|%s|%s' % (_tune_key_str(geo.key), mode, TUNE_SALT)).encode()) % len(candidates)]


def _tuned_tile(geo, mode, run, candidates, cold=False):
    """run(tile) launches the op into scratch buffers."""
    global TIMED_PICKS
    tile = geo.tuned.get(mode)
    if tile is not None:
        return tile
    if not AUTOTUNE:
        geo.tuned[mode] = 0
        return 0
    if COMPUTE == 'bf16':
        candidates = [t for t in candidates if not (t & 16)]
    candidates = list(candidates)
    tile = _TUNE_LOADED.get(geo.key, {}).get(mode)
    if tile is not None and tile in candidates:         # a file's proposal, still on offer for this problem
        geo.tuned[mode] = tile
        return tile
    if TUNE_POLICY == 'fixed':
        tile = geo.tuned[mode] = _fixed_pick(geo, mode, candidates)
        if TUNE_VERBOSE:
            print('[tune] %-5s B=%d %dx%dx%d -> %d k%d s%d : fixed pick, tile %d of %s' % (
                mode, geo.B, geo.H, geo.W, geo.Cin, geo.Cout, geo.k, geo.stride, tile, candidates), flush=True)
        return tile
    TIMED_PICKS += 1
    times = {t: _time_call(lambda: run(t), cold=cold) for t in candidates}
    _cold.clear()           # the 512 MB fill buffer of a cold-timed shape is not kept for the life of the process
    tile = min(times, key=times.get)
    geo.tuned[mode] = tile
    if TUNE_VERBOSE:
        print('[tune] %-5s B=%d %dx%dx%d -> %d k%d s%d : %s -> tile %d' % (
            mode, geo.B, geo.H, geo.W, geo.Cin, geo.Cout, geo.k, geo.stride,
            ' '.join('%d:%.3fms' % kv for kv in sorted(times.items())), tile), flush=True)
    return tile


def reduce_channels_ok(C_):
    """channel counts the row-reduction kernels (and loans_igemm_finalize_f32) tile: C/4 divides 256 or is a multiple of it"""
    c4 = C_ // 4
    return C_ % 4 == 0 and c4 >= 1 and ((256 % c4 == 0) if c4 <= 256 else (c4 % 256 == 0))


# Split-K adds partial tiles with atomics: the summation order, hence the last bits of a small-batch forward, vary from
# run to run.  LOANS_SPLITK=0 (or ops.SPLITK = False) keeps the forward convolutions bit-reproducible.
SPLITK = os.environ.get('LOANS_SPLITK', '1') != '0'


@_memo
def _splitk_candidates(M, Cout, nchunks):
    """split-K forms of the 64x64 tile (LOANS_TILE_SPLITK) for grids that cannot fill the machine: few tiles, long K --
    the deep layers at small batch and single-image inference.  tile id = 3 | (splits << 8)."""
    tiles = ((M + 63) // 64) * ((Cout + 63) // 64)
    if not SPLITK or COMPUTE != 'f32' or tiles >= 512 or nchunks < 16:
        return ()
    return tuple(3 | (s << 8) for s in (2, 4, 8, 16) if tiles * s <= 4096 and nchunks // s >= 4)


def _igemm_splitk(lib, fn_in, w, out, d_list, flags, tile, bias, stats, ref, addend, rows, Cout, st):
    """A split-K convolution: zero the output (unless it already holds the addend), the launches ADD raw partial tiles,
    one finalize pass applies the epilogue flags to the finished sums.  d_list = [(desc, weight_tensor)]."""
    inplace = addend is not None and addend.data_ptr() == out.data_ptr()
    if not inplace:
        out.zero_()
    for d, wt in d_list:
        check(lib.loans_igemm_f32(_ptr(fn_in), _ptr(wt), _ptr(out), 0, 0, 0, 0,
                                  C.byref(_with_flags(d, flags & (F_RELU_IN | F_DENSE), tile)), st), 'loans_igemm_f32[split-K]')
    fin = flags & (F_BIAS | F_STATS | F_MASK | F_ADDEND | F_ADDEND_MASK)
    if inplace:
        assert not (fin & (F_MASK | F_ADDEND_MASK)), 'a masked conv term cannot be summed into an aliased addend'
        fin &= ~F_ADDEND
    if fin:
        check(lib.loans_igemm_finalize_f32(_ptr(out), _ptr(bias), _ptr(stats), _ptr(ref), _ptr(None if inplace else addend),
                                           fin, rows, Cout, st), 'loans_igemm_finalize_f32')


TILE_STEM = 10          # LOANS_TILE_STEM: the dense RGB stem as a direct convolution (csrc/stem.hip)


@_memo
def stem_tile_rows(geo):
    """output rows per block of LOANS_TILE_STEM for this geometry, 0 = not covered (loans_stem7_rows of csrc/stem.hip)"""
    if not (geo.dense and geo.k == 7 and geo.stride == 2 and geo.pad == 3 and geo.Cout == 64):
        return 0
    wp3 = geo.Wp * 3
    if 2 * geo.Ho + 5 > geo.Hp or geo.Hp % 2:
        return 0
    for R in (4, 2, 1):
        if geo.Ho % R or (R * geo.Wo) % 64 or R * geo.Wo // 32 > 14:
            continue
        if ((2 * R + 5) * wp3 + 154 * 65) * 4 > 80 * 1024:
            continue
        return R
    return 0


@_memo
def stem_wgrad_ok(geo):
    """LOANS_TILE_STEM of loans_wgrad_f32 covers this geometry (loans_stem7_wgrad_launch of csrc/stem.hip)"""
    if not (STEM_DIRECT and geo.dense and geo.k == 7 and geo.stride == 2 and geo.pad == 3 and geo.Cout == 64):
        return False
    if 2 * geo.Ho + 5 > geo.Hp or geo.Hp % 2:
        return False
    return 2 * ((7 * geo.Wp * 3 * 4 + 1023) // 1024 + (geo.Wo + 3) // 4) * 1024 <= 156 * 1024


STEM_WGRAD16_DIRECT = True      # (False: the A/B arm of tools/ab_ops_attr.py)


def stem16_wgrad_ok(geo):
    """LOANS_TILE_STEM of loans_wgrad_bf16s covers this geometry (loans_stem7_wgrad_bf16_slabs of csrc/stem.hip: the bf16 frame
    buffer, rows of whole 16-pixel steps and whole 12-byte cells, at most 256 output pixels per row)"""
    if not (STEM_DIRECT and STEM_WGRAD16_DIRECT and geo.dense and geo.k == 7 and geo.stride == 2 and geo.pad == 3 and geo.Cout == 64):
        return False
    if 2 * geo.Ho + 5 > geo.Hp or geo.Hp % 2 or geo.Wo % 16 or geo.Wo > 256 or geo.Wp != 2 * geo.Wo + 6:
        return False
    return 7 * (geo.Wo + 3) <= 2048 and 2 * (7 * (geo.Wo + 3) * 16 + geo.Wo * 144) <= 156 * 1024


@_memo
def stem16_tile_rows(geo):
    """output rows per unit of LOANS_TILE_STEM on the bf16 MFMA (loans_stem7_bf16_rows of csrc/stem.hip), 0 = not covered"""
    if not (geo.dense and geo.k == 7 and geo.stride == 2 and geo.pad == 3 and geo.Cout == 64):
        return 0
    if 2 * geo.Ho + 5 > geo.Hp or geo.Hp % 2:
        return 0
    for R in (4, 2, 1):
        if geo.Ho % R:
            continue
        if ((((2 * R + 5) * geo.Wp * 3 + 8) * 2 + 15) & ~15) + 4 * 32 * 68 * 4 > 78 * 1024:
            continue
        return R
    return 0


TILE_FINETAIL = 8       # LOANS_TILE_FINETAIL (+16 = LDS-DMA): whole 64x64 tiles, then K-slices of the uneven rest in the same launch
FINETAIL = True
STEM_DIRECT = True


@_memo
def _finetail_plan(M, Cout, nchunks, device):
    """(rows computed at full K, K slices of the remaining tiles) of LOANS_TILE_FINETAIL -- the arithmetic of igemm_impl"""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    tiles_n = (Cout + 63) // 64
    ntile = ((M + 63) // 64) * tiles_n
    n_full = ntile // cus * cus
    n_full -= n_full % tiles_n
    n_tail = ntile - n_full
    sl = min(16, cus // n_tail) if n_tail > 0 else 0
    while sl > 1 and nchunks // sl < 4:
        sl -= 1
    if n_full <= 0 or n_tail <= 0 or sl < 2:
        return M, 1
    return n_full // tiles_n * 64, sl


def _igemm_launches(M, Cout, tile, device, nchunks=0):
    """Kernel launches behind one loans_igemm call: LOANS_TILE_SPLIT is two when both row ranges are non-empty,
    LOANS_TILE_FINETAIL two (the convolution and the finalize pass over the sliced rows) when it slices
    (same arithmetic as igemm_impl)."""
    if (tile & 15) == TILE_FINETAIL:
        return 2 if _finetail_plan(M, Cout, nchunks, device)[1] > 1 else 1
    if (tile & 15) != 6:
        return 1
    slots = 2 * torch.cuda.get_device_properties(device).multi_processor_count
    tiles_n = (Cout + 127) // 128
    rows_big = ((M // 128) * tiles_n // slots) * slots // tiles_n * 128
    return 2 if 0 < rows_big < M else 1


def conv_fprop(x, w, geo, out=None, bias=None, stats=None, relu_in=False, addend=None, tile=0, out_bf16=False):
    """out[B,Ho,Wo,Cout] = conv(x[B,H,W,Cin], w[Cout,k,k,Cin]) (+bias) (+addend); optional BN statistics.
    out_bf16 (bf16 compute arm, fp32 input): write a bf16 tensor -- the stem conv of the bf16-storage arm."""
    lib = _lib.load()
    if _is16(x):
        return _conv_fprop16(lib, x, w, geo, out, bias, stats, relu_in, addend, tile)
    if out_bf16:
        assert COMPUTE == 'bf16' and addend is None and out is None
        out = _empty((geo.B, geo.Ho, geo.Wo, geo.Cout), device=x.device, dtype=BF16)
    if out is None:
        out = _empty((geo.B, geo.Ho, geo.Wo, geo.Cout), device=x.device, dtype=torch.float32)
    flags = (F_RELU_IN if relu_in else 0) | (F_BIAS if bias is not None else 0) | (F_OUT_BF16 if out_bf16 else 0) | \
            (F_STATS if stats is not None else 0) | (F_ADDEND if addend is not None else 0) | geo.base_flags
    assert x.numel() == geo.in_numel and w.numel() == geo.w_numel
    if tile == 0:
        vkey = _variant(geo, 'fprop', stats is not None, out_bf16, addend is None, relu_in, bias is not None)
        tile = geo.tuned.get(vkey, 0)
    if tile == 0:
        tflags = flags & (F_RELU_IN | F_STATS | F_DENSE | F_OUT_BF16)
        sstats = stats_buffer(geo.Cout, x.device) if stats is not None else None

        def run(t):
            scratch = _empty((geo.B, geo.Ho, geo.Wo, geo.Cout), device=x.device, dtype=BF16 if out_bf16 else torch.float32)
            if t >> 8:
                _igemm_splitk(lib, x, w, scratch, [(geo.fwd, w)], tflags, t, None, sstats, None, None,
                              geo.B * geo.Ho * geo.Wo, geo.Cout, _stream())
                return
            check(_igemm_fn(lib)(_ptr(x), _ptr(w), _ptr(scratch), 0, _ptr(sstats), 0, 0,
                                      C.byref(_with_flags(geo.fwd, tflags, t)), _stream()), 'loans_igemm_f32[tune]')
        M_ = geo.B * geo.Ho * geo.Wo
        nch = (geo.w_numel // geo.Cout + 31) // 32
        sk = () if (out_bf16 or not reduce_channels_ok(geo.Cout)) else _splitk_candidates(M_, geo.Cout, nch)
        # LOANS_TILE_FINETAIL slices its last tiles along K (atomics): offered under the same switch as split-K
        ft = ()
        if SPLITK and FINETAIL and COMPUTE == 'f32' and not out_bf16 and addend is None and reduce_channels_ok(geo.Cout) \
                and _finetail_plan(M_, geo.Cout, nch, x.device)[1] > 1:
            ft = (TILE_FINETAIL, TILE_FINETAIL | 16)
        stem = ()
        if STEM_DIRECT and not relu_in and addend is None:
            if (stem_tile_rows(geo) if not out_bf16 else stem16_tile_rows(geo)) and (COMPUTE == 'f32') != out_bf16:
                stem = (TILE_STEM,)
        tile = _tuned_tile(geo, COMPUTE + 'fprop' + ('_stats' if stats is not None else '') + ('_sk' if sk else '') +
                           ('_ft' if ft else '') + ('_st' if stem else ''), run,
                           _FPROP_TILES + sk + ft + stem)   # fp32 scratch output: the tile choice carries over
        geo.tuned[vkey] = tile
    d = _with_flags(geo.fwd, flags, tile)
    _count_flops('fprop', geo)
    if CLASS_COUNT is not None:
        rd, wr = _conv_bytes(geo, x, w, out)
        _acct('conv', _conv_flop(geo), rd + _nbytes(addend), wr)
    log = EVENT_LOG
    if log is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if tile >> 8:
        _igemm_splitk(lib, x, w, out, [(geo.fwd, w)], flags, tile, bias, stats, None, addend,
                      geo.B * geo.Ho * geo.Wo, geo.Cout, _stream())
    else:
        check(_igemm_fn(lib)(_ptr(x), _ptr(w), _ptr(out), _ptr(bias), _ptr(stats), 0, _ptr(addend),
                             C.byref(d), _stream()), 'loans_igemm[fprop]')
    if log is not None:
        ev1.record()
        # algorithmic FLOPs: logical input channels (3 for the RGB stem), no padding, no im2col redundancy
        log.append(('fprop_bn' if stats is not None else 'fprop',
                    2 * geo.B * geo.Ho * geo.Wo * geo.Cout * geo.k * geo.k * geo.cin_logical, ev0, ev1,
                    _igemm_launches(geo.B * geo.Ho * geo.Wo, geo.Cout, tile, x.device,
                                    (geo.w_numel // geo.Cout + 31) // 32), 1, _conv_bytes(geo, x, w, out)))
    return out


# LOANS_PAIR16=0: the bf16-storage arm launches a unit's first conv and its conv shortcut separately again
PAIR16 = True


def fprop_pair_ok(x, geo_a, geo_b):
    """conv_fprop_pair applies: same input / kernel / stride / padding, no dense rows; fp32 tensors and arithmetic
    (loans_igemm_pair_f32: one grid, one tail) or bf16 storage with equal channel counts (loans_igemm_pair_bf16s: one GEMM
    with the weights stacked along N, the input tile staged once for both)"""
    same = (not geo_a.dense and not geo_b.dense and
            (geo_a.B, geo_a.H, geo_a.W, geo_a.Cin, geo_a.k, geo_a.stride, geo_a.pad) ==
            (geo_b.B, geo_b.H, geo_b.W, geo_b.Cin, geo_b.k, geo_b.stride, geo_b.pad))
    if _is16(x):
        # strided units only: at stride 1 each convolution runs faster alone on the halo-staged tiles (res2: 2 x 0.205 ms
        # against 0.455 ms for the pair at 128 x 3 x 512^2)
        return same and PAIR16 and geo_a.stride > 1 and geo_a.Cout == geo_b.Cout and geo_a.Cout % 32 == 0 and geo_a.Cin % 8 == 0 and \
            2 * geo_a.B * geo_a.Ho * geo_a.Wo * geo_a.Cout * 2 < 0xFFFFFFF0
    return same and COMPUTE == 'f32'


_PAIR_TILES = tuple(t for t in _FPROP_TILES if (t & 15) not in (6, 8, 10))


def conv_fprop_pair(x, w_a, w_b, geo_a, geo_b, stats_a=None, stats_b=None, tile=0):
    """Two forward convolutions of the same input in ONE launch (loans_igemm_pair_f32): BasicA's conv1 and conv shortcut,
    a bottleneck's conv1 and conv4.  Returns (out_a, out_b); statistics for both or for neither."""
    lib = _lib.load()
    assert fprop_pair_ok(x, geo_a, geo_b) and (stats_a is None) == (stats_b is None)
    if _is16(x):
        return _conv_fprop_pair16(lib, x, w_a, w_b, geo_a, geo_b, stats_a, stats_b, tile)
    mk = lambda g: _empty((g.B, g.Ho, g.Wo, g.Cout), device=x.device, dtype=torch.float32)      # noqa: E731
    out_a, out_b = mk(geo_a), mk(geo_b)
    flags = F_STATS if stats_a is not None else 0
    if tile == 0:
        vkey = _variant(geo_a, 'pair', geo_b.Cout, flags)
        tile = geo_a.tuned.get(vkey, 0)
    if tile == 0:
        sa = stats_buffer(geo_a.Cout, x.device) if flags else None
        sb = stats_buffer(geo_b.Cout, x.device) if flags else None
        ta, tb = mk(geo_a), mk(geo_b)

        def run(t):
            check(lib.loans_igemm_pair_f32(_ptr(x), _ptr(w_a), _ptr(ta), _ptr(sa), _ptr(w_b), _ptr(tb), _ptr(sb), geo_b.Cout,
                                           C.byref(_with_flags(geo_a.fwd, flags, t)), _stream()), 'loans_igemm_pair_f32[tune]')
        tile = _tuned_tile(geo_a, 'f32fprop_pair%d%s' % (geo_b.Cout, '_stats' if flags else ''), run, _PAIR_TILES)
        geo_a.tuned[vkey] = tile
    _count_flops('fprop', geo_a)
    _count_flops('fprop', geo_b)
    if CLASS_COUNT is not None:
        rd, wr = _sum_bytes(_conv_bytes(geo_a, x, w_a, out_a), _conv_bytes(geo_b, None, w_b, out_b))
        _acct('conv', _conv_flop(geo_a) + _conv_flop(geo_b), rd, wr)
    log = EVENT_LOG
    if log is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib.loans_igemm_pair_f32(_ptr(x), _ptr(w_a), _ptr(out_a), _ptr(stats_a), _ptr(w_b), _ptr(out_b), _ptr(stats_b),
                                   geo_b.Cout, C.byref(_with_flags(geo_a.fwd, flags, tile)), _stream()), 'loans_igemm_pair_f32')
    if log is not None:
        ev1.record()
        fl = 2 * geo_a.B * geo_a.Ho * geo_a.Wo * (geo_a.Cout + geo_b.Cout) * geo_a.k * geo_a.k * geo_a.cin_logical
        log.append(('fprop_bn' if flags else 'fprop', fl, ev0, ev1, 1, 2,         # one launch, two convolutions
                    _sum_bytes(_conv_bytes(geo_a, x, w_a, out_a), _conv_bytes(geo_b, None, w_b, out_b))))
    return out_a, out_b


_PAIR16_TILES = (1, 2, 3, 7)


def _conv_fprop_pair16(lib, x, w_a, w_b, geo_a, geo_b, stats_a, stats_b, tile):
    """the bf16-storage pair: weights cast into the two halves of one [2][Cout][K] matrix, outputs two views of one allocation"""
    n = geo_a.w_numel
    w_ab = _empty(2 * n, device=x.device, dtype=BF16)
    for i, w in enumerate((w_a, w_b)):
        assert w.numel() == n and not _is16(w)
        check(lib.loans_cast_bf16(_ptr(w), _ptr(w_ab[i * n:]), n, _stream()), 'loans_cast_bf16')
    flags = F_STATS if stats_a is not None else 0
    mk = lambda: _empty((2, geo_a.B, geo_a.Ho, geo_a.Wo, geo_a.Cout), device=x.device, dtype=BF16)      # noqa: E731
    if tile == 0:
        vkey = _variant(geo_a, 'pair16', flags)
        tile = geo_a.tuned.get(vkey, 0)
    if tile == 0:
        sa = stats_buffer(geo_a.Cout, x.device) if flags else None
        sb = stats_buffer(geo_a.Cout, x.device) if flags else None
        scratch = mk()

        def run(t):
            check(lib.loans_igemm_pair_bf16s(_ptr(x), _ptr(w_ab), _ptr(scratch), _ptr(sa), _ptr(sb),
                                             C.byref(_with_flags(geo_a.fwd, flags, t)), _stream()), 'loans_igemm_pair_bf16s[tune]')
        tile = _tuned_tile(geo_a, 'bf16s_fprop_pair' + ('_stats' if flags else ''), run,
                           _PAIR16_TILES + _wide16_tiles(2 * geo_a.Cout, geo_a.B * geo_a.Ho * geo_a.Wo))
        geo_a.tuned[vkey] = tile
    out = mk()
    _count_flops('fprop', geo_a)
    _count_flops('fprop', geo_b)
    if CLASS_COUNT is not None:
        rd, wr = _sum_bytes(_conv_bytes(geo_a, x, w_ab[:n], out[0]), _conv_bytes(geo_b, None, w_ab[n:], out[1]))
        _acct('conv', _conv_flop(geo_a) + _conv_flop(geo_b), rd, wr)
    log = EVENT_LOG
    if log is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib.loans_igemm_pair_bf16s(_ptr(x), _ptr(w_ab), _ptr(out), _ptr(stats_a), _ptr(stats_b),
                                     C.byref(_with_flags(geo_a.fwd, flags, tile)), _stream()), 'loans_igemm_pair_bf16s')
    if log is not None:
        ev1.record()
        fl = 2 * geo_a.B * geo_a.Ho * geo_a.Wo * 2 * geo_a.Cout * geo_a.k * geo_a.k * geo_a.cin_logical
        log.append(('fprop_bn' if flags else 'fprop', fl, ev0, ev1, 1, 2,         # one launch, two convolutions
                    _sum_bytes(_conv_bytes(geo_a, x, w_ab[:n], out[0]), _conv_bytes(geo_b, None, w_ab[n:], out[1]))))
    return out[0], out[1]


def _conv_fprop16(lib, x, w, geo, out, bias, stats, relu_in, addend, tile):
    """bf16-storage forward conv: x / out / addend bf16, w fp32 master weights (cast per call) or already bf16."""
    if out is None:
        out = _empty((geo.B, geo.Ho, geo.Wo, geo.Cout), device=x.device, dtype=BF16)
    assert out.dtype == BF16 and (addend is None or addend.dtype == BF16)
    assert x.numel() == geo.in_numel and w.numel() == geo.w_numel
    w16 = w if _is16(w) else _bf16_shadow(w)            # inside a step: the arena's bf16 shadow, cast whole at its start
    if w16 is None:
        w16 = cast_bf16(w)
    flags = (F_RELU_IN if relu_in else 0) | (F_BIAS if bias is not None else 0) | (F_STATS if stats is not None else 0) | \
            (F_ADDEND if addend is not None else 0) | geo.base_flags
    if tile == 0:
        vkey = _variant(geo, 'fprop16', stats is not None, relu_in, addend is None, bias is not None)
        tile = geo.tuned.get(vkey, 0)
    if tile == 0:
        tflags = flags & (F_STATS | F_RELU_IN | F_DENSE)
        sstats = stats_buffer(geo.Cout, x.device) if stats is not None else None

        def run(t):
            scratch = _empty((geo.B, geo.Ho, geo.Wo, geo.Cout), device=x.device, dtype=BF16)
            if t >> 8:
                _igemm16_splitk(lib, x, [(geo.fwd, w16)], scratch, tflags, t, None, sstats, None, None,
                                geo.B * geo.Ho * geo.Wo, geo.Cout, _stream())
                return
            wt = _pw_packed(lib, w, w16, geo, _stream()) if t == TILE_PW else w16
            check(lib.loans_igemm_bf16s(_ptr(x), _ptr(wt), _ptr(scratch), 0, _ptr(sstats), 0, 0,
                                        C.byref(_with_flags(geo.fwd, tflags, t)), _stream()), 'loans_igemm_bf16s[tune]')
        # where LOANS_TILE_PW is on offer the candidates are timed COLD (a 512 MB fill before every launch): these layers write four
        # times what they read, and a back-to-back repetition flatters the tiles that re-read a cached input -- in the step the
        # 256 x 256 tile takes 0.120 ms on res3's expansion and this one 0.096, timed warm it is 0.107 against 0.110
        pw = _pw_tiles(geo, not relu_in and addend is None and bias is None)
        halo = _halo_tiles(geo, geo.Cin, geo.Cout, (geo.Ho, geo.Wo), relu_in) + _wide16_tiles(geo.Cout, geo.B * geo.Ho * geo.Wo)
        sk = () if geo.dense else _splitk16_candidates(geo.B * geo.Ho * geo.Wo, geo.Cout, geo.w_numel // geo.Cout)
        stem = (TILE_STEM,) if (STEM_DIRECT and geo.dense and not relu_in and addend is None and stem16_tile_rows(geo)) else ()
        # (relu_in is part of the key: the weight-stationary tiles do not take it, so a tile tuned without it may not apply)
        tile = _tuned_tile(geo, 'bf16s_fprop' + ('_stats' if stats is not None else '') + ('_h' if halo else '') +
                           ('_sk' if sk else '') + ('_st' if stem else '') + ('_relu' if relu_in else '') + ('_pw' if pw else ''), run,
                           _IGEMM16_TILES + halo + sk + stem + pw, cold=bool(pw))
        geo.tuned[vkey] = tile
    d = _with_flags(geo.fwd, flags, tile)
    _count_flops('fprop', geo)
    if CLASS_COUNT is not None:
        rd, wr = _conv_bytes(geo, x, w16, out)
        _acct('conv', _conv_flop(geo), rd + _nbytes(addend), wr)
    log = EVENT_LOG
    if log is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if tile >> 8:
        _igemm16_splitk(lib, x, [(geo.fwd, w16)], out, flags, tile, bias, stats, None, addend,
                        geo.B * geo.Ho * geo.Wo, geo.Cout, _stream())
    else:
        if tile == TILE_PW:
            packs = PW_PACK_CALLS
            w16 = _pw_packed(lib, w, w16, geo, _stream())
            packs = PW_PACK_CALLS - packs           # 1 outside a prepared step (the packing launch), else 0
        check(lib.loans_igemm_bf16s(_ptr(x), _ptr(w16), _ptr(out), _ptr(bias), _ptr(stats), 0, _ptr(addend),
                                    C.byref(d), _stream()), 'loans_igemm_bf16s[fprop]')
    if log is not None:
        ev1.record()
        log.append(('fprop_bn' if stats is not None else 'fprop',
                    2 * geo.B * geo.Ho * geo.Wo * geo.Cout * geo.k * geo.k * geo.cin_logical, ev0, ev1,
                    2 if tile >> 8 else (1 + packs if tile == TILE_PW else 1), 1,
                    _conv_bytes(geo, x, w16, out)))          # split-K: the partial launch and the finalize pass (the memset is torch's);
        #                                                      LOANS_TILE_PW outside a prepared step: the packing launch and the convolution
    return out


def _conv_dgrad16(lib, gy, w, geo, out, mask_ref, addend, addend_mask_ref, tile, bn_sums=None):
    if out is None:
        out = _empty((geo.B, geo.H, geo.W, geo.Cin), device=gy.device, dtype=BF16)
    assert out.dtype == BF16 and not _is16(w)
    if geo.dgrad_has_empty_class:
        assert mask_ref is None and addend_mask_ref is None
        if addend is None:
            out.zero_()
        elif addend.data_ptr() != out.data_ptr():
            out.copy_(addend)
        if addend is not None:
            addend = out
    flags = (F_MASK if mask_ref is not None else 0) | (F_ADDEND if addend is not None else 0) | \
            (F_ADDEND_MASK if addend_mask_ref is not None else 0)
    ref = mask_ref if mask_ref is not None else addend_mask_ref
    assert not (mask_ref is not None and addend_mask_ref is not None)
    coef = sums = None
    if bn_sums is not None:             # LOANS_F_BNSUMS: the epilogue takes the two sums of the BN below (see conv_dgrad)
        assert flags == 0 and len(geo.dgrad) == 1 and not geo.dgrad_has_empty_class
        y, bst = bn_sums
        assert _is16(y) and y.is_contiguous() and y.numel() == out.numel()
        flags, ref, coef, sums = F_BNSUMS, y, bst.mean, stats_buffer(geo.Cin, gy.device)
    st = _stream()
    wp = _prepacked_dgrad_weights(w, geo, True)          # inside a step: made at its start, all layers in one launch
    if wp is None:
        wp = _empty(geo.dgrad_weight_floats, device=gy.device, dtype=BF16)
        for d, tapsel, off in geo.dgrad:
            check(lib.loans_repack_dgrad_bf16(_ptr(w), _ptr(wp[off:]), geo.Cout, geo.Cin, geo.k * geo.k, tapsel,
                                              d.ntaps, st), 'loans_repack_dgrad_bf16')
    dl = [(d, wp[off:]) for d, _, off in geo.dgrad]
    rows_in = geo.B * geo.H * geo.W
    # a split-K data gradient finishes in ONE pass over the whole tensor: the tap-less classes of a strided 1x1 (which only
    # take the addend, handled above by a copy) and an addend aliasing `out` do not go through it
    sk_ok = not geo.dgrad_has_empty_class and not (addend is not None and addend.data_ptr() == out.data_ptr())
    if tile == 0:
        vkey = _variant(geo, 'dgrad16', sk_ok, bn_sums is not None)
        tile = geo.tuned.get(vkey, 0)
    if tile == 0:
        def run(t):
            scratch = _empty((geo.B, geo.H, geo.W, geo.Cin), device=gy.device, dtype=BF16)
            if t >> 8:
                _igemm16_splitk(lib, gy, dl, scratch, 0, t, None, None, None, None, rows_in, geo.Cin, st)
                return
            for d, _, off in geo.dgrad:
                check(lib.loans_igemm_bf16s(_ptr(gy), _ptr(wp[off:]), _ptr(scratch), 0, 0, 0, 0,
                                            C.byref(_with_flags(d, 0, t)), st), 'loans_igemm_bf16s[tune]')
        halo = _halo_tiles(geo, geo.Cout, geo.Cin, (geo.H, geo.W)) + _wide16_tiles(geo.Cin, geo.B * min(d.gridH * d.gridW for d, _, _ in geo.dgrad))
        sk = ()
        if sk_ok and bn_sums is None:
            cls_rows = geo.B * min(d.gridH * d.gridW for d, _, _ in geo.dgrad)
            sk = _splitk16_candidates(cls_rows, geo.Cin, min(d.ntaps for d, _, _ in geo.dgrad) * geo.Cout)
        if bn_sums is not None:         # ws8_kernel's epilogue does not take the sums
            halo = tuple(t for t in halo if t != TILE_WS64)
        tile = _tuned_tile(geo, 'bf16s_dgrad' + ('_h' if halo else '') + ('_sk' if sk else '') + ('_bn' if bn_sums is not None else ''),
                           run, _IGEMM16_TILES + halo + sk)
        geo.tuned[vkey] = tile
    if tile >> 8:
        assert sk_ok and bn_sums is None
        _igemm16_splitk(lib, gy, dl, out, flags, tile, None, None, ref, addend, rows_in, geo.Cin, st)
        return out
    for d, tapsel, off in geo.dgrad:
        _with_flags(d, flags, tile)
        check(lib.loans_igemm_bf16s(_ptr(gy), _ptr(wp[off:]), _ptr(out), _ptr(coef), _ptr(sums), _ptr(ref), _ptr(addend),
                                    C.byref(d), st), 'loans_igemm_bf16s[dgrad]')
    return out if bn_sums is None else (out, sums)




def bn_sums_ok(geo, y):
    """conv_dgrad(..., bn_sums=) covers this data gradient: one stride-parity class (stride 1), a gradient tensor the
    row-reduction layout tiles, operands in one storage type"""
    return len(geo.dgrad) == 1 and not geo.dgrad_has_empty_class and not geo.dense and geo.Cin != 4 and \
        geo.Cin % 8 == 0 and y.is_contiguous()


def conv_dgrad(gy, w, geo, out=None, mask_ref=None, addend=None, addend_mask_ref=None, tile=0, bn_sums=None):
    """gx[B,H,W,Cin] = conv_transpose(gy, w); epilogue: (* (mask_ref>0)), (+ addend [masked by addend_mask_ref>0]).
    Re-packs w per stride-parity class first (weights change every step).
    bn_sums=(y, BNState): gx is the gradient that reaches a BatchNormalization (input y, batch coefficients in the BNState)
    followed by its own ReLU; the epilogue also takes that BN's two backward sums from the tile (LOANS_F_BNSUMS) and the call
    returns (gx, sums) with sums = fp64 [replicas][2][C] for bn_backward_from_sums -- the reduction pass over gx disappears."""
    lib = _lib.load()
    assert gy.numel() == geo.B * geo.Ho * geo.Wo * geo.Cout
    _count_flops('dgrad', geo)
    if CLASS_COUNT is not None:
        es = 2 if (_is16(gy) and geo.Cin != 4) else 4
        _acct('crop' if geo.Cin == 4 else 'conv', _conv_flop(geo),
              _nbytes(gy, w, mask_ref, addend, addend_mask_ref, bn_sums[0] if bn_sums is not None else None),
              geo.B * geo.H * geo.W * geo.Cin * es)
    if _is16(gy) and geo.Cin != 4:
        return _conv_dgrad16(lib, gy, w, geo, out, mask_ref, addend, addend_mask_ref, tile, bn_sums)
    if out is None:
        out = _empty((geo.B, geo.H, geo.W, geo.Cin), device=gy.device, dtype=torch.float32)
    if geo.dgrad_has_empty_class:
        # pixels of a tap-less stride class receive only the addend (or zero)
        assert mask_ref is None and addend_mask_ref is None
        if addend is None:
            out.zero_()
        elif addend.data_ptr() != out.data_ptr():
            out.copy_(addend)
        if addend is not None:
            addend = out
    if geo.Cin == 4 and addend_mask_ref is None and geo.Cout % 32 == 0:
        # gradient w.r.t. a 4-channel (RGB) input: dedicated VALU kernel, forward weights, no re-pack; the crops and
        # their gradient stay fp32 in every arm, the incoming gradient may be a bf16 tensor
        fl = (F_MASK if mask_ref is not None else 0) | (F_ADDEND if addend is not None else 0)
        fn = lib.loans_dgrad_c4_bf16_f32 if _is16(gy) else lib.loans_dgrad_c4_f32
        if out is None:
            out = _empty((geo.B, geo.H, geo.W, geo.Cin), device=gy.device, dtype=torch.float32)
        for d, tapsel, _ in geo.dgrad:
            check(fn(_ptr(gy), _ptr(w), _ptr(out), _ptr(mask_ref), _ptr(addend),
                     C.byref(_with_flags(d, fl, 0)), tapsel, geo.k * geo.k, _stream()), 'loans_dgrad_c4')
        return out
    flags = (F_MASK if mask_ref is not None else 0) | (F_ADDEND if addend is not None else 0) | \
            (F_ADDEND_MASK if addend_mask_ref is not None else 0)
    ref = mask_ref if mask_ref is not None else addend_mask_ref
    assert not (mask_ref is not None and addend_mask_ref is not None)
    coef = sums = None
    if bn_sums is not None:
        assert flags == 0 and len(geo.dgrad) == 1 and not geo.dgrad_has_empty_class
        y, bst = bn_sums
        assert not _is16(y) and y.is_contiguous() and y.numel() == out.numel()
        flags, ref, coef, sums = F_BNSUMS, y, bst.mean, stats_buffer(geo.Cin, gy.device)
    st = _stream()
    wp = _prepacked_dgrad_weights(w, geo, False)         # inside a step: made at its start, all layers in one launch
    if wp is None:
        wp = _empty(geo.dgrad_weight_floats, device=gy.device, dtype=torch.float32)
        for d, tapsel, off in geo.dgrad:
            check(lib.loans_repack_dgrad_f32(_ptr(w), _ptr(wp[off:]), geo.Cout, geo.Cin, geo.k * geo.k, tapsel,
                                             d.ntaps, st), 'loans_repack_dgrad_f32')
    # split-K adds raw partial sums: a masked conv term cannot land on an addend that already sits in `out`
    inplace_masked = addend is not None and addend.data_ptr() == out.data_ptr() and mask_ref is not None
    dl = [(d, wp[off:]) for d, _, off in geo.dgrad]
    rows_in = geo.B * geo.H * geo.W
    if tile == 0:
        vkey = _variant(geo, 'dgrad', bn_sums is not None, inplace_masked)
        tile = geo.tuned.get(vkey, 0)
    if tile == 0:
        def run(t):
            scratch = _empty((geo.B, geo.H, geo.W, geo.Cin), device=gy.device, dtype=torch.float32)
            if t & TILE_CLASSES:
                _igemm_classes(lib, gy, wp, scratch, geo, 0, t & 0xFF, None, None, st)
                return
            if t >> 8:
                _igemm_splitk(lib, gy, None, scratch, dl, 0, t, None, None, None, None, rows_in, geo.Cin, st)
                return
            for d, _, off in geo.dgrad:
                check(_igemm_fn(lib)(_ptr(gy), _ptr(wp[off:]), _ptr(scratch), 0, 0, 0, 0,
                                     C.byref(_with_flags(d, 0, t)), st), 'loans_igemm[tune]')
        cl = _class_candidates(geo)
        cands, key = _IGEMM_TILES + cl, COMPUTE + 'dgrad' + ('_cl' if cl else '')
        if bn_sums is not None:
            key += '_bn'
        elif not inplace_masked and reduce_channels_ok(geo.Cin):
            # the smallest class grid decides: (Ho x Wo pixels of one parity class) x Cin columns, K = its taps x Cout
            cls_rows = geo.B * geo.dgrad[0][0].gridH * geo.dgrad[0][0].gridW
            sk = _splitk_candidates(cls_rows, geo.Cin, (min(d.ntaps for d, _, _ in geo.dgrad) * geo.Cout + 31) // 32)
            cands, key = cands + sk, key + ('_sk' if sk else '')
        tile = _tuned_tile(geo, key, run, cands)
        geo.tuned[vkey] = tile
    if tile & TILE_CLASSES:
        _igemm_classes(lib, gy, wp, out, geo, flags, tile & 0xFF, ref, addend, st)
        return out
    if tile >> 8:
        _igemm_splitk(lib, gy, None, out, dl, flags, tile, None, None, ref, addend, rows_in, geo.Cin, st)
        return out
    for d, tapsel, off in geo.dgrad:
        _with_flags(d, flags, tile)
        check(_igemm_fn(lib)(_ptr(gy), _ptr(wp[off:]), _ptr(out), _ptr(coef), _ptr(sums), _ptr(ref), _ptr(addend),
                             C.byref(d), st), 'loans_igemm[dgrad]')
    return out if bn_sums is None else (out, sums)


# tile id bit (host side only): every stride-parity class of a strided data gradient in ONE launch (loans_igemm_classes_f32)
TILE_CLASSES = 1 << 16
CLASS_LAUNCH = True


@_memo
def _class_candidates(geo):
    if not CLASS_LAUNCH or COMPUTE != 'f32' or not 2 <= len(geo.dgrad) <= 4 or max(d.ntaps for d, _, _ in geo.dgrad) > 16:
        return ()
    return tuple(t | TILE_CLASSES for t in _IGEMM_TILES if (t & 15) in (1, 2, 3, 4))


def _igemm_classes(lib, gy, wp, out, geo, flags, tile, ref, addend, st):
    """classes in descending K: the long blocks start first, the 1-tap class fills the tail"""
    order = sorted(geo.dgrad, key=lambda e: -e[0].ntaps)
    n = len(order)
    descs = (_lib.IgemmDesc * n)()
    ws = (C.c_void_p * n)()
    for i, (d, _, off) in enumerate(order):
        C.memmove(C.byref(descs[i]), C.byref(_with_flags(d, flags, tile)), C.sizeof(_lib.IgemmDesc))
        ws[i] = _ptr(wp[off:])
    check(lib.loans_igemm_classes_f32(_ptr(gy), ws, _ptr(out), _ptr(ref), _ptr(addend), descs, n, st),
          'loans_igemm_classes_f32')


# LOANS_CROP_DGRAD=0: the gradient w.r.t. the 4-channel crops goes back to one loans_dgrad_c4 launch per stride-parity class
CROP_DGRAD = True


def crop_dgrad_ok(geo_a, geo_b=None, gy_a=None, gy_b=None):
    """loans_crop_dgrad covers these convolutions of a 4-channel input (csrc/cropgrad.hip) and -- when the gradient tensors
    are given -- these operands: contiguous, one storage type for both (a caller falls back to conv_dgrad otherwise)"""
    ok = CROP_DGRAD
    for gy, g in ((gy_a, geo_a), (gy_b, geo_b)):
        if gy is not None:
            ok = ok and g is not None and gy.is_contiguous() and gy.dtype in (torch.float32, BF16) \
                and gy.numel() == g.B * g.Ho * g.Wo * g.Cout
    if gy_a is not None and gy_b is not None:
        ok = ok and gy_a.dtype == gy_b.dtype
    for g in (geo_a, geo_b):
        if g is None:
            continue
        ok = ok and g.Cin == 4 and not g.dense and g.k <= 4 and g.stride <= 2 and g.pad < g.k and g.Cout == 128 \
            and g.B * g.Ho * g.Wo * g.Cout * 4 < (1 << 31)
    if geo_b is not None:
        ok = ok and (geo_a.B, geo_a.H, geo_a.W, geo_a.Cout) == (geo_b.B, geo_b.H, geo_b.W, geo_b.Cout)
    return ok


_crop_wpacks = {}


def _crop_wpack(device):
    """the kernel's fragment-order weight workspace (LOANS_CROP_WPACK_FLOATS), one per (device, stream): re-packed by every
    call's pre-pass and consumed by the same call on the same stream, so calls on one stream may share it"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    w = _crop_wpacks.get(key)
    if w is None:
        w = _crop_wpacks[key] = torch.empty(16384, device=device, dtype=torch.float32)
    return w


def crop_dgrad(gy_a, w_a, geo_a, gy_b=None, w_b=None, geo_b=None, addend=None):
    """gx[B,H,W,4] = dgrad(gy_a, w_a) (+ dgrad(gy_b, w_b)) (+ addend) for convolutions of the SAME 4-channel input, in one
    launch that reads each gradient tensor once; weights are the forward OHWI tensors.  fp32 or bf16 gradients, fp32 gx."""
    lib = _lib.load()
    assert crop_dgrad_ok(geo_a, geo_b, gy_a, gy_b) and (gy_b is None) == (geo_b is None)
    for gy, g in ((gy_a, geo_a), (gy_b, geo_b)):
        if gy is not None:
            _count_flops('dgrad', g)
            if CLASS_COUNT is not None: _acct('crop', _conv_flop(g), _nbytes(gy))
    out = _empty((geo_a.B, geo_a.H, geo_a.W, 4), device=gy_a.device, dtype=torch.float32)
    if CLASS_COUNT is not None: _acct('crop', 0, _nbytes(addend), _nbytes(out))
    mk = lambda g: _lib.SmallConv(g.k, g.stride, g.pad, g.Ho, g.Wo)      # noqa: E731
    ca, cb = mk(geo_a), (mk(geo_b) if geo_b is not None else None)
    fn = lib.loans_crop_dgrad_bf16_f32 if _is16(gy_a) else lib.loans_crop_dgrad_f32
    wpack = _crop_wpack(gy_a.device)
    check(fn(_ptr(gy_a), _ptr(w_a), C.byref(ca), _ptr(gy_b), _ptr(w_b), C.byref(cb) if cb is not None else None,
             _ptr(out), _ptr(addend), _ptr(wpack), geo_a.B, geo_a.H, geo_a.W, geo_a.Cout, _stream()), 'loans_crop_dgrad')
    return out


# Weight gradients are consumed only by the optimiser, so they run on a second HIP stream: the
# MFMA-bound wgrad kernels overlap the HBM-bound BN / ReLU passes of the data-gradient chain and
# fill the tails of its dgrad launches.  `join_side_stream()` is the barrier the consumers call.
ASYNC_WGRAD = True
# while a step is being recorded into a hipGraph the side streams fork from / join the capturing stream, so the graph keeps
# the eager step's concurrency (weight gradients beside the data-gradient chain, the assessor's chain beside the localizer's)
CAPTURE_STREAMS = True
_side = {}
_side_dirty = set()
# (Tried and removed, DESIGN 7c / 7d: HIP stream priorities for the side streams -- no effect in any combination --, and a CU mask
# for the weight-gradient stream (hipExtStreamCreateWithCUMask): configs[2] 18.9 -> 25.2-26.0 ms, a masked queue changes how the
# whole device dispatches.)


def side_stream_cus(device=None):
    """CUs the weight-gradient streams may use (block-count candidates of the weight gradients are laid out for these)"""
    return torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else device).multi_processor_count


def _new_side_stream(device):
    return torch.cuda.Stream(device=device)


# Consecutive weight gradients are independent of each other: with LOANS_WGRAD_STREAMS = n > 1 they rotate over n streams, so that
# the tail of one launch (its last, partly empty round of tiles; its closing float atomics) runs under the main loop of the
# next.  Measured (round 3, same box, two pairs): fp32 configs[1] 50.24 -> 49.71 ms with two streams -- the fp32 weight gradients
# run 3-6 rounds of 128 x 128 tiles with a ragged last one --, bf16 configs[2] / ResNet-50 +0.06 / +0.15 ms (their launches are
# one or two rounds of long blocks that already fill the machine; a second launch beside them only takes CUs from the first).
# Default: two streams while the fp32 kernels are selected, one on the bf16 arms.
_WGRAD_STREAMS_ENV = ''          # (tests set it: '1' .. '4' overrides the default below)
_side_more = {}
_side_turn = {}


def wgrad_streams():
    if _WGRAD_STREAMS_ENV:
        return max(1, min(4, int(_WGRAD_STREAMS_ENV)))
    return 2 if COMPUTE == 'f32' else 1


def _side_stream(device):
    """the weight-gradient stream; with several: the FIRST, after it has been made to wait for the others -- what callers that
    order something behind every weight gradient issued so far (the staged exchange) rely on"""
    st = _side.get(device.index)
    if st is None:
        st = _side[device.index] = _new_side_stream(device)
    for other in _side_more.get(device.index, ()):
        st.wait_stream(other)
    return st


def _wgrad_stream(device):
    n = wgrad_streams()
    idx = device.index
    if n == 1 and idx not in _side_more:
        return _side_stream(device)
    if idx not in _side:
        _side[idx] = _new_side_stream(device)
    more = _side_more.setdefault(idx, [])
    while len(more) < n - 1:
        more.append(_new_side_stream(device))
    turn = _side_turn[idx] = (_side_turn.get(idx, 0) + 1) % n
    return _side[idx] if turn == 0 else more[turn - 1]


def join_side_stream(device=None):
    """Make the current stream wait for every weight-gradient kernel issued so far."""
    for idx in list(_side_dirty):
        if device is not None and device.index != idx:
            continue
        cur = torch.cuda.current_stream(idx)
        cur.wait_stream(_side[idx])
        for other in _side_more.get(idx, ()):
            cur.wait_stream(other)
        _side_dirty.discard(idx)


def _wgrad_key(x, gy, relu_in):
    """autotune table key of a weight-gradient problem: arm, operand storage, relu(x) or not"""
    return ('bf16s_' if _is16(x) else COMPUTE) + ('g16' if _is16(gy) and not _is16(x) else '') + \
           ('relu_' if relu_in else '') + 'wgrad'


def conv_wgrad(x, gy, dw, geo, relu_in=False, splits=0, tile=0, in_affine=None):
    """dw[Cout,k,k,Cin] += sum over pixels (atomic accumulate into the gradient arena).  in_affine (a BNState; affine_in_ok): x is
    the INPUT of the BN in front of the convolution, the kernel contracts gy with relu(x * scale + shift) rounded to bf16."""
    key = _wgrad_key(x, gy, relu_in)
    tuned = geo.tuned.get(key) is not None or geo.tuned.get(key + '_st') is not None      # ('_st': the stem's candidates, _conv_wgrad)
    if ASYNC_WGRAD and tuned and (CAPTURE_STREAMS or not torch.cuda.is_current_stream_capturing()):
        side = _wgrad_stream(x.device)
        side.wait_stream(_current_stream_obj(x.device.index))
        x.record_stream(side)
        gy.record_stream(side)
        _side_dirty.add(x.device.index)
        if not geo.dense:
            # the launch goes to the side stream by handle: making it torch's current stream (a context manager: two Python-level
            # stream switches per weight gradient) cost 2 ms of host time per ResNet-50 step; nothing is allocated on this path
            _conv_wgrad(x, gy, dw, geo, relu_in, splits, tile, stream=side.cuda_stream, in_affine=in_affine)
        else:
            with torch.cuda.stream(side):       # (the dense stem's mask pass allocates)
                _conv_wgrad(x, gy, dw, geo, relu_in, splits, tile, in_affine=in_affine)
        return
    _conv_wgrad(x, gy, dw, geo, relu_in, splits, tile, in_affine=in_affine)


_stream_objs = {}


def _current_stream_obj(idx):
    """torch's Stream object of the current stream, looked up by raw handle (torch.cuda.current_stream() resolves the device in
    Python and builds a new object every time)"""
    raw = _raw_stream(idx)
    st = _stream_objs.get((idx, raw))
    if st is None:
        st = _stream_objs[(idx, raw)] = torch.cuda.current_stream(idx)
    return st


_WGRAD_TILE_DIMS = {1: (128, 128, 2), 3: (64, 64, 4), 5: (64, 128, 3)}      # tile id -> (BCO, BJ, blocks per CU by LDS)
_WGRAD16_TILE_DIMS = {1: (128, 128, 3), 3: (64, 64, 5), 5: (64, 128, 4), 9: (256, 256, 1)}    # bf16 tiles: smaller LDS images, register-bound


@_memo
def _wgrad_candidates(geo, tiles, chunk_px, dims=None):
    """(tile, splits) candidates of a weight gradient, encoded tile | splits << 8 (0 = the library's default).  The
    reduction over pixels is cut into `splits` slices per output tile; how many blocks that makes against the machine's
    block slots decides the tail (res4: 36 tiles x 29 slices = 2.04 rounds of 512 slots runs at 96 TFLOP/s, x 14 = 0.98
    rounds at 118), so slot-aligned counts are offered beside the default and its half / double."""
    K = geo.w_numel // geo.Cout
    chunks = (geo.B * geo.Ho * geo.Wo + chunk_px - 1) // chunk_px
    cus = side_stream_cus()
    out = []
    for t in tiles:
        bco, bj, per_cu = (dims or _WGRAD_TILE_DIMS)[t]
        ntile = ((geo.Cout + bco - 1) // bco) * ((K + bj - 1) // bj)
        auto = max(1, min((1024 + ntile - 1) // ntile, (chunks + 7) // 8))
        cand = {0, max(1, auto // 2), auto * 2}
        for rounds in (1, 2, 3, 4):
            cand.add(max(1, rounds * per_cu * cus // ntile))
        for sp in sorted(cand):
            if sp == 0 or (sp != auto and chunks // sp >= 4):
                out.append(t | (sp << 8))
    return tuple(out)


# LOANS_TILE_WGHALO_* (csrc/wgrad_halo_bf16.hip): all nine taps of a stride-1 3x3 weight gradient in one block
TILE_WGHALO_64, TILE_WGHALO_128 = 38, 39
WGHALO = True


@_memo
def wghalo_tiles(geo):
    """the halo weight-gradient tiles that cover this geometry (loans_wgrad_halo16_covers), where an 8 x 16 pixel tile is not
    mostly empty"""
    if not WGHALO or geo.dense or geo.k != 3 or geo.stride != 1 or geo.Cin % 64 or geo.Cout % 64 or (geo.Ho, geo.Wo) != (geo.H, geo.W):
        return ()
    if geo.H < 6 or geo.W < 12:
        return ()
    return (TILE_WGHALO_64,) + ((TILE_WGHALO_128,) if geo.Cout % 128 == 0 else ())


@_memo
def _wghalo_candidates(geo):
    """(tile | blocks per channel-tile pair << 8): whole rounds of the machine's block slots (two 4-wave blocks or one 8-wave
    block per CU), 0 = the library's default"""
    cus = side_stream_cus()
    ntiles = geo.B * ((geo.H + 7) // 8) * ((geo.W + 15) // 16)
    out = []
    for t in wghalo_tiles(geo):
        bco, per_cu = (64, 2) if t == TILE_WGHALO_64 else (128, 1)
        npairs = (geo.Cout // bco) * (geo.Cin // 64)
        cand = {0}
        for rounds in (1, 2, 3, 4, 6):
            sp = max(1, rounds * per_cu * cus // npairs)
            if ntiles // sp >= 2:
                cand.add(sp)
        out += [t | (sp << 8) for sp in sorted(cand)]
    return tuple(out)


# Weight gradients of the bf16-storage arm without atomics (round 5): the blocks store raw partial tiles into slabs of a
# workspace and one deterministic pass folds them into the gradient arena (loans_wgrad_bf16s_ws).  One workspace per (device,
# stream), grown to the largest request: launches on one stream run one after the other and may share it.
# LOANS_WGRAD_SLABS=0: fp32 atomics into the arena again (the order of the sums then changes from run to run).
WGRAD_SLABS = os.environ.get('LOANS_WGRAD_SLABS', '1') != '0'
# The tuned number of pixel slices of a bf16 weight gradient x this.  The tile tables are timed with the launch ALONE on the
# device, where one or two full rounds of block slots win; in the step the weight gradients run beside the main stream, and
# three quarters of that leave compute units to its kernels: configs[2] 18.975 -> 18.830 ms, ResNet-50 29.574 -> 29.491
# (0.5: 18.94 / 29.43; profiles/r5_wgrad_split_scale_ab.txt, tools/ab_wgrad_scale.py)
WGRAD_SPLIT_SCALE = 0.75
WGRAD_SPLIT_SCALE_F32 = 1.0       # the fp32 arm is MFMA-bound on both streams: 0.75 -> 49.72 ms against 49.67, 0.5 -> 50.08 (same file)
_wgrad_ws = {}
_wgrad_ws_captured = []      # workspaces handed out while a hipGraph was being captured: the graph has their addresses baked in,
#                              so they stay alive for the life of the process even after a larger one replaces them


def _wgrad_workspace(lib, geo, desc, tile, splits, device, st):
    """(tensor, floats needed) for loans_wgrad_bf16s_ws on stream handle `st`, or None where it does not apply (a request the
    library rejects; a workspace that would have to grow while a hipGraph is being captured).  A later eager step with a larger
    need replaces the slot's workspace; one a capture has seen is kept in `_wgrad_ws_captured`, so a replayed graph never stores
    its slabs into memory that went back to the allocator."""
    plan = geo.__dict__.setdefault('_ws_need', {})
    need = plan.get((tile, splits))
    if need is None:
        need = plan[(tile, splits)] = int(lib.loans_wgrad_bf16s_ws_floats(C.byref(desc), splits))
    if need <= 0:
        return None
    key = (device.index, st)
    ws = _wgrad_ws.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ws is None or ws.numel() < need:
        if capturing:
            return None
        ws = _wgrad_ws[key] = torch.empty(max(need, 1 << 22), device=device, dtype=torch.float32)
        if st != _stream():
            ws.record_stream(torch.cuda.ExternalStream(st, device=device))      # allocated on the current stream, used on `st`
    elif capturing and not any(b is ws for b in _wgrad_ws_captured):
        _wgrad_ws_captured.append(ws)
    return ws, need


def _conv_wgrad(x, gy, dw, geo, relu_in, splits, tile, stream=None, in_affine=None):
    lib = _lib.load()
    _count_flops('wgrad', geo)
    if CLASS_COUNT is not None: _acct('wgrad', _conv_flop(geo), _nbytes(x, gy), _nbytes(dw))
    assert dw.numel() == geo.w_numel and x.numel() == geo.in_numel
    fl = (F_RELU_IN if relu_in else 0) | geo.base_flags
    s16 = _is16(x)
    if _is16(gy) and not s16:           # the stem of the bf16-storage arm: fp32 frames, bf16 gradient
        assert COMPUTE == 'bf16'
        fl |= F_GY_BF16
    else:
        assert _is16(gy) == s16, 'x and gy must share their storage type'
    wfn = lib.loans_wgrad_bf16s if s16 else (lib.loans_wgrad_bf16_f32 if COMPUTE == 'bf16' else lib.loans_wgrad_f32)
    if tile == 0:
        vkey = _variant(geo, 'wgrad', s16, _is16(gy), relu_in, splits)
        tile = geo.tuned.get(vkey, 0)
    if tile == 0:
        def run(t):
            scratch = _empty(dw.numel(), device=x.device, dtype=torch.float32)
            dt = _with_flags(geo.fwd, fl, t & 0xFF)
            ws = _wgrad_workspace(lib, geo, dt, t & 0xFF, (t >> 8) or splits, x.device, _stream()) if (s16 and WGRAD_SLABS) else None
            if ws is not None:
                check(lib.loans_wgrad_bf16s_ws(_ptr(x), _ptr(gy), _ptr(scratch), C.byref(dt), (t >> 8) or splits, _ptr(ws[0]), ws[1],
                                               _stream()), 'loans_wgrad_bf16s_ws[tune]')
                return
            check(wfn(_ptr(x), _ptr(gy), _ptr(scratch), C.byref(dt), (t >> 8) or splits,
                      _stream()), 'loans_wgrad[tune]')
        cands = _WGRAD16_TILES if s16 else _WGRAD_TILES
        if s16 and geo.Cout % 256 == 0:
            cands = cands + (TILE_256x256,)         # one 512-thread block per CU (csrc/igemm_bf16.hip, wgrad16_kernel<256, 256, 8>)
        if splits == 0:
            cands = _wgrad_candidates(geo, cands, 32, _WGRAD16_TILE_DIMS if s16 else None)
            if s16:
                cands = tuple(cands) + _wghalo_candidates(geo)
        stem = fl == F_DENSE and ((stem_wgrad_ok(geo) and wfn is lib.loans_wgrad_f32) or (s16 and stem16_wgrad_ok(geo)))
        if stem:
            cands = tuple(cands) + (TILE_STEM,)
        tile = _tuned_tile(geo, _wgrad_key(x, gy, relu_in) + ('' if splits == 0 else '_s%d' % splits) + ('_st' if stem else ''),
                           run, cands)
        geo.tuned[vkey] = tile
    if tile >> 8:
        tile, splits = tile & 0xFF, tile >> 8
        scale = WGRAD_SPLIT_SCALE if s16 else WGRAD_SPLIT_SCALE_F32
        if scale != 1.0:
            splits = max(1, int(splits * scale))
    if in_affine is not None:           # x is the input of the BN in front of the convolution: relu(bn(x)) applied on load
        assert s16 and WGRAD_SLABS and fl == 0
        fl = F_AFFINE_IN
    d = _with_flags(geo.fwd, fl, tile)
    st = stream if stream is not None else _stream()
    ws = _wgrad_workspace(lib, geo, d, tile, splits, x.device, st) if (s16 and WGRAD_SLABS) else None
    if in_affine is not None:
        if ws is None:          # (the workspace planner refused the descriptor: not a 1 x 1 / 1 convolution on a GEMM tile)
            raise RuntimeError('loans_wgrad_bf16s_affine_ws does not cover this convolution')
        check(lib.loans_wgrad_bf16s_affine_ws(_ptr(x), _ptr(gy), _ptr(dw), C.byref(d), splits, _ptr(ws[0]), ws[1], _ptr(in_affine.affine()),
                                              st), 'loans_wgrad_bf16s_affine_ws')
    elif ws is not None:
        check(lib.loans_wgrad_bf16s_ws(_ptr(x), _ptr(gy), _ptr(dw), C.byref(d), splits, _ptr(ws[0]), ws[1], st), 'loans_wgrad_bf16s_ws')
    else:
        check(wfn(_ptr(x), _ptr(gy), _ptr(dw), C.byref(d), splits, st), 'loans_wgrad')
    if geo.dense and tile != TILE_STEM:         # the direct kernel never writes those columns
        # the window-padding columns of the dense layout saw real pixels: their "gradient" is not one
        check(lib.loans_mul_f32(_ptr(dw), _ptr(geo.wmask(x.device)), _ptr(dw), dw.numel(), st), 'loans_mul_f32')


# --------------------------------------------------------------------------- #
# preprocessing / layout
# --------------------------------------------------------------------------- #
# LOANS_DENSE_BF16=0: the stem of the bf16 storage arm reads fp32 frames (loans_igemm_bf16_f32 + LOANS_F_OUT_BF16)
DENSE_BF16 = True


def prep_images(images_nchw, geo=None):
    """NHWC4 frames, or -- for a dense-row stem geometry -- the zero-padded packed-RGB buffer [B][Hp][Wp][3]
    (tagged with the frame size, which the padded shape alone does not determine)."""
    B, c, H, W = images_nchw.shape
    assert c == 3
    _chk(images_nchw, 'images')
    if CLASS_COUNT is not None: _acct('heads', 0, _nbytes(images_nchw), B * H * W * 3 * (2 if (STORAGE == 'bf16' and DENSE_BF16) else 4))
    if geo is not None and geo.dense:
        assert (geo.B, geo.H, geo.W) == (B, H, W)
        # bf16 storage arm: the frames leave this kernel as bf16 (the same rounding the bf16 arm applies to fp32 frames
        # while it stages them) and conv1 / its weight gradient run on the bf16-storage kernels
        s16 = STORAGE == 'bf16' and DENSE_BF16 and geo.stride % 2 == 0     # odd pixel steps would misalign the K units
        out = _empty((B, geo.Hp, geo.Wp, 3), device=images_nchw.device, dtype=BF16 if s16 else torch.float32)
        fn = _lib.load().loans_prep_images_dense_bf16 if s16 else _lib.load().loans_prep_images_dense_f32
        check(fn(_ptr(images_nchw), _ptr(out), B, H, W, geo.pad, geo.Hp, geo.Wp, _stream()), 'loans_prep_images_dense')
        out.frame_hw = (H, W)
        return out
    out = _empty((B, H, W, 4), device=images_nchw.device, dtype=torch.float32)
    check(_lib.load().loans_prep_images_f32(_ptr(images_nchw), _ptr(out), B, H, W, _stream()), 'loans_prep_images_f32')
    return out


def nchw3_to_nhwc4(x):
    B, c, H, W = x.shape
    assert c == 3
    _chk(x, 'images')
    out = _empty((B, H, W, 4), device=x.device, dtype=torch.float32)
    check(_lib.load().loans_nchw3_to_nhwc4_f32(_ptr(x), _ptr(out), B, H, W, _stream()), 'loans_nchw3_to_nhwc4_f32')
    return out


# --------------------------------------------------------------------------- #
# batch normalisation
# --------------------------------------------------------------------------- #
STATS_REPLICAS = 32      # LOANS_STATS_REPLICAS


class _StepArena:
    """Every tensor an op allocates inside a training step (activations, gradients, staging copies: ~700 requests per step)
    is a slice of ONE device buffer, handed out in call order and never re-used inside the step -- 288 GB of HBM hold the SUM of
    a step's requests (15 .. 60 GB at the bench sizes), not only their peak -- and the next step starts at offset 0 again.
    Why not torch's caching allocator: its blocks are per stream and a block used on a side stream (the weight gradients) only
    returns when the host has seen that stream's event; the host runs ~10 ms ahead of the GPU, so requests keep missing the cache
    and 150 - 300 hipMalloc calls landed INSIDE the timed region of every leg, for as long as 40 steps.
    Safety is stream order alone: all streams of a step join the main stream before its optimiser update, the next step's first
    launch is behind that update.  Sized from the step before (the first step of a shape runs on torch's allocator and is
    measured); a request that does not fit, a request from another thread (the feed) and everything outside begin_step ..
    end_step falls back to ``torch.empty``.  Tensors of under 4 KiB (the reported losses, coefficient vectors) stay with torch:
    observers read them after the step.
    THE CONTRACT: a tensor of 4 KiB or more made by an op between begin_step and end_step -- a link's outputs, the
    gradient of a non-parameter input -- is overwritten by the NEXT step; whoever keeps one across ``update()`` calls must copy
    it.  What the package itself exposes across steps is allocated outside the arena: the reported losses (small), the parameters
    and their gradients (the links' own arenas), ``SheepLocalizer.last_transform_params`` (``mul(..., keep=True)``).
    ``tools/arena_poison.py`` fills the arena with NaN bit patterns at every begin_step to catch a violation in development."""
    SMALL = 4096
    ALIGN = 256
    MAX_FRACTION = 0.5

    def __init__(self):
        self.buf, self.views, self.off, self.need, self.want, self.live, self.pinned, self.thread = None, {}, 0, 0, 0, False, [], None
        self.misses, self.poison, self.picks, self.tried = 0, False, -1, -1

    def begin(self, device):
        capturing = torch.cuda.is_current_stream_capturing()
        # what the step that just ended asked for is a step's demand -- unless it timed tile candidates (every candidate of every
        # shape allocates its own outputs: 195 GB in a tuning step of configs[2]), which says nothing about the steps to come
        if self.need > 0 and self.picks == TIMED_PICKS:
            self.want = max(self.want, self.need)
        self.picks = TIMED_PICKS
        have = self.buf.numel() if self.buf is not None else 0
        if self.want > have and self.want != self.tried and not capturing:
            self.tried = self.want                  # one attempt per demand: a buffer that does not fit is not asked for every step
            size = (int(self.want * 1.06) + (64 << 20)) // 4096 * 4096
            total = torch.cuda.get_device_properties(device).total_memory
            if size <= int(total * self.MAX_FRACTION):  # (more than half the device for one step's tensors: leave it to torch's allocator)
                self.buf, self.views = None, {}
                torch.cuda.empty_cache()            # what the sizing step left in torch's cache is this buffer's memory now
                free, _ = torch.cuda.mem_get_info(device)
                if size <= int(free * 0.9):
                    self.buf = torch.empty(size, device=device, dtype=torch.uint8)
                    self.views = {dt: self.buf.view(dt) for dt in (torch.float32, BF16, torch.float64, torch.uint8, torch.int32, torch.int64)}
        if capturing and self.buf is not None and not any(b is self.buf for b in self.pinned):
            self.pinned.append(self.buf)         # a graph captured now has this buffer's addresses baked in: it outlives the graph
        self.off, self.need, self.live, self.thread = 0, 0, True, threading.get_ident()       # live: inside a step (sizing or serving)
        if self.poison and self.buf is not None and not capturing:
            self.buf.fill_(255)         # development (tools/arena_poison.py): every float of the workspace is a NaN until written

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= d
        item = _ITEMSIZE.get(dtype)
        if item is None:
            return None
        nbytes = n * item
        if nbytes < self.SMALL:
            return None
        nb = (nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.need += nb
        if self.buf is None or self.off + nb > self.buf.numel():
            self.misses += 1
            return None
        st, acc = [0] * len(shape), 1
        for i in range(len(shape) - 1, -1, -1):
            st[i] = acc
            acc *= shape[i]
        t = self.views[dtype].as_strided(shape, st, self.off // item)
        self.off += nb
        return t


_ITEMSIZE = {torch.float32: 4, torch.bfloat16: 2, torch.float64: 8, torch.uint8: 1, torch.int32: 4, torch.int64: 8}
_step_arenas = {}
STEP_ARENA = os.environ.get('LOANS_STEP_ARENA', '1') == '1'     # 0: every request goes to torch's caching allocator


def _empty(shape, device, dtype):
    """``torch.empty`` for a tensor that does not outlive the training step it is made in"""
    a = _step_arenas.get(device.index)
    if a is not None and a.live and a.thread == threading.get_ident():
        t = a.take((shape,) if isinstance(shape, int) else tuple(shape), dtype)
        if t is not None:
            return t
    return torch.empty(shape, device=device, dtype=dtype)


def step_arena_state(device):
    a = _step_arenas.get(torch.device(device).index if not isinstance(device, torch.device) else device.index)
    if a is None:
        return {'bytes': 0, 'used': 0, 'misses': 0}
    return {'bytes': a.buf.numel() if a.buf is not None else 0, 'used': a.off, 'misses': a.misses}


def _empty_like(x):
    if not x.is_contiguous():
        return torch.empty_like(x)
    return _empty(x.shape, x.device, x.dtype)


class _ZeroPool:
    """fp64 accumulators of one training step (BN statistics, BN-backward sums) carved from ONE buffer that a single
    memset clears at the start of the step, instead of one fill kernel per conv / BN (50 - 200 launches of ~5 us).
    Every consumer runs on the step's main stream, where the next step's memset is ordered behind it.  Outside a step
    (`begin_step` not called: tests, inference) and when the pool is exhausted, callers get a plain `torch.zeros`."""

    def __init__(self):
        self.buf, self.off, self.need, self.live, self.pinned = None, 0, 0, False, False

    def begin(self, device):
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # a hipGraph recorded from here on has this buffer's addresses baked into its memset, its conv-statistics atomics
            # and its BN kernels: the buffer must outlive the graph and never move.  A later eager step that needs more
            # (taller frames enabling res6 / res7 after a shape change) gets plain torch.zeros for what does not fit
            # (`take` returns None) instead of a reallocation that would hand the old memory back to the allocator.
            self.pinned = True
        elif not self.pinned and (self.buf is None or self.need > self.buf.numel()):
            self.buf = torch.empty(max(int(self.need * 1.25), 1 << 16), device=device, dtype=torch.float64)
        self.off, self.need, self.live = 0, 0, self.buf is not None
        if self.live:
            self.buf.zero_()

    def take(self, n):
        n8 = (n + 1) // 2 * 2               # 16-byte aligned slices
        self.need += n8
        if not self.live or self.off + n8 > self.buf.numel():
            return None
        v = self.buf[self.off:self.off + n]
        self.off += n8
        return v


_zero_pools = {}


# --------------------------------------------------------------------------- #
# per-step weight preparation: the operand forms of the weights that every step derives from the fp32 masters --
# bf16 copies for the bf16 arm's forward convolutions, per-stride-parity-class re-packs for the data gradients -- made ONCE at
# the start of a step (weights only change in the optimiser's update) by two launches instead of one cast per forward
# convolution and one re-pack per class and layer (47 + 94 launches of 4 - 20 us per step at configs[2], 72 at configs[1],
# all on the step's critical stream).  LOANS_WEIGHT_PREP=0 restores the per-call launches.
# --------------------------------------------------------------------------- #
WEIGHT_PREP = True
_MAX_PREP_JOBS = 2048


class _WeightPrep:
    """State of one device.  `arenas`: parameter arenas registered by ParamArena (their bf16 shadows are cast whole at
    begin_step).  `repacks`: (weight pointer, geometry key, bf16) -> persistent re-packed buffer; an entry is created (and
    filled by per-call launches) the first time a data gradient asks inside a step, every later step fills all entries with
    one loans_repack_dgrad_batch launch."""

    def __init__(self):
        self.arenas, self.repacks, self.order = [], {}, []
        self.table = None          # device copy of the job table (uint8 tensor), rebuilt when entries were added
        self.table_host = None
        self.total_tiles = 0
        self.dirty = False
        self.live = False          # inside a step whose begin_step prepared everything registered so far
        self.prepared = set()      # keys filled by this step's batch launch
        self.shadow_ok = {}        # id(arena) -> floats of its bf16 shadow cast by THIS step's begin_step
        self.step = 0
        # a captured hipGraph has the job table's address and every destination buffer of its entries baked into its
        # loans_repack_dgrad_batch launch; replays refresh neither `used` nor `step`.  Tables and keys that were current during a
        # capture are therefore kept for good: never freed by a rebuild (an eager step of another shape adds entries and makes
        # a NEW table), never evicted as stale.
        self.captured_tables, self.captured_keys = [], set()
        # LOANS_TILE_PW's fragment-order weights: (weight pointer, Cout, Cin) -> persistent packed buffer, all filled by one
        # loans_pw_pack_batch_f32 launch per step; the same life cycle as the re-packs
        self.pw, self.pw_order, self.pw_table, self.pw_units = {}, [], None, 0
        self.pw_dirty, self.pw_prepared, self.captured_pw_keys = False, set(), set()

    def _begin_pw(self, device, lib, st, capturing):
        self.pw_prepared = set()
        stale = [k for k in self.pw_order if self.pw[k]['used'] < self.step - 2 and k not in self.captured_pw_keys]
        if stale and not capturing:
            for k in stale:
                del self.pw[k]
            self.pw_order = [k for k in self.pw_order if k in self.pw]
            self.pw_dirty = True
        if not self.pw_order:
            self.pw_dirty = False
            return
        if self.pw_dirty:
            if capturing:           # a host -> device copy cannot be captured: this step packs per call (into the same buffers)
                return
            jobs = (_lib.PwPackJob * len(self.pw_order))()
            unit = 0
            for i, k in enumerate(self.pw_order):
                e = self.pw[k]
                jobs[i].src, jobs[i].dst, jobs[i].Cout, jobs[i].Cin, jobs[i].first_unit = k[0], e['buf'].data_ptr(), k[1], k[2], unit
                unit += k[1] * k[2] // 8
            raw = np.frombuffer(bytes(jobs), dtype=np.uint8).copy()
            self.pw_table = torch.from_numpy(raw).to(device)        # (a NEW tensor: a captured graph keeps the one it was captured with)
            self.pw_units, self.pw_njobs, self.pw_dirty = unit, len(self.pw_order), False
        if capturing and not any(t is self.pw_table for t in self.captured_tables):
            self.captured_tables.append(self.pw_table)
            self.captured_pw_keys.update(self.pw_order)
        check(lib.loans_pw_pack_batch_f32(_ptr(self.pw_table), self.pw_njobs, self.pw_units, st), 'loans_pw_pack_batch_f32')
        self.pw_prepared = set(self.pw_order)

    def begin(self, device):
        lib = _lib.load()
        st = _stream()
        self.prepared, self.shadow_ok = set(), {}
        self.live = True
        self.step += 1
        capturing = torch.cuda.is_current_stream_capturing()
        # entries no step has asked for lately belong to a model or a batch shape that is gone: dropped with their buffers
        # (never the entries a captured graph writes to)
        stale = [k for k in self.order if self.repacks[k]['used'] < self.step - 2 and k not in self.captured_keys]
        if stale and not capturing:
            for k in stale:
                del self.repacks[k]
            self.order = [k for k in self.order if k in self.repacks]
            self.dirty = True
        for ref in list(self.arenas):
            arena = ref()
            if arena is None:
                self.arenas.remove(ref)
                continue
            if (arena.precision or (COMPUTE, STORAGE))[1] == 'bf16':           # the model's own arm, else the process default
                # the WHOLE arena, not the active prefix: the prefix this step will use is only set later (the localizer's
                # __call__ looks at the frame height), so a step whose frames cross 224 / 300 px would read res6 / res7 -- or
                # Chainer's fc6 -- from a shadow that last step's shorter cast never wrote.  36 M floats:
                # 40 us.  _bf16_shadow checks every view against the length recorded here.
                if getattr(arena, 'data16', None) is None:
                    arena.data16 = torch.empty(arena.numel, device=arena.device, dtype=BF16)
                check(lib.loans_cast_bf16(_ptr(arena.data), _ptr(arena.data16), arena.numel, st), 'loans_cast_bf16')
                if CLASS_COUNT is not None: _acct('optimizer', 0, arena.numel * 4, arena.numel * 2)
                self.shadow_ok[id(arena)] = arena.numel
        self._begin_pw(device, lib, st, capturing)
        if not self.order:
            self.dirty = False
            return
        if self.dirty:
            jobs = (_lib.RepackJob * sum(len(self.repacks[k]['jobs']) for k in self.order))()
            i = tile = 0
            for k in self.order:
                for job in self.repacks[k]['jobs']:
                    C.memmove(C.byref(jobs[i]), C.byref(job), C.sizeof(_lib.RepackJob))
                    jobs[i].first_tile = tile
                    tile += job.tiles_co * job.tiles_ci * job.ntaps
                    i += 1
            raw = np.frombuffer(bytes(jobs), dtype=np.uint8).copy()
            if capturing:
                raise RuntimeError('the weight-preparation table changed inside a captured step')
            self.table_host = torch.from_numpy(raw)
            self.table = self.table_host.to(device)
            self.njobs, self.total_tiles, self.dirty = i, tile, False
        if capturing and not any(t is self.table for t in self.captured_tables):
            self.captured_tables.append(self.table)
            self.captured_keys.update(self.order)
        check(lib.loans_repack_dgrad_batch(_ptr(self.table), self.njobs, self.total_tiles, st), 'loans_repack_dgrad_batch')
        if CLASS_COUNT is not None:
            n = sum(self.repacks[k]['buf'].numel() for k in self.order)
            _acct('optimizer', 0, n * 4, sum(_nbytes(self.repacks[k]['buf']) for k in self.order))
        self.prepared = set(self.order)


_weight_preps = {}


def register_arena(arena):
    """ParamArena calls this: the arena's bf16 shadow is refreshed at every begin_step of the bf16 arm"""
    import weakref
    idx = arena.device.index if arena.device.type == 'cuda' else None
    if idx is not None:
        _weight_preps.setdefault(idx, _WeightPrep()).arenas.append(weakref.ref(arena))


def _bf16_shadow(w):
    """the bf16 copy of a parameter tensor inside its arena's shadow, cast by this step's begin_step; None outside a step or
    for a tensor that is no arena view"""
    wp = _weight_preps.get(w.device.index)
    if wp is None or not wp.live or not WEIGHT_PREP:
        return None
    ptr = w.data_ptr()
    for ref in wp.arenas:
        arena = ref()
        cast = wp.shadow_ok.get(id(arena)) if arena is not None else None
        if cast is None:
            continue
        base = arena.data.data_ptr()
        off = ptr - base
        if 0 <= off and off // 4 + w.numel() <= cast and off % 4 == 0 and w.is_contiguous():
            return arena.data16[off // 4:off // 4 + w.numel()].view(w.shape)
    return None


def _prepacked_dgrad_weights(w, geo, bf16):
    """the re-packed data-gradient weights of (w, geo) -- all stride-parity classes in one buffer, class c at geo.dgrad[c]'s
    offset -- prepared by this step's begin_step, or prepared now by per-class launches and registered for the next steps;
    None outside a step (tests, inference: the caller re-packs per call)"""
    wp = _weight_preps.setdefault(w.device.index, _WeightPrep())
    if not wp.live or not WEIGHT_PREP or not geo.dgrad or max(d.ntaps for d, _, _ in geo.dgrad) > _lib.REPACK_JOB_TAPS:
        return None
    key = (w.data_ptr(), geo.key, bool(bf16))
    entry = wp.repacks.get(key)
    if entry is not None:
        entry['used'] = wp.step
        if key in wp.prepared:
            return entry['buf']
    lib = _lib.load()
    if entry is None:
        if len(wp.order) >= _MAX_PREP_JOBS or torch.cuda.is_current_stream_capturing():
            return None
        buf = torch.empty(geo.dgrad_weight_floats, device=w.device, dtype=BF16 if bf16 else torch.float32)
        jobs = []
        for d, tapsel, off in geo.dgrad:
            j = _lib.RepackJob()
            j.src, j.dst = w.data_ptr(), buf.data_ptr() + off * buf.element_size()
            j.Cout, j.Cin, j.src_taps, j.ntaps = geo.Cout, geo.Cin, geo.k * geo.k, d.ntaps
            for t in range(d.ntaps):
                j.tapsel[t] = tapsel[t]
            j.tiles_co, j.tiles_ci, j.dst_bf16 = (geo.Cout + 63) // 64, (geo.Cin + 63) // 64, int(bool(bf16))
            jobs.append(j)
        entry = wp.repacks[key] = {'buf': buf, 'jobs': jobs, 'w': w, 'used': wp.step}      # `w` keeps the source (its arena) alive
        wp.order.append(key)
        wp.dirty = True
    fn = lib.loans_repack_dgrad_bf16 if bf16 else lib.loans_repack_dgrad_f32
    for d, tapsel, off in geo.dgrad:
        check(fn(_ptr(w), _ptr(entry['buf'][off:]), geo.Cout, geo.Cin, geo.k * geo.k, tapsel, d.ntaps, _stream()),
              'loans_repack_dgrad')
    return entry['buf']


def begin_step(device):
    """Called by the updater at the start of a training step: one memset for the step's accumulators (_ZeroPool), the step's
    weight preparation (_WeightPrep)."""
    device = torch.device(device) if not isinstance(device, torch.device) else device
    idx = device.index if device.index is not None else torch.cuda.current_device()
    _zero_pools.setdefault(idx, _ZeroPool()).begin(torch.device('cuda', idx))
    if STEP_ARENA:
        _step_arenas.setdefault(idx, _StepArena()).begin(torch.device('cuda', idx))
    if WEIGHT_PREP:
        _weight_preps.setdefault(idx, _WeightPrep()).begin(torch.device('cuda', idx))


def end_step(device=None):
    """Accumulators requested after this come from `torch.zeros` again (the pool is only valid inside a step), weights are
    cast / re-packed per call again (the optimisers have moved them)."""
    for p in _zero_pools.values():
        p.live = False
    for p in _step_arenas.values():
        p.live = False
    for wp in _weight_preps.values():
        wp.live = False


def _zeros_f64(shape, device):
    n = 1
    for d in shape:
        n *= d
    pool = _zero_pools.get(device.index if isinstance(device, torch.device) else torch.device(device).index)
    v = pool.take(n) if pool is not None else None
    if v is None:
        return torch.zeros(shape, device=device, dtype=torch.float64)
    return v.view(shape)


def stats_buffer(C_, device):
    """Zeroed fp64 accumulators for the conv epilogue's BN statistics: [replica][sum | sumsq][C]."""
    return _zeros_f64((STATS_REPLICAS, 2, C_), device)


class BNState:
    """Per-call batch statistics + affine coefficients (device vectors of C floats)."""
    __slots__ = ('mean', 'rstd', 'scale', 'shift', 'count')

    def __init__(self, C_, device):
        buf = _empty((4, C_), device=device, dtype=torch.float32)
        self.mean, self.rstd, self.scale, self.shift = buf[0], buf[1], buf[2], buf[3]
        self.count = 0

    def affine(self):
        """float[2][C] = scale, shift as one table (LOANS_F_AFFINE_IN): the two rows lie behind each other in this state's buffer"""
        assert self.shift.data_ptr() == self.scale.data_ptr() + 4 * self.scale.numel()
        return self.scale


def bn_finalize(stats, count, gamma, beta, running_mean, running_var):
    C_ = gamma.numel()
    st = BNState(C_, gamma.device)
    st.count = count
    check(_lib.load().loans_bn_finalize_f32(_ptr(stats), C_, count, BN_EPS, BN_DECAY, _ptr(gamma), _ptr(beta),
                                            _ptr(running_mean), _ptr(running_var), RUNNING_VAR_INCLUDES_EPS,
                                            _ptr(st.mean), _ptr(st.rstd), _ptr(st.scale), _ptr(st.shift), _stream()),
          'loans_bn_finalize_f32')
    return st


def bn_eval_coeffs(gamma, beta, running_mean, running_var):
    C_ = gamma.numel()
    st = BNState(C_, gamma.device)
    check(_lib.load().loans_bn_eval_coeffs_f32(C_, BN_EPS, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                               _ptr(running_var), _ptr(st.mean), _ptr(st.rstd), _ptr(st.scale),
                                               _ptr(st.shift), _stream()), 'loans_bn_eval_coeffs_f32')
    return st




def bn_apply(x, st, relu=True, residual=None, x2=None, st2=None, want_bits=False):
    """y = act(bn(x) [+ residual | + bn2(x2)]).  want_bits: also write the sign bits of y (one byte per four channels) and
    hang them on the result as ``y.relu_bits`` -- bn_backward then takes the ReLU mask from them instead of from y."""
    C_ = x.shape[-1]
    rows = x.numel() // C_
    y = _empty_like(x)
    mode = 0
    second = None
    if residual is not None:
        mode, second = 1, residual
    elif x2 is not None:
        mode, second = 2, x2
    lib = _lib.load()
    assert second is None or second.dtype == x.dtype
    if CLASS_COUNT is not None: _acct('bn_fwd', 0, _nbytes(x, second), _nbytes(y) + (rows * (C_ // 4) if (want_bits and relu) else 0))
    if want_bits and relu:
        bits = _empty(rows * (C_ // 4), device=x.device, dtype=torch.uint8)
        fn = lib.loans_bn_apply_bits_bf16 if _is16(x) else lib.loans_bn_apply_bits_f32
        check(fn(_ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(second),
                 _ptr(st2.scale if st2 else None), _ptr(st2.shift if st2 else None),
                 _ptr(y), _ptr(bits), rows, C_, mode, 1, _stream()), 'loans_bn_apply_bits')
        y.relu_bits = bits
        return y
    fn = lib.loans_bn_apply_bf16 if _is16(x) else lib.loans_bn_apply_f32
    check(fn(_ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(second),
             _ptr(st2.scale if st2 else None), _ptr(st2.shift if st2 else None),
             _ptr(y), rows, C_, mode, 1 if relu else 0, _stream()), 'loans_bn_apply')
    return y


# The stem's pool also writes the RAW conv output at every pooled element's argmax (a pooled-size tensor): the sums of the
# stem's BN backward then read two contiguous pooled-size tensors instead of gathering out of the four times larger conv
# output (configs[2]: 1.34 GB -> 0.54 GB for that pass, +0.27 GB written here).  False: the gathering pass (tests of it).
POOL_ARGMAX_VALUES = True


def bn_relu_maxpool(x, st, want_sel=False):
    """(y, idx), or (y, idx, xsel) with want_sel (xsel None where the 16-byte-unit kernel does not tile the channel count)"""
    B, H, W, C_ = x.shape
    OH, OW = conv_outsize(H, 3, 2, 0, True), conv_outsize(W, 3, 2, 0, True)
    s16 = _is16(x)
    y = _empty((B, OH, OW, C_), device=x.device, dtype=x.dtype)
    idx = _empty((B, OH, OW, C_), device=x.device, dtype=torch.uint8)
    lib = _lib.load()
    if want_sel and POOL_ARGMAX_VALUES and bn_units_ok(C_, s16) and reduce_channels_ok(C_) and C_ <= 1024:
        xsel = _empty((B, OH, OW, C_), device=x.device, dtype=x.dtype)
        if CLASS_COUNT is not None: _acct('stem', 0, _nbytes(x), _nbytes(y, idx, xsel))
        fn = lib.loans_bn_relu_maxpool_sel_bf16 if s16 else lib.loans_bn_relu_maxpool_sel_f32
        check(fn(_ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(y), _ptr(idx), _ptr(xsel), B, H, W, C_, OH, OW, _stream()),
              'loans_bn_relu_maxpool_sel')
        return y, idx, xsel
    if CLASS_COUNT is not None: _acct('stem', 0, _nbytes(x), _nbytes(y, idx))
    fn = lib.loans_bn_relu_maxpool_bf16 if s16 else lib.loans_bn_relu_maxpool_f32
    check(fn(_ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(y), _ptr(idx), B, H, W, C_, OH, OW, _stream()),
          'loans_bn_relu_maxpool')
    return (y, idx, None) if want_sel else (y, idx)


def maxpool_relu_bwd(gy, idx, x, st):
    B, H, W, C_ = x.shape
    OH, OW = gy.shape[1], gy.shape[2]
    gx = _empty_like(x)
    lib = _lib.load()
    assert gy.dtype == x.dtype
    fn = lib.loans_maxpool_relu_bwd_bf16 if _is16(gy) else lib.loans_maxpool_relu_bwd_f32
    check(fn(_ptr(gy), _ptr(idx), _ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(gx), B, H, W, C_, OH, OW, _stream()),
          'loans_maxpool_relu_bwd')
    return gx


# (Settled in rounds 2-3 and no longer switchable: a BN's own ReLU mask is recomputed from its input instead of read from the
# activation; a unit's final BN reads its mask as sign bits; the reductions add into replicated accumulators.  The older kernels
# remain as the fall-back for channel counts the 16-byte-unit kernels do not tile.)


def bn_units_ok(C_, s16):
    """the 16-byte-unit BN kernels tile this channel count: C / V a divisor of 256 (V = 8 bf16 / 4 fp32 channels)"""
    v = 8 if s16 else 4
    return C_ % v == 0 and C_ // v <= 256 and 256 % (C_ // v) == 0


def _bn_backward_rep(lib, gy, mask, x, st, gamma, ggamma, gbeta, x2, st2, gamma2, ggamma2, gbeta2, mask_is_own_relu):
    """bn_backward on the 16-byte-unit kernels with replicated accumulators: reduce (every mask form through one entry) ->
    coefficients from the replicas -> apply"""
    C_ = x.shape[-1]
    rows = x.numel() // C_
    dual = x2 is not None
    s16 = _is16(x)
    s = _stream()
    assert _is16(gy) == s16 and (x2 is None or _is16(x2) == s16)
    own = mask_is_own_relu and not dual and mask is not None
    bits = None if own else (getattr(mask, 'relu_bits', None) if mask is not None else None)
    if bits is not None:
        assert bits.numel() == rows * (C_ // 4)
    kind = 2 if own else (3 if bits is not None else (1 if mask is not None else 0))
    mten = None if kind in (0, 2) else (bits if kind == 3 else mask)
    assert kind != 1 or _is16(mask) == s16
    ns = 4 if dual else 2
    sums = _zeros_f64((STATS_REPLICAS, ns, C_), x.device)
    red = lib.loans_bn_bwd_reduce_rep_bf16 if s16 else lib.loans_bn_bwd_reduce_rep_f32
    check(red(_ptr(gy), _ptr(mten), kind, _ptr(x), _ptr(st.mean), _ptr(st.rstd), _ptr(x2), _ptr(st2.mean if dual else None),
              _ptr(st2.rstd if dual else None), _ptr(st.scale if own else None), _ptr(st.shift if own else None), _ptr(sums),
              STATS_REPLICAS, rows, C_, s), 'loans_bn_bwd_reduce_rep')
    k = _empty((6 if dual else 3, C_), device=x.device, dtype=torch.float32)
    check(lib.loans_bn_bwd_coeffs_rep_f32(_ptr(sums), STATS_REPLICAS, ns * C_, 0, C_, rows, _ptr(gamma), _ptr(st.mean),
                                          _ptr(st.rstd), _ptr(ggamma), _ptr(gbeta), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), s),
          'loans_bn_bwd_coeffs_rep_f32')
    gx = _empty_like(x)
    if own:
        app = lib.loans_bn_bwd_apply_xmask_bf16 if s16 else lib.loans_bn_bwd_apply_xmask_f32
        check(app(_ptr(gy), _ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(gx), rows, C_, s),
              'loans_bn_bwd_apply_xmask')
        return gx
    gx2 = None
    if dual:
        check(lib.loans_bn_bwd_coeffs_rep_f32(_ptr(sums[0, 2]), STATS_REPLICAS, ns * C_, 0, C_, rows, _ptr(gamma2), _ptr(st2.mean),
                                              _ptr(st2.rstd), _ptr(ggamma2), _ptr(gbeta2), _ptr(k[3]), _ptr(k[4]), _ptr(k[5]), s),
              'loans_bn_bwd_coeffs_rep_f32')
        gx2 = _empty_like(x2)
    if kind == 3:
        app = lib.loans_bn_bwd_apply_bits_bf16 if s16 else lib.loans_bn_bwd_apply_bits_f32
    else:
        app = lib.loans_bn_bwd_apply_bf16 if s16 else lib.loans_bn_bwd_apply_f32
    check(app(_ptr(gy), _ptr(mten), _ptr(x), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(gx), _ptr(x2), _ptr(k[3]) if dual else 0,
              _ptr(k[4]) if dual else 0, _ptr(k[5]) if dual else 0, _ptr(gx2), rows, C_, s), 'loans_bn_bwd_apply')
    return (gx, gx2) if dual else gx


def bn_backward(gy, mask, x, st, gamma, ggamma, gbeta, x2=None, st2=None, gamma2=None, ggamma2=None, gbeta2=None,
                mask_is_own_relu=False):
    """Training-mode BN backward for one or two BNs fed by the same upstream gradient
    g = gy * (mask > 0).  Accumulates ggamma/gbeta in place, returns gx (and gx2).
    mask_is_own_relu: `mask` is relu(x*scale+shift), this BN's own activation -- its sign is recomputed from x and the
    mask tensor is not read (one tensor less in both passes)."""
    lib = _lib.load()
    C_ = x.shape[-1]
    rows = x.numel() // C_
    dual = x2 is not None
    if CLASS_COUNT is not None:
        own = mask_is_own_relu and not dual and mask is not None
        bits = None if own else (getattr(mask, 'relu_bits', None) if mask is not None else None)
        m = None if own else (bits if bits is not None else mask)
        _acct('bn_bwd', 0, 2 * _nbytes(gy, x, x2, m), _nbytes(x) * (2 if dual else 1))       # sums, then apply: two passes
    if bn_units_ok(C_, _is16(x)):
        return _bn_backward_rep(lib, gy, mask, x, st, gamma, ggamma, gbeta, x2, st2, gamma2, ggamma2, gbeta2, mask_is_own_relu)
    if mask_is_own_relu and not dual and mask is not None:
        s = _stream()
        s16 = _is16(x)
        assert _is16(gy) == s16
        sums = _zeros_f64((2, C_), x.device)
        red_fn = lib.loans_bn_bwd_reduce_xmask_bf16 if s16 else lib.loans_bn_bwd_reduce_xmask_f32
        app_fn = lib.loans_bn_bwd_apply_xmask_bf16 if s16 else lib.loans_bn_bwd_apply_xmask_f32
        check(red_fn(_ptr(gy), _ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(st.mean), _ptr(st.rstd), _ptr(sums), rows, C_, s),
              'loans_bn_bwd_reduce_xmask')
        k = _empty((3, C_), device=x.device, dtype=torch.float32)
        check(lib.loans_bn_bwd_coeffs_f32(_ptr(sums), C_, rows, _ptr(gamma), _ptr(st.mean), _ptr(st.rstd), _ptr(ggamma),
                                          _ptr(gbeta), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), s), 'loans_bn_bwd_coeffs_f32')
        gx = _empty_like(x)
        check(app_fn(_ptr(gy), _ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(gx), rows, C_,
                     s), 'loans_bn_bwd_apply_xmask')
        return gx
    sums = _zeros_f64((4 if dual else 2, C_), x.device)
    s = _stream()
    s16 = _is16(x)
    assert _is16(gy) == s16 and (mask is None or _is16(mask) == s16) and (x2 is None or _is16(x2) == s16)
    red_fn = lib.loans_bn_bwd_reduce_bf16 if s16 else lib.loans_bn_bwd_reduce_f32
    app_fn = lib.loans_bn_bwd_apply_bf16 if s16 else lib.loans_bn_bwd_apply_f32
    bits = getattr(mask, 'relu_bits', None) if mask is not None else None
    if bits is not None:            # the mask as sign bits (ops.bn_apply(..., want_bits=True)) instead of the tensor
        assert bits.numel() == rows * (C_ // 4)
        mask = bits
        red_fn = lib.loans_bn_bwd_reduce_bits_bf16 if s16 else lib.loans_bn_bwd_reduce_bits_f32
        app_fn = lib.loans_bn_bwd_apply_bits_bf16 if s16 else lib.loans_bn_bwd_apply_bits_f32
    check(red_fn(_ptr(gy), _ptr(mask), _ptr(x), _ptr(st.mean), _ptr(st.rstd), _ptr(x2),
                                      _ptr(st2.mean if dual else None), _ptr(st2.rstd if dual else None),
                                      _ptr(sums), rows, C_, s), 'loans_bn_bwd_reduce_f32')
    k = _empty((6 if dual else 3, C_), device=x.device, dtype=torch.float32)
    check(lib.loans_bn_bwd_coeffs_f32(_ptr(sums), C_, rows, _ptr(gamma), _ptr(st.mean), _ptr(st.rstd), _ptr(ggamma),
                                      _ptr(gbeta), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), s), 'loans_bn_bwd_coeffs_f32')
    gx = _empty_like(x)
    gx2 = None
    if dual:
        check(lib.loans_bn_bwd_coeffs_f32(_ptr(sums[2]), C_, rows, _ptr(gamma2), _ptr(st2.mean), _ptr(st2.rstd),
                                          _ptr(ggamma2), _ptr(gbeta2), _ptr(k[3]), _ptr(k[4]), _ptr(k[5]), s),
              'loans_bn_bwd_coeffs_f32')
        gx2 = _empty_like(x2)
    check(app_fn(_ptr(gy), _ptr(mask), _ptr(x), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(gx),
                                     _ptr(x2), _ptr(k[3]) if dual else 0, _ptr(k[4]) if dual else 0,
                                     _ptr(k[5]) if dual else 0, _ptr(gx2), rows, C_, s), 'loans_bn_bwd_apply_f32')
    return (gx, gx2) if dual else gx


def bn_backward_from_sums(gy, x, st, sums, gamma, ggamma, gbeta):
    """bn_backward(gy, relu(bn(x)), x, st, ..., mask_is_own_relu=True) with the reduction already done: `sums` are the
    [replicas][2][C] accumulators conv_dgrad(..., bn_sums=(x, st)) filled while gy was in registers.  One pass over the
    tensors (gx = k1 g m + k2 x + k3) instead of two."""
    lib = _lib.load()
    C_ = x.shape[-1]
    rows = x.numel() // C_
    s = _stream()
    s16 = _is16(x)
    if CLASS_COUNT is not None: _acct('bn_bwd', 0, _nbytes(gy, x), _nbytes(x))              # the sums rode in the data gradient's epilogue: one pass
    assert _is16(gy) == s16 and sums.numel() == STATS_REPLICAS * 2 * C_
    k = _empty((3, C_), device=x.device, dtype=torch.float32)
    check(lib.loans_bn_bwd_coeffs_rep_f32(_ptr(sums), STATS_REPLICAS, 2 * C_, 1, C_, rows, _ptr(gamma), _ptr(st.mean), _ptr(st.rstd),
                                          _ptr(ggamma), _ptr(gbeta), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), s),
          'loans_bn_bwd_coeffs_rep_f32')
    gx = _empty_like(x)
    app_fn = lib.loans_bn_bwd_apply_xmask_bf16 if s16 else lib.loans_bn_bwd_apply_xmask_f32
    check(app_fn(_ptr(gy), _ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(gx), rows, C_, s),
          'loans_bn_bwd_apply_xmask')
    return gx




def pool_bn_backward(gy, idx, x, st, gamma, ggamma, gbeta, gbias=None, xsel=None):
    """The stem's tail backwards (max_pooling_2d -> relu -> bn1, sheep/resnet.py:72-73) without the dense gradient
    between pool and BN: gx w.r.t. the conv output x from the pooled gradient gy and the argmax positions idx;
    accumulates ggamma / gbeta (and gbias += per-channel sum of gx, conv1's bias gradient) in place.  Same result as bn_backward(maxpool_relu_bwd(gy, idx, x, st), None, x, st, ...)
    up to summation order (and, on bf16 tensors, without the rounding of the intermediate)."""
    B, H, W, C_ = x.shape
    OH, OW = gy.shape[1], gy.shape[2]
    if xsel is not None:
        if CLASS_COUNT is not None: _acct('stem', 0, _nbytes(gy, xsel) + _nbytes(gy, idx) + _nbytes(x), _nbytes(x))  # sums over (gy, xsel), then gx
    else:
        if CLASS_COUNT is not None: _acct('stem', 0, 2 * _nbytes(gy, idx) + 2 * _nbytes(x), _nbytes(x))    # sums over the pooled gradient (gathers x), then gx
    if not (reduce_channels_ok(C_) and C_ <= 1024):        # (the three-pass form: channel counts the fused pair does not tile)
        gx = bn_backward(maxpool_relu_bwd(gy, idx, x, st), None, x, st, gamma, ggamma, gbeta)
        if gbias is not None:
            colsum_acc(gx, gbias)
        return gx
    lib = _lib.load()
    s16 = _is16(x)
    assert _is16(gy) == s16
    s = _stream()
    app_fn = lib.loans_pool_bn_bwd_apply_bf16 if s16 else lib.loans_pool_bn_bwd_apply_f32
    k = _empty((3, C_), device=x.device, dtype=torch.float32)
    sums = _zeros_f64((STATS_REPLICAS, 2, C_), x.device)
    if xsel is not None:
        # sum g' and sum g' xhat over the POOLED elements, g' = gy * (bn(xsel) > 0): every window's gradient lands on its argmax,
        # whose x is xsel -- the same two sums as over the conv output's pixels, term by term
        assert xsel.shape == gy.shape and _is16(xsel) == s16
        red = lib.loans_bn_bwd_reduce_rep_bf16 if s16 else lib.loans_bn_bwd_reduce_rep_f32
        check(red(_ptr(gy), None, 2, _ptr(xsel), _ptr(st.mean), _ptr(st.rstd), None, None, None, _ptr(st.scale), _ptr(st.shift),
                  _ptr(sums), STATS_REPLICAS, gy.numel() // C_, C_, s), 'loans_bn_bwd_reduce_rep')
    else:
        red_fn = lib.loans_pool_bn_bwd_reduce_rep_bf16 if s16 else lib.loans_pool_bn_bwd_reduce_rep_f32
        check(red_fn(_ptr(gy), _ptr(idx), _ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(st.mean), _ptr(st.rstd), _ptr(sums),
                     STATS_REPLICAS, B, H, W, C_, OH, OW, s), 'loans_pool_bn_bwd_reduce_rep')
    check(lib.loans_bn_bwd_coeffs_rep_f32(_ptr(sums), STATS_REPLICAS, 2 * C_, 0, C_, B * H * W, _ptr(gamma), _ptr(st.mean),
                                          _ptr(st.rstd), _ptr(ggamma), _ptr(gbeta), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), s),
          'loans_bn_bwd_coeffs_rep_f32')
    gx = _empty_like(x)
    if gbias is not None and C_ // 4 <= 256 and 256 % (C_ // 4) == 0:
        # the bias-gradient sums go through 32 replicas (fp32 views of the step's zeroed accumulator pool), then one fold
        reps = _zeros_f64((STATS_REPLICAS // 2, C_), x.device).view(torch.float32).view(STATS_REPLICAS, C_)
        rep_fn = lib.loans_pool_bn_bwd_apply_rep_bf16 if s16 else lib.loans_pool_bn_bwd_apply_rep_f32
        check(rep_fn(_ptr(gy), _ptr(idx), _ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(gx),
                     _ptr(reps), STATS_REPLICAS, B, H, W, C_, OH, OW, s), 'loans_pool_bn_bwd_apply_rep')
        check(lib.loans_fold_replicas_f32(_ptr(reps), _ptr(gbias), STATS_REPLICAS, C_, s), 'loans_fold_replicas_f32')
        return gx
    check(app_fn(_ptr(gy), _ptr(idx), _ptr(x), _ptr(st.scale), _ptr(st.shift), _ptr(k[0]), _ptr(k[1]), _ptr(k[2]), _ptr(gx),
                 _ptr(gbias), B, H, W, C_, OH, OW, s), 'loans_pool_bn_bwd_apply')
    return gx


def colsum_acc(x, out):
    C_ = x.shape[-1]
    lib = _lib.load()
    fn = lib.loans_colsum_bf16 if _is16(x) else lib.loans_colsum_f32
    check(fn(_ptr(x), _ptr(out), x.numel() // C_, C_, _stream()), 'loans_colsum')


# --------------------------------------------------------------------------- #
# small dense ops
# --------------------------------------------------------------------------- #
def gap_fwd(x):
    B, H, W, C_ = x.shape
    y = _empty((B, C_), device=x.device, dtype=torch.float32)
    if CLASS_COUNT is not None: _acct('heads', 0, _nbytes(x), _nbytes(y))
    lib = _lib.load()
    fn = lib.loans_gap_fwd_bf16_f32 if _is16(x) else lib.loans_gap_fwd_f32
    check(fn(_ptr(x), _ptr(y), B, H * W, C_, _stream()), 'loans_gap_fwd')
    return y


def gap_bwd(gy, shape, dtype=torch.float32):
    B, H, W, C_ = shape
    gx = _empty(shape, device=gy.device, dtype=dtype)
    if CLASS_COUNT is not None: _acct('heads', 0, _nbytes(gy), _nbytes(gx))
    lib = _lib.load()
    fn = lib.loans_gap_bwd_f32_bf16 if dtype == BF16 else lib.loans_gap_bwd_f32
    check(fn(_ptr(gy), _ptr(gx), B, H * W, C_, _stream()), 'loans_gap_bwd')
    return gx


def linear_fwd(x, W, b, act_in=False, act_out=False):
    B = x.shape[0]
    K = x.numel() // B
    N = W.numel() // K
    y = _empty((B, N), device=x.device, dtype=torch.float32)
    if CLASS_COUNT is not None: _acct('heads', 2 * B * K * N, _nbytes(x, W), _nbytes(y))
    lib = _lib.load()
    fn = lib.loans_linear_fwd_bf16 if _is16(x) else lib.loans_linear_fwd_f32
    check(fn(_ptr(x), _ptr(W), _ptr(b), _ptr(y), B, K, N, int(act_in), int(act_out), _stream()), 'loans_linear_fwd')
    return y


def linear_bwd(x, W, y, gy, gW=None, gb=None, need_gx=True, act_in=False, act_out=False):
    B = x.shape[0]
    K = x.numel() // B
    N = W.numel() // K
    gx = _empty_like(x) if need_gx else None
    if CLASS_COUNT is not None: _acct('heads', (4 if need_gx else 2) * B * K * N, _nbytes(x, W, gy), _nbytes(gx, gW))
    lib = _lib.load()
    fn = lib.loans_linear_bwd_bf16 if _is16(x) else lib.loans_linear_bwd_f32
    check(fn(_ptr(x), _ptr(W), _ptr(y), _ptr(gy), _ptr(gx), _ptr(gW), _ptr(gb),
             B, K, N, int(act_in), int(act_out), _stream()), 'loans_linear_bwd')
    return gx


def mul(x, m, keep=False):
    """keep: the result outlives the step (a tensor callers hold across update() calls): never a slice of the step arena"""
    y = torch.empty_like(x) if keep else _empty_like(x)
    check(_lib.load().loans_mul_f32(_ptr(x), _ptr(m), _ptr(y), x.numel(), _stream()), 'loans_mul_f32')
    return y


def axpby(a, x, b, y):
    """y = a*x + b*y in place (fp32, contiguous)."""
    assert x.numel() == y.numel()
    _chk(x, 'x')
    _chk(y, 'y')
    check(_lib.load().loans_axpby_f32(a, _ptr(x), b, _ptr(y), x.numel(), _stream()), 'loans_axpby_f32')
    return y


# --------------------------------------------------------------------------- #
# VisualBackprop / grayscale rois (insights/visual_backprop.py; sheep_localizer.py:65-68)
# --------------------------------------------------------------------------- #
# When a list, the fused block functions append one record per Convolution2D / max-pooling node of the MAIN branch they run,
# in forward order: {'avg': channel mean of the node's input [B][H][W] fp32, 'k', 's', 'p'} -- what VisualBackprop's walk
# over Chainer's graph reads off each such node (visual_backprop.py:18-20,29-40)
VBP_TAPS = None


def channel_mean(x, cdiv=None, st=None):
    """mean over the channels of an NHWC tensor -> [B][H][W] fp32; st: apply relu(x * scale + shift) first"""
    C_ = x.shape[-1]
    rows = x.numel() // C_
    out = _empty(x.shape[:-1], device=x.device, dtype=torch.float32)
    fn = _lib.load().loans_channel_mean_bf16 if _is16(x) else _lib.load().loans_channel_mean_f32
    check(fn(_ptr(x), _ptr(st.scale if st is not None else None), _ptr(st.shift if st is not None else None), _ptr(out), rows, C_,
             cdiv or C_, _stream()), 'loans_channel_mean')
    return out


def vbp_tap(x, k, s, p, cdiv=None, st=None):
    if VBP_TAPS is not None:
        VBP_TAPS.append({'avg': channel_mean(x, cdiv, st), 'k': k, 's': s, 'p': p})


def vbp_scale(feat, avg, k, s, p):
    """deconvolution_2d(feat, ones, stride s, pad p, outsize = avg's size) * avg; the all-ones kernel's size follows from the
    sizes (visual_backprop.py:29-30)"""
    B, fh, fw = feat.shape
    _, H, W = avg.shape
    kh, kw = H + 2 * p - s * (fh - 1), W + 2 * p - s * (fw - 1)
    out = _empty_like(avg)
    check(_lib.load().loans_vbp_scale_f32(_ptr(feat), _ptr(avg), _ptr(out), B, fh, fw, H, W, kh, kw, s, s, p, p, _stream()),
          'loans_vbp_scale_f32')
    return out


def minmax_normalize_(x):
    B = x.shape[0]
    check(_lib.load().loans_minmax_normalize_f32(_ptr(x), B, x.numel() // B, _stream()), 'loans_minmax_normalize_f32')
    return x


def gray_fwd(rois_nhwc4):
    B, h, w, c = rois_nhwc4.shape
    assert c == 4
    _chk(rois_nhwc4, 'rois')
    out = _empty((B, h, w), device=rois_nhwc4.device, dtype=torch.float32)
    check(_lib.load().loans_gray_fwd_f32(_ptr(rois_nhwc4), _ptr(out), B * h * w, _stream()), 'loans_gray_fwd_f32')
    return out


def gray_bwd(g):
    _chk(g, 'gradient')
    out = _empty(tuple(g.shape) + (4,), device=g.device, dtype=torch.float32)
    check(_lib.load().loans_gray_bwd_f32(_ptr(g), _ptr(out), g.numel(), _stream()), 'loans_gray_bwd_f32')
    return out


# --------------------------------------------------------------------------- #
# spatial transformer
# --------------------------------------------------------------------------- #
def st_grid_fwd(theta, out_size):
    B = theta.shape[0]
    th, tw = out_size
    grid = _empty((B, 2, th, tw), device=theta.device, dtype=torch.float32)
    check(_lib.load().loans_st_grid_fwd_f32(_ptr(theta), _ptr(grid), B, th, tw, _stream()), 'loans_st_grid_fwd_f32')
    return grid


def st_grid_bwd(ggrid):
    B, _, th, tw = ggrid.shape
    gtheta = _empty((B, 2, 3), device=ggrid.device, dtype=torch.float32)
    check(_lib.load().loans_st_grid_bwd_f32(_ptr(ggrid), _ptr(gtheta), B, th, tw, _stream()), 'loans_st_grid_bwd_f32')
    return gtheta


def st_sampler_fwd(images_nchw, grid):
    B, _, H, W = images_nchw.shape
    th, tw = grid.shape[2:]
    rois = _empty((B, th, tw, 4), device=grid.device, dtype=torch.float32)
    if CLASS_COUNT is not None: _acct('crop', 0, _nbytes(grid) + 4 * 3 * 4 * B * th * tw, _nbytes(rois))        # four taps of three channels per crop pixel
    check(_lib.load().loans_st_sampler_fwd_f32(_ptr(images_nchw), _ptr(grid), _ptr(rois), B, H, W, th, tw, _stream()),
          'loans_st_sampler_fwd_f32')
    return rois


def st_sampler_bwd_grid(images_nchw, grid, grois):
    B, _, H, W = images_nchw.shape
    th, tw = grid.shape[2:]
    ggrid = _empty_like(grid)
    if CLASS_COUNT is not None: _acct('crop', 0, _nbytes(grid, grois) + 4 * 3 * 4 * B * th * tw, _nbytes(ggrid))
    check(_lib.load().loans_st_sampler_bwd_grid_f32(_ptr(images_nchw), _ptr(grid), _ptr(grois), _ptr(ggrid), 0,
                                                    B, H, W, th, tw, _stream()), 'loans_st_sampler_bwd_grid_f32')
    return ggrid


# --------------------------------------------------------------------------- #
# losses / optimiser
# --------------------------------------------------------------------------- #
def mse_fwd(y, target=None, tconst=0.0):
    loss = _empty((), device=y.device, dtype=torch.float32)
    check(_lib.load().loans_mse_fwd_f32(_ptr(y), _ptr(target), tconst, _ptr(loss), y.numel(), _stream()), 'loans_mse_fwd_f32')
    return loss


def mse_bwd(y, gloss, target=None, tconst=0.0):
    gy = _empty_like(y)
    check(_lib.load().loans_mse_bwd_f32(_ptr(y), _ptr(target), tconst, _ptr(gloss), _ptr(gy), y.numel(), _stream()),
          'loans_mse_bwd_f32')
    return gy


def grid_loss_fwd(grid, kind, img_h=0.0, img_w=0.0, oob_scale=1.0):
    B, _, th, tw = grid.shape
    loss = _empty((), device=grid.device, dtype=torch.float32)
    check(_lib.load().loans_grid_loss_fwd_f32(_ptr(grid), _ptr(loss), kind, img_h, img_w, oob_scale, B, th, tw,
                                              _stream()), 'loans_grid_loss_fwd_f32')
    return loss


def grid_loss_bwd(grid, gloss, kind, img_h=0.0, img_w=0.0, oob_scale=1.0):
    B, _, th, tw = grid.shape
    ggrid = _empty_like(grid).zero_()
    check(_lib.load().loans_grid_loss_bwd_f32(_ptr(grid), _ptr(gloss), _ptr(ggrid), kind, img_h, img_w, oob_scale,
                                              B, th, tw, _stream()), 'loans_grid_loss_bwd_f32')
    return ggrid


def adam_amsgrad(p, g, m, v, vhat, lr_t, beta1, beta2, eps, eta, weight_decay_rate, grad_scale=1.0):
    """lr_t: a Python float, or a 1-element device tensor read by the kernel when it runs (graph-capturable).
    vhat=None: plain Adam (amsgrad=False)."""
    if CLASS_COUNT is not None: _acct('optimizer', 0, _nbytes(p, g, m, v, vhat), _nbytes(p, m, v, vhat))
    if vhat is None:
        dev_lr = torch.is_tensor(lr_t)
        if dev_lr:
            assert lr_t.dtype == torch.float32 and lr_t.numel() == 1 and lr_t.is_cuda
        check(_lib.load().loans_adam_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), 0.0 if dev_lr else lr_t,
                                         _ptr(lr_t) if dev_lr else 0, beta1, beta2, eps, eta, weight_decay_rate, grad_scale,
                                         _stream()), 'loans_adam_f32')
        return
    if torch.is_tensor(lr_t):
        assert lr_t.dtype == torch.float32 and lr_t.numel() == 1 and lr_t.is_cuda
        check(_lib.load().loans_adam_amsgrad_devlr_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vhat), p.numel(), _ptr(lr_t),
                                                       beta1, beta2, eps, eta, weight_decay_rate, grad_scale, _stream()),
              'loans_adam_amsgrad_devlr_f32')
        return
    check(_lib.load().loans_adam_amsgrad_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vhat), p.numel(), lr_t, beta1,
                                             beta2, eps, eta, weight_decay_rate, grad_scale, _stream()),
          'loans_adam_amsgrad_f32')
