"""CPU-side checks (no GPU): the C-ABI library exports every symbol include/loans_hip.h
declares, argument validation rejects bad shapes before any launch, and the host-side
link tree / parameter layouts / geometry are right."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from loans_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, 'include', 'loans_hip.h')).read()
    declared = set(re.findall(r'\b(loans_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 30
    from loans_amd import _lib
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared - {'loans_hip_version'} == set(_lib.SIGNATURES), 'ctypes table out of sync with the header'
    assert b'gfx950' in lib.loans_hip_version()


def test_argument_validation_without_gpu(lib):
    from loans_amd._lib import IgemmDesc
    d = IgemmDesc()
    # null pointers / zero sizes are rejected with LOANS_EINVAL (-1) before any HIP call
    assert lib.loans_igemm_f32(0, 0, 0, 0, 0, 0, 0, ctypes.byref(d), 0) == -1
    assert lib.loans_prep_images_f32(0, 0, 1, 8, 8, 0) == -1
    assert lib.loans_bn_apply_f32(0, 0, 0, 0, 0, 0, 0, 4, 64, 0, 1, 0) == -1
    assert lib.loans_adam_amsgrad_f32(0, 0, 0, 0, 0, 16, 1e-3, .9, .999, 1e-8, 1., 0., 1., 0) == -1
    d.B, d.inH, d.inW, d.Cin, d.outH, d.outW, d.Cout = 1, 8, 8, 6, 8, 8, 64      # Cin not a multiple of 4
    d.gridH = d.gridW = 8; d.osy = d.osx = d.isy = d.isx = 1; d.ntaps = 9
    assert lib.loans_igemm_f32(8, 8, 8, 0, 0, 0, 0, ctypes.byref(d), 0) == -1
    d.Cin = 8; d.B = 1 << 20; d.inH = d.inW = 1 << 10                            # beyond 32-bit indexing
    assert lib.loans_igemm_f32(8, 8, 8, 0, 0, 0, 0, ctypes.byref(d), 0) == -2


def test_conv_geometry_classes():
    from loans_amd import ops
    g = ops.ConvGeometry(2, 56, 56, 64, 128, 3, 2, 1)
    assert (g.Ho, g.Wo) == (28, 28) and g.fwd.ntaps == 9
    assert sorted(d.ntaps for d, _, _ in g.dgrad) == [1, 2, 2, 4]                # 3x3 / stride 2 parity classes
    assert g.dgrad_weight_floats == 64 * 9 * 128
    g = ops.ConvGeometry(2, 75, 75, 128, 128, 4, 2, 1)
    assert (g.Ho, g.Wo) == (37, 37) and [d.ntaps for d, _, _ in g.dgrad] == [4, 4, 4, 4]
    g = ops.ConvGeometry(2, 224, 224, 4, 64, 7, 2, 3)
    assert (g.Ho, g.Wo) == (112, 112) and g.fwd.ntaps == 49
    # every (input pixel, tap) pair of the forward conv appears exactly once over the dgrad classes
    g = ops.ConvGeometry(1, 9, 8, 4, 4, 3, 2, 1)
    pairs = set()
    for d, tapsel, _ in g.dgrad:
        for y in range(d.gridH):
            for t in range(d.ntaps):
                h = y * d.osy + d.oy0
                o = y + d.dy[t]
                r = tapsel[t] // 3
                if 0 <= o < g.Ho:
                    assert o * 2 - 1 + r == h
                    pairs.add((h, o, r))
    fw = {(o * 2 - 1 + r, o, r) for o in range(g.Ho) for r in range(3) if 0 <= o * 2 - 1 + r < 9}
    assert {(h, o, r) for (h, o, r) in pairs} == fw


def test_link_tree_matches_reference_paths_and_counts():
    import loans_amd
    np.random.seed(0)
    loc = loans_amd.SheepLocalizer((75, 75))
    keys = [k for k, _ in loc.namedparams()]
    assert '/feature_extractor/res3/0/conv3/W' in keys and '/res7/1/bn2/beta' in keys
    logical = {k: p.logical_shape for k, p in loc.namedparams()}
    n224 = sum(int(np.prod(s)) for k, s in logical.items() if not k.startswith(('/res6', '/res7')))
    assert n224 == 12592902
    assert logical['/feature_extractor/conv1/W'] == (64, 3, 7, 7)
    w = loc.feature_extractor.conv1.W
    assert w.physical_shape == (64, 7, 8, 3) and not w.host[:, :, 7].any()       # dense K rows: 8-pixel RGB windows, the 8th weightless
    np.testing.assert_array_equal(loc.param_predictor.b.host, np.array([0.8, 0, 0, 0, 0.8, 0], np.float32))
    assert not loc.param_predictor.W.host.any()
    st = loc.state_dict_chainer()
    assert st['feature_extractor/bn1/avg_var'].shape == (64,) and st['feature_extractor/conv1/W'].shape == (64, 3, 7, 7)
    # OIHW <-> OHWI(+pad) round trip
    a = np.random.standard_normal((64, 3, 7, 7)).astype(np.float32)
    w.set_logical(a)
    np.testing.assert_array_equal(w.get_logical(), a)
    dis = loans_amd.ResnetAssessor()
    assert [k for k, _ in dis.namedparams()][:3] == ['/r0/c0/W', '/r0/c1/W', '/r0/cs/W']


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'loans_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M) or 'from oracle' in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_every_entry_point_rejects_null_pointers(lib):
    """each launcher validates before it launches: with every pointer NULL (and otherwise sane scalars) it returns
    LOANS_EINVAL / LOANS_ERANGE, never a hipError and never a crash -- on a machine without a GPU"""
    from loans_amd import _lib
    descs = (_lib.IgemmDesc * 4)()        # loans_igemm_classes_f32 reads up to n = 4 of them
    desc = descs[0]
    desc.B = desc.inH = desc.inW = desc.outH = desc.outW = desc.gridH = desc.gridW = 4
    desc.Cin = desc.Cout = 8
    desc.osy = desc.osx = desc.isy = desc.isx = 1
    desc.ntaps = 1
    for i in range(1, 4):
        ctypes.memmove(ctypes.byref(descs[i]), ctypes.byref(desc), ctypes.sizeof(_lib.IgemmDesc))
    tapsel = (ctypes.c_int32 * 1)(0)
    null_table = (ctypes.c_void_p * 4)()
    for name, argtypes in sorted(_lib.SIGNATURES.items()):
        if name in _lib.RESTYPES:           # host-side planners (no pointers to reject): a size, not a status
            continue
        args = []
        for t in argtypes:
            if t is ctypes.c_void_p:
                args.append(0)
            elif t is ctypes.POINTER(_lib.IgemmDesc):
                args.append(descs)
            elif t is ctypes.POINTER(ctypes.c_void_p):
                args.append(null_table)
            elif t is ctypes.POINTER(ctypes.c_int32):
                args.append(tapsel)
            elif t is ctypes.POINTER(_lib.SmallConv):
                args.append(ctypes.byref(_lib.SmallConv(3, 1, 1, 4, 4)))
            elif t in (ctypes.c_float, ctypes.c_double):
                args.append(1.0)
            else:
                args.append(4)
        rc = getattr(lib, name)(*args)
        assert rc in (-1, -2), (name, rc)


def test_tile_selection_host_logic():
    """pure host logic of loans_amd/ops.py that decides which kernel variants are offered (no GPU): split-K candidates
    only for small grids with a long K in the fp32 arm, channel counts the finalize / reduction kernels tile, the keys
    that keep the autotune tables of differently-staged weight gradients apart"""
    import torch
    from loans_amd import ops
    assert ops._splitk_candidates(16 * 49, 512, 144) == ()        # the suite runs with LOANS_SPLITK=0 (tests/conftest.py)
    old, ops.SPLITK = ops.SPLITK, True
    try:
        assert ops._splitk_candidates(16 * 49, 512, 144) == tuple(3 | (s << 8) for s in (2, 4, 8, 16))  # res5 at B = 16
        assert ops._splitk_candidates(256 * 49, 512, 144) == ()                                           # ... at B = 256
        assert ops._splitk_candidates(16 * 49, 512, 8) == ()                                              # short K
        assert all((t >> 8) * 4 <= 18 for t in ops._splitk_candidates(49, 64, 18))                        # >= 4 chunks per slice
        ops.set_compute_dtype('bf16')
        try:
            assert ops._splitk_candidates(16 * 49, 512, 144) == ()                                        # fp32 arm only
        finally:
            ops.set_compute_dtype('f32')
    finally:
        ops.SPLITK = old
    for c, ok in ((4, True), (64, True), (96, False), (1024, True), (2048, True), (1536, False), (6, False)):
        assert ops.reduce_channels_ok(c) is ok, c
    f32, b16 = torch.zeros(1), torch.zeros(1, dtype=torch.bfloat16)
    keys = {ops._wgrad_key(f32, f32, False), ops._wgrad_key(f32, f32, True), ops._wgrad_key(b16, b16, False),
            ops._wgrad_key(b16, b16, True), ops._wgrad_key(f32, b16, False)}
    assert len(keys) == 5
    with __import__('pytest').raises(ValueError):
        ops.set_storage_dtype('bf16')            # bf16 storage needs the bf16 compute arm first


def test_round2_tile_offers_host_logic():
    """which of the round-2 kernel forms the autotuner is offered, from shapes alone (no GPU): the one-launch class form of a
    strided fp32 data gradient, the 256-column / deep-ring / halo / weight-stationary tiles and the stacked pair of the bf16 arm,
    the direct conv1 kernels -- each mirrors a launcher-side condition of csrc/*.hip that would answer LOANS_EINVAL"""
    import torch
    from loans_amd import ops
    g32 = ops.ConvGeometry(4, 56, 56, 64, 128, 3, 2, 1)           # four parity classes, at most four taps each
    g11 = ops.ConvGeometry(4, 56, 56, 64, 128, 1, 2, 0)           # 1x1 / 2: a single class with taps
    g33 = ops.ConvGeometry(4, 57, 57, 64, 64, 3, 3, 1)            # stride 3: nine classes, more than the launcher takes
    assert all(t & ops.TILE_CLASSES for t in ops._class_candidates(g32)) and len(ops._class_candidates(g32)) >= 4
    assert ops._class_candidates(g11) == () and ops._class_candidates(g33) == ()
    ops.set_compute_dtype('bf16')
    try:
        assert ops._class_candidates(g32) == ()                   # fp32 arm only
    finally:
        ops.set_compute_dtype('f32')
    # 256 columns: the 256 x 256 tile, its ping-pong form (round 6) and that on 16 x 16 x 32 MFMAs
    wide = (ops.TILE_256x256, ops.TILE_256x256PP, ops.TILE_256x256PP16)
    assert ops._wide16_tiles(512) == wide and ops._wide16_tiles(128) == ()
    deep = ops._wide16_tiles(512, rows=2048)                      # res7 at 512 px: 32 x 8 tiles of 64 x 64
    assert set(deep) == set(wide) | {1 | ops.TILE_DEEP, 2 | ops.TILE_DEEP, 3 | ops.TILE_DEEP}
    assert ops._wide16_tiles(512, rows=128 * 32 * 32) == wide                     # res4: a grid that fills the machine
    res4 = ops.ConvGeometry(2, 32, 32, 256, 256, 3, 1, 1)
    assert ops.TILE_HALO_256x256 in ops._halo_tiles(res4, 256, 256, (32, 32))     # 16 x 16 pixels x 256 channels (round 6)
    assert ops.TILE_HALO_256x256 not in ops._halo_tiles(res4, 256, 128, (32, 32))
    res2 = ops.ConvGeometry(2, 128, 128, 64, 64, 3, 1, 1)
    t = ops._halo_tiles(res2, 64, 64, (128, 128))
    assert ops.TILE_WS64 in t and ops.TILE_WSW64 in t and ops.TILE_HALO_256x128 not in t
    assert ops.TILE_WSW64 not in ops._halo_tiles(res2, 64, 64, (128, 128), relu_in=True)     # no pre-activation form
    res3 = ops.ConvGeometry(2, 64, 64, 128, 128, 3, 1, 1)
    assert ops.TILE_HALO_256x128 in ops._halo_tiles(res3, 128, 128, (64, 64)) and ops.TILE_WSW64 not in ops._halo_tiles(res3, 128, 128, (64, 64))
    assert ops._halo_tiles(g32, 64, 128, (28, 28)) == ()          # strided: no halo form
    x16, x32 = torch.zeros(1, dtype=torch.bfloat16), torch.zeros(1)
    a, b = ops.ConvGeometry(2, 32, 32, 64, 128, 3, 2, 1), ops.ConvGeometry(2, 32, 32, 64, 128, 3, 2, 1)
    assert ops.fprop_pair_ok(x16, a, b) and ops.fprop_pair_ok(x32, a, b)
    s1a, s1b = ops.ConvGeometry(2, 32, 32, 64, 64, 3, 1, 1), ops.ConvGeometry(2, 32, 32, 64, 64, 3, 1, 1)
    assert not ops.fprop_pair_ok(x16, s1a, s1b) and ops.fprop_pair_ok(x32, s1a, s1b)       # bf16: strided units only
    assert not ops.fprop_pair_ok(x16, a, ops.ConvGeometry(2, 32, 32, 64, 256, 3, 2, 1))    # bf16: equal channel counts
    stem224 = ops.ConvGeometry(2, 224, 224, 3, 64, 7, 2, 3, dense=True)
    stem512 = ops.ConvGeometry(2, 512, 512, 3, 64, 7, 2, 3, dense=True)
    assert ops.stem_tile_rows(stem224) == 4 and ops.stem16_tile_rows(stem224) == 4 and ops.stem16_tile_rows(stem512) == 4
    assert ops.stem_wgrad_ok(stem224) and not ops.stem_wgrad_ok(stem512)          # two unit buffers of a 512 px row: 215 KB
    odd = ops.ConvGeometry(1, 18, 23, 3, 64, 7, 2, 3, dense=True)            # 9 x 12 output pixels: no whole 64-pixel tile
    assert ops.stem_tile_rows(odd) == 0 and ops.stem16_tile_rows(odd) == 1 and ops.stem_wgrad_ok(odd)
    odd_rows = ops.ConvGeometry(1, 17, 23, 3, 64, 7, 2, 3, dense=True)       # an odd padded height: runs are not 16-byte aligned
    assert ops.stem16_tile_rows(odd_rows) == 0 and not ops.stem_wgrad_ok(odd_rows)


def test_launcher_state_is_per_device_and_capture_safe():
    """include/loans_hip.h promises: re-entrant, thread-safe given distinct streams, device = the caller's current device,
    no blocking runtime call.  So the launchers may keep no process-wide mutable state: what they cache (a kernel's raised
    dynamic-LDS limit, the CU count) is keyed by the device ordinal in lock-free tables (csrc/common.h)."""
    header = open(os.path.join(ROOT, 'include', 'loans_hip.h')).read()
    assert 'per kernel AND per device' in header and 'hipGraph stream capture' in header
    csrc = os.path.join(ROOT, 'loans_amd', 'csrc')
    common = open(os.path.join(csrc, 'common.h')).read()
    assert 'std::atomic<uint64_t> bits' in common and 'hipDeviceGetAttribute' in common
    for name in sorted(os.listdir(csrc)):
        if not name.endswith('.hip'):
            continue
        src = open(os.path.join(csrc, name)).read()
        # function-local mutable statics other than the per-device tables; `static const` / `static constexpr` / functions are fine
        bad = re.findall(r'^\s+static\s+(?!const\b|constexpr\b|loans_device_once\b|__device__|inline\b|std::atomic)[\w:<>]+\s+\w+\s*(?:=[^;]*)?;',
                         src, flags=re.M)
        assert not bad, (name, bad)
        assert 'hipGetDeviceProperties' not in src, name            # blocking, not capture-safe
        assert src.count('hipFuncSetAttribute') == 0, name          # only through loans_raise_lds_limit (per device)


def test_wgrad_workspace_a_capture_has_seen_is_never_freed(monkeypatch):
    """the slab workspace of the bf16 weight gradients is regrown by replacing the slot's tensor.  A hipGraph captured
    earlier has the OLD tensor's address baked in, so a workspace handed out during capture must outlive its replacement (the
    replayed graph would otherwise store slabs into memory the allocator re-uses).  Host logic, on CPU tensors."""
    import types
    import torch
    from loans_amd import ops
    need = {'n': 1 << 22}
    lib = types.SimpleNamespace(loans_wgrad_bf16s_ws_floats=lambda desc, splits: need['n'])
    capturing = {'on': False}
    monkeypatch.setattr(torch.cuda, 'is_current_stream_capturing', lambda: capturing['on'])
    monkeypatch.setattr(ops, '_stream', lambda: 7)
    monkeypatch.setattr(ops, '_wgrad_ws', {})
    monkeypatch.setattr(ops, '_wgrad_ws_captured', [])
    monkeypatch.setattr(ops.C, 'byref', lambda d: d)
    dev = torch.device('cpu')
    geo = lambda: types.SimpleNamespace()                 # a fresh geometry per call: no cached need
    ws0, n0 = ops._wgrad_workspace(lib, geo(), None, 1, 4, dev, 7)            # eager: allocated
    assert n0 == 1 << 22 and not ops._wgrad_ws_captured
    capturing['on'] = True
    ws1, _ = ops._wgrad_workspace(lib, geo(), None, 1, 4, dev, 7)             # captured: the same tensor, now pinned
    assert ws1 is ws0 and ops._wgrad_ws_captured == [ws0]
    need['n'] = 1 << 23
    assert ops._wgrad_workspace(lib, geo(), None, 1, 8, dev, 7) is None       # never grows inside a capture
    capturing['on'] = False
    ws2, n2 = ops._wgrad_workspace(lib, geo(), None, 1, 8, dev, 7)            # a later eager step with a larger need
    assert ws2 is not ws0 and ws2.numel() >= n2 == 1 << 23
    assert any(b is ws0 for b in ops._wgrad_ws_captured)                      # the graph's workspace is still alive
    assert ws0.data_ptr() != ws2.data_ptr()


def test_precision_scope_is_loud_about_a_second_thread():
    """ops.precision sets process-wide globals.  A second thread asking for ANOTHER arithmetic while a scope is open used
    to run (or make the owner run) in the wrong one silently; now it raises.  The same arithmetic from another thread, nesting on
    the owner's thread and the restore on exit keep working."""
    import threading
    from loans_amd import ops
    seen = {}

    def other():
        try:
            with ops.precision('f32'):
                seen['f32'] = 'entered'
        except RuntimeError as e:
            seen['f32'] = str(e)
        with ops.precision('bf16'):
            seen['bf16'] = ops.current_precision()
    base = ops.current_precision()
    with ops.precision('bf16'):
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert 'second thread' in seen['f32'] and seen['bf16'] == ('bf16', 'bf16')
        with ops.precision('f32'):
            assert ops.current_precision() == ('f32', 'f32')
        assert ops.current_precision() == ('bf16', 'bf16')
    assert ops.current_precision() == base and ops.precision._depth == 0 and ops.precision._owner is None


def test_tune_table_round_trip(tmp_path, monkeypatch):
    """A normal run writes its tile table, the profiled runs of the same command read it: same kernels in both
    (ops.save_tune_table / load_tune_table, bench.py --tune-file).  What a file holds is a PROPOSAL: it is launched only if
    the problem's current candidate list still offers that tile (ablation switches, another library revision), and only on
    the chip / tile-id schema the table was stamped with."""
    import json
    from loans_amd import ops
    g = ops.ConvGeometry(3, 20, 24, 8, 16, 3, 1, 1)
    gd = ops.ConvGeometry(3, 32, 32, 3, 64, 7, 2, 3, dense=True)
    g.tuned.update({'f32fprop_stats': 19, 'f32wgrad': 3 | (7 << 8)})
    gd.tuned['f32fprop_stats_st'] = 10
    path = str(tmp_path / 'tune.json')
    assert ops.save_tune_table(path) >= 2
    assert json.load(open(path))['stamp']['schema'] == ops.TUNE_SCHEMA
    g.tuned.clear(); gd.tuned.clear(); ops._TUNE_LOADED.clear()
    assert ops.load_tune_table(path) >= 2
    assert g.tuned == {} and ops._TUNE_LOADED[g.key] == {'f32fprop_stats': 19, 'f32wgrad': 3 | (7 << 8)}
    ran = []
    monkeypatch.setattr(ops, 'TUNE_POLICY', 'time')                   # (the test session's default is 'fixed', conftest.py)
    monkeypatch.setattr(ops, '_time_call', lambda fn, reps=5, cold=False: (fn(), float(len(ran)))[1])
    run = ran.append
    # on offer: taken without timing anything
    assert ops._tuned_tile(g, 'f32fprop_stats', run, (1, 2, 19)) == 19 and not ran
    assert ops._tuned_tile(gd, 'f32fprop_stats_st', run, (1, 10)) == 10 and not ran
    # not on offer any more (an ablation switch removed the tile): tuned afresh among the candidates
    assert ops._tuned_tile(g, 'f32wgrad', run, (1, 3)) == 1 and ran == [1, 3]
    g2 = ops.ConvGeometry(3, 20, 24, 8, 16, 3, 1, 1)                  # a geometry made later shares the entry
    assert g2.tuned is g.tuned and g2.tuned['f32fprop_stats'] == 19
    g.tuned['f32fprop_stats'] = 1                                     # entries tuned in this process win over the file
    ops.load_tune_table(path)
    assert ops._tuned_tile(g, 'f32fprop_stats', run, (1, 2, 19)) == 1
    # a table stamped for another chip, and one from before the stamps, are ignored
    doc = json.load(open(path))
    doc['stamp']['compute_units'] = 7
    json.dump(doc, open(path, 'w'))
    ops._TUNE_LOADED.clear()
    assert ops.load_tune_table(path) == 0 and not ops._TUNE_LOADED
    del doc['stamp']
    json.dump(doc, open(path, 'w'))
    assert ops.load_tune_table(path) == 0 and not ops._TUNE_LOADED
    g.tuned.clear(); gd.tuned.clear()


def test_weight_gradient_stream_count_follows_the_arm(monkeypatch):
    """ops.wgrad_streams (round 3): two streams while the fp32 kernels are selected (their launches end in ragged rounds the
    next launch can run under), one on the bf16 arms (measured slower with two); LOANS_WGRAD_STREAMS overrides, clamped to 1..4."""
    from loans_amd import ops
    old = ops.COMPUTE
    try:
        monkeypatch.setattr(ops, '_WGRAD_STREAMS_ENV', '')
        ops.set_compute_dtype('f32')
        assert ops.wgrad_streams() == 2
        ops.set_compute_dtype('bf16')
        assert ops.wgrad_streams() == 1
        for env, want in (('1', 1), ('3', 3), ('9', 4), ('0', 1)):
            monkeypatch.setattr(ops, '_WGRAD_STREAMS_ENV', env)
            assert ops.wgrad_streams() == want
    finally:
        ops.set_compute_dtype(old)


def test_gpu_suite_is_collected_in_order_of_importance():
    """`pytest -m gpu -x` stops at the first failure, so what it collects first is what is certainly graded: the fp32 kernels,
    then the fp32 model + golden fixtures, configs[1] at full size, the data-parallel tests, the callers -- and the statistical
    bf16 comparisons last (tests/conftest.py: SUITE_ORDER)."""
    import glob
    import subprocess
    import sys
    from tests import conftest
    out = subprocess.run([sys.executable, '-m', 'pytest', 'tests', '--collect-only', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600).stdout
    files = []
    for line in out.splitlines():
        if '::' in line:
            f = os.path.splitext(os.path.basename(line.split('::')[0]))[0]
            if not files or files[-1] != f:
                files.append(f)
    assert len(files) == len(set(files)), files                   # every file's tests are contiguous
    assert files == [f for f in conftest.SUITE_ORDER if f in files], files
    assert files[0] == 'test_gpu_kernels' and files[-1] == 'test_gpu_configs'
    assert files.index('test_gpu_fullsize') < files.index('test_gpu_parallel') < files.index('test_gpu_bf16_storage')
    # no GPU test file outside the order
    on_disk = {os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(ROOT, 'tests', 'test_gpu_*.py'))}
    assert on_disk <= set(conftest.SUITE_ORDER), on_disk - set(conftest.SUITE_ORDER)
    assert set(files) >= on_disk


def test_fixed_tile_policy_is_deterministic_and_never_times(monkeypatch):
    """ops.TUNE_POLICY = 'fixed' (the parity session): a shape's tile is a pure function of (shape, mode, salt) taken from the
    candidates on offer, nothing is launched to decide it, a loaded table's proposal still wins."""
    from loans_amd import ops
    monkeypatch.setattr(ops, 'TUNE_POLICY', 'fixed')
    monkeypatch.setattr(ops, 'AUTOTUNE', True)
    ran = []
    run = lambda t: ran.append(t)                                   # noqa: E731
    before = ops.TIMED_PICKS
    picks = {}
    for B in range(1, 40):
        g = ops.ConvGeometry(B, 20, 20, 64, 64, 3, 1, 1)
        g.tuned.clear()
        picks[B] = ops._tuned_tile(g, 'f32fprop', run, (1, 2, 3, 17, 18, 19))
        assert picks[B] in (1, 2, 3, 17, 18, 19)
        g.tuned.clear()
        assert ops._tuned_tile(g, 'f32fprop', run, (1, 2, 3, 17, 18, 19)) == picks[B]
        g.tuned.clear()
    assert not ran and ops.TIMED_PICKS == before
    assert len(set(picks.values())) >= 4                            # over many shapes the picks cover the candidates
    monkeypatch.setattr(ops, 'TUNE_SALT', '1')
    g = ops.ConvGeometry(7, 20, 20, 64, 64, 3, 1, 1)
    other = [ops._fixed_pick(ops.ConvGeometry(B, 20, 20, 64, 64, 3, 1, 1), 'f32fprop', (1, 2, 3, 17, 18, 19)) for B in range(1, 40)]
    assert other != [picks[B] for B in range(1, 40)]                # another salt, another assignment
    ops._TUNE_LOADED[g.key] = {'f32fprop': 18}
    try:
        assert ops._tuned_tile(g, 'f32fprop', run, (1, 2, 3, 17, 18, 19)) == 18
    finally:
        ops._TUNE_LOADED.pop(g.key)
        g.tuned.clear()


def test_resolved_variants_live_with_the_tile_picks_and_stay_out_of_tables(tmp_path, monkeypatch):
    """ops._variant (round 4): a convolution wrapper keeps the tile it resolved for one call variant under a '~' key of
    geo.tuned, so that tuned calls build no candidate lists.  Those keys depend on the module's switches (another switch
    state resolves afresh), die with the shape's picks, and are never written to a tile table."""
    import json
    from loans_amd import ops
    g = ops.ConvGeometry(5, 20, 24, 64, 64, 3, 1, 1)
    g.tuned.clear()
    k1 = ops._variant(g, 'fprop16', True, False, True, False)
    assert k1.startswith('~') and k1 == ops._variant(g, 'fprop16', True, False, True, False)
    assert k1 != ops._variant(g, 'fprop16', False, False, True, False)
    monkeypatch.setattr(ops, 'SPLITK', not ops.SPLITK)
    assert ops._variant(g, 'fprop16', True, False, True, False) != k1           # another switch state: resolved afresh
    monkeypatch.undo()
    g.tuned.update({k1: 7, 'bf16s_fprop_stats': 7})
    path = str(tmp_path / 'tune.json')
    ops.save_tune_table(path)
    saved = json.load(open(path))['entries'][ops._tune_key_str(g.key)]
    assert saved == {'bf16s_fprop_stats': 7}                                     # the variant key stays in the process
    g.tuned.clear()                                                              # (what conftest's timed_autotune fixture does)
    assert ops.ConvGeometry(5, 20, 24, 64, 64, 3, 1, 1).tuned == {}
    # the memoised candidate functions follow the switches too
    a = ops._class_candidates(ops.ConvGeometry(4, 56, 56, 64, 128, 3, 2, 1))
    monkeypatch.setattr(ops, 'CLASS_LAUNCH', False)
    assert a and ops._class_candidates(ops.ConvGeometry(4, 56, 56, 64, 128, 3, 2, 1)) == ()


def test_pointwise_tile_offer_host_logic(monkeypatch):
    """LOANS_TILE_PW is offered exactly where loans_pw16_covers (csrc/pw_bf16.hip) accepts the problem: a 1 x 1 / 1 convolution
    without padding, no epilogue beyond the BN statistics, Cin 64 / 128 with Cout a multiple of 64 up to 512, or Cin 256 with Cout a
    multiple of 128 up to 1024 -- and nowhere else (a tile the launcher refuses would fail the autotuner's timing run)."""
    from loans_amd import ops
    G = ops.ConvGeometry
    yes = [G(2, 16, 16, 64, 256, 1, 1, 0), G(1, 5, 7, 64, 64, 1, 1, 0), G(2, 8, 8, 64, 512, 1, 1, 0), G(2, 8, 8, 128, 512, 1, 1, 0),
           G(2, 8, 8, 128, 192, 1, 1, 0), G(2, 8, 8, 256, 1024, 1, 1, 0), G(2, 8, 8, 256, 128, 1, 1, 0)]
    no = [G(2, 16, 16, 64, 256, 3, 1, 1),         # not 1 x 1
          G(2, 16, 16, 64, 256, 1, 2, 0),         # strided
          G(2, 16, 16, 64, 96, 1, 1, 0),          # Cout not a multiple of 64
          G(2, 16, 16, 64, 576, 1, 1, 0),         # Cout beyond the slab
          G(2, 16, 16, 32, 128, 1, 1, 0), G(2, 16, 16, 512, 2048, 1, 1, 0),       # Cin the kernels are not built for
          G(2, 16, 16, 256, 192, 1, 1, 0), G(2, 16, 16, 256, 2048, 1, 1, 0)]       # Cin = 256: 128-column phases, Cout <= 1024
    for g in yes:
        assert ops._pw_tiles(g, True) == (ops.TILE_PW,), g.key
        assert ops._pw_tiles(g, False) == (), g.key               # ReLU on the input / bias / addend: not this tile's epilogue
    for g in no:
        assert ops._pw_tiles(g, True) == (), g.key
    monkeypatch.setattr(ops, 'PW', False)                         # LOANS_PW=0 (the switch is part of the memo key)
    assert ops._pw_tiles(yes[0], True) == ()


def test_struct_mirrors_have_the_headers_layout(tmp_path):
    """Every struct of include/loans_hip.h that crosses the C ABI has a Python mirror (ctypes.Structure in loans_amd/_lib.py, a NumPy
    record dtype for the resampling jobs): sizes and field offsets are compared with what a C compiler makes of the header."""
    import ctypes
    import subprocess
    from loans_amd import _lib
    from loans_amd.common.datasets.resample import RESAMPLE_JOB
    mirrors = {'loans_igemm_desc': _lib.IgemmDesc, 'loans_repack_job': _lib.RepackJob, 'loans_pw_pack_job': _lib.PwPackJob,
               'loans_small_conv': _lib.SmallConv}
    probes = [('loans_%s' % n.split('loans_')[1], f) for n, m in mirrors.items() for f, _ in m._fields_]
    job_fields = [f for f in RESAMPLE_JOB.names if not f.startswith('_')]         # ('_pad' = the struct's tail padding)
    probes += [('loans_resample_job', f) for f in job_fields]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "loans_hip.h"', 'int main(void) {']
    for s in list(mirrors) + ['loans_resample_job']:
        src.append('  printf("%s %%zu\\n", sizeof(%s));' % (s, s))
    for s, f in probes:
        src.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (s, f, s, f))
    src += ['  return 0;', '}']
    c_file, exe = tmp_path / 'layout.c', tmp_path / 'layout'
    c_file.write_text('\n'.join(src))
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', str(exe), str(c_file)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for s, m in mirrors.items():
        assert int(got[s]) == ctypes.sizeof(m), s
        for f, _ in m._fields_:
            assert int(got['%s.%s' % (s, f)]) == getattr(m, f).offset, (s, f)
    assert int(got['loans_resample_job']) == RESAMPLE_JOB.itemsize
    for f in job_fields:
        assert int(got['loans_resample_job.%s' % f]) == RESAMPLE_JOB.fields[f][1], f


# every environment switch of the product, and the test that runs its non-default arm ("<= 15 environment
# switches, each with a test of its non-default arm").  The LOANS_*DBG* reads of csrc/ only exist in LOANS_EXPERIMENT /
# LOANS_STAMPS builds (tools/), the LOANS_BENCH_* ones belong to bench.py's own tests (tests/test_bench_cpu.py).
ENV_SWITCHES = {
    'LOANS_CONCURRENT_CHAINS': 'tests/test_gpu_model.py::test_update_core_gradients_and_parameters_parity',
    'LOANS_EARLY_CHAIN': 'tests/test_gpu_model.py::test_update_core_gradients_and_parameters_parity',
    'LOANS_SPLITK': 'tests/test_gpu_model.py::test_split_k_autotuned_step',
    'LOANS_TUNE_FILE': 'tests/test_host_cpu.py::test_tune_file_named_by_the_environment_is_loaded_at_import',
    'LOANS_TUNE_POLICY': 'tests/conftest.py',
    'LOANS_TUNE_SALT': 'tests/test_host_cpu.py::test_fixed_tile_policy_is_deterministic_and_never_times',
    'LOANS_WGRAD_SLABS': 'tests/test_gpu_bf16_storage.py::test_weight_gradient_slabs_are_deterministic_and_equal_the_atomic_form',
    'LOANS_STEP_ARENA': 'tests/test_gpu_model.py::test_step_workspace_serves_every_request_and_changes_nothing',
    'LOANS_DIST_SELFTEST': 'tests/test_gpu_launch.py::test_bench_rccl_at_world_size_one',
    'LOANS_DIST_BACKEND': 'tests/test_launch_cpu.py',
    'LOANS_CONV_NT_MB': 'tests/test_gpu_nontemporal.py',
    'LOANS_BN_NT': 'tests/test_gpu_nontemporal.py',
}


def test_environment_switches_are_few_and_each_has_a_test():
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = set()
    for path in glob.glob(os.path.join(root, 'loans_amd', '**', '*.py'), recursive=True) + \
            [os.path.join(root, n) for n in ('train_sheep_localizer.py', 'evaluate.py')]:
        found |= set(re.findall(r"environ(?:\.get\(|\[)\s*'(LOANS_[A-Z0-9_]+)'", open(path).read()))
    for path in glob.glob(os.path.join(root, 'loans_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'loans_amd', 'csrc', '*.h')):
        guarded = 0
        for line in open(path):
            if re.match(r'\s*#\s*if(def)?\s.*LOANS_(EXPERIMENT|STAMPS)', line):
                guarded += 1
            elif guarded and re.match(r'\s*#\s*if', line):
                guarded += 1
            elif guarded and re.match(r'\s*#\s*endif', line):
                guarded -= 1
            elif not guarded:
                found |= set(re.findall(r'getenv\("(LOANS_[A-Z0-9_]+)"\)', line))
    assert found == set(ENV_SWITCHES), (sorted(found - set(ENV_SWITCHES)), sorted(set(ENV_SWITCHES) - found))
    assert len(found) <= 15
    for name, where in ENV_SWITCHES.items():
        path, _, fn = where.partition('::')
        text = open(os.path.join(root, path)).read()
        assert name.replace('LOANS_', '') in text or name in text, (name, 'is not mentioned in', path)
        if fn:
            assert re.search(r'def %s\b' % fn, text), (fn, 'is not in', path)


def test_tune_file_named_by_the_environment_is_loaded_at_import(tmp_path):
    """LOANS_TUNE_FILE: a table written by ops.save_tune_table is in force from the import on"""
    import json
    import subprocess
    import sys
    from loans_amd import ops
    geo = ops.ConvGeometry(2, 16, 16, 64, 64, 3, 1, 1)
    path = str(tmp_path / 'tune.json')
    old = dict(ops._TUNE_CACHE)
    try:
        geo.tuned['f32fprop'] = 3               # (geo.tuned IS the cache's entry of this shape)
        ops.save_tune_table(path)
    finally:
        ops._TUNE_CACHE.clear()
        ops._TUNE_CACHE.update(old)
    assert json.load(open(path))['entries']
    code = ("from loans_amd import ops; g = ops.ConvGeometry(2, 16, 16, 64, 64, 3, 1, 1); "
            "print(ops._TUNE_LOADED.get(g.key, {}).get('f32fprop'))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env_value, want in ((path, '3'), ('', 'None')):
        env = dict(os.environ, LOANS_TUNE_FILE=env_value, PYTHONPATH=root)
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.strip().splitlines()[-1] == want, (env_value, out.stdout)
