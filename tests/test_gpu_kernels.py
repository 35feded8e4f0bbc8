"""-m gpu: every HIP kernel family against the CPU oracle, through the C ABI
(loans_amd.ops is a 1:1 ctypes wrapper of include/loans_hip.h).  fp32 tolerances are
written next to each check; integer/index results (argmax, uint8 truncation) are exact."""
import numpy as np
import pytest
import torch

from oracle import chainer_ops as C
from tests.gpu_util import dev, rel_err

pytestmark = pytest.mark.gpu


def _nhwc(x, cpad=None):
    x = np.transpose(x, (0, 2, 3, 1))
    if cpad and x.shape[-1] < cpad:
        x = np.concatenate([x, np.zeros(x.shape[:-1] + (cpad - x.shape[-1],), x.dtype)], axis=-1)
    return np.ascontiguousarray(x)


def _nchw(t, c=None):
    a = t.detach().cpu().numpy().transpose(0, 3, 1, 2)
    return a if c is None else a[:, :c]


def _ohwi(w, cpad=None):
    return _nhwc(w, cpad)


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad
    (2, 3, 32, 32, 64, 7, 2, 3),      # stem
    (3, 64, 14, 14, 64, 3, 1, 1),     # res2-like
    (2, 64, 15, 13, 128, 3, 2, 1),    # strided, odd sizes
    (2, 128, 9, 9, 128, 4, 2, 1),     # assessor 4x4/2
    (2, 3, 19, 19, 128, 4, 2, 1),     # assessor cs on rgb
    (1, 256, 7, 7, 512, 3, 2, 1),     # res5 entry
    (5, 128, 6, 6, 200, 3, 1, 1),     # Cout not a multiple of the tile
    (2, 64, 9, 10, 128, 1, 2, 0),     # 1x1 / stride 2 (bottleneck shortcut): three of four dgrad classes have no tap
    (3, 256, 7, 7, 64, 1, 1, 0),      # 1x1 bottleneck reduce
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 17, 18, 19, 20, 22])     # +16 = LOANS_TILE_DMA
def test_conv_fprop_dgrad_wgrad(case, tile):
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    rng = np.random.RandomState(hash(case) % 1000)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    cp = (Cin + 3) // 4 * 4
    geo = ops.ConvGeometry(B, H, W, cp, Cout, k, s, p)
    xd, wd, bd = dev(_nhwc(x, cp)), dev(_ohwi(w, cp)), dev(b)

    y_ref, col = C.conv2d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s, p)
    stats_r = ops.stats_buffer(Cout, 'cuda')
    itile = tile if tile != 5 else 0        # tile 5 exists for wgrad only
    y = ops.conv_fprop(xd, wd, geo, bias=bd, stats=stats_r, tile=itile)
    stats = stats_r.sum(dim=0)
    assert rel_err(_nchw(y), y_ref) < 2e-6          # exact-f32 MFMA chain vs f64
    np.testing.assert_allclose(stats[0].cpu().numpy(), y_ref.sum(axis=(0, 2, 3)), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(stats[1].cpu().numpy(), (y_ref ** 2).sum(axis=(0, 2, 3)), rtol=1e-5)

    # relu_in + addend epilogue
    add = rng.standard_normal(y_ref.shape).astype(np.float32)
    y2_ref, _ = C.conv2d_fwd(np.maximum(x, 0).astype(np.float64), w.astype(np.float64), None, s, p)
    y2 = ops.conv_fprop(xd, wd, geo, relu_in=True, addend=dev(_nhwc(add)), tile=itile)
    assert rel_err(_nchw(y2), y2_ref + add) < 2e-6

    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    gx_ref, gw_ref, _ = C.conv2d_bwd(x.shape, col, w.astype(np.float64), gy.astype(np.float64), s, p, False)
    gyd = dev(_nhwc(gy))
    if tile in (0, 1, 2, 3, 4, 6, 17, 18, 19, 20, 22):
        gx = ops.conv_dgrad(gyd, wd, geo, tile=tile)
        assert rel_err(_nchw(gx, Cin), gx_ref) < 2e-6
        if geo.dgrad_has_empty_class:
            addx = rng.standard_normal(x.shape).astype(np.float32)
            gx2 = ops.conv_dgrad(gyd, wd, geo, addend=dev(_nhwc(addx, cp)), tile=tile)
            assert rel_err(_nchw(gx2, Cin), gx_ref + addx) < 2e-6
            acc = dev(_nhwc(addx, cp))
            ops.conv_dgrad(gyd, wd, geo, out=acc, addend=acc, tile=tile)      # in-place accumulate
            assert rel_err(_nchw(acc, Cin), gx_ref + addx) < 2e-6
            return_early = True
        else:
            return_early = False
        # mask + addend epilogues
        ref_t = rng.standard_normal(x.shape).astype(np.float32)
        addx = rng.standard_normal(x.shape).astype(np.float32)
        if not return_early:
            gx2 = ops.conv_dgrad(gyd, wd, geo, mask_ref=dev(_nhwc(ref_t, cp)), addend=dev(_nhwc(addx, cp)), tile=tile)
            assert rel_err(_nchw(gx2, Cin), gx_ref * (ref_t > 0) + addx) < 2e-6
            gx3 = ops.conv_dgrad(gyd, wd, geo, addend=dev(_nhwc(addx, cp)), addend_mask_ref=dev(_nhwc(ref_t, cp)), tile=tile)
            assert rel_err(_nchw(gx3, Cin), gx_ref + addx * (ref_t > 0)) < 2e-6
    if tile in (0, 1, 3, 5):
        dw = torch.zeros_like(wd)
        ops.conv_wgrad(xd, gyd, dw, geo, tile=tile)
        ops.join_side_stream()
        got = dw.cpu().numpy().transpose(0, 3, 1, 2)[:, :Cin]
        assert rel_err(got, gw_ref) < 5e-6
        ops.conv_wgrad(xd, gyd, dw, geo, splits=3, tile=tile)       # accumulates
        ops.join_side_stream()
        got = dw.cpu().numpy().transpose(0, 3, 1, 2)[:, :Cin]
        assert rel_err(got, 2 * gw_ref) < 5e-6


@pytest.mark.parametrize("case", [(2, 64, 15, 13, 128, 3, 2, 1), (3, 128, 9, 9, 128, 4, 2, 1), (1, 256, 7, 7, 512, 3, 2, 1),
                                  (2, 64, 56, 56, 128, 3, 2, 1), (2, 64, 9, 10, 128, 1, 2, 0), (2, 32, 11, 11, 64, 3, 3, 1)])
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 17, 18, 19, 20])
def test_conv_dgrad_class_launch(case, tile):
    """loans_igemm_classes_f32: the stride-parity classes of a strided data gradient in one grid.  Every block contracts its
    class's K in the order of the per-class launch, so the two are bit-identical -- plain, masked and with addends; the
    per-class path itself is checked against the oracle above."""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    rng = np.random.RandomState(7)
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    w = dev((rng.standard_normal((Cout, k, k, Cin)) / np.sqrt(Cin * k * k)).astype(np.float32))
    gy = dev(rng.standard_normal((B, geo.Ho, geo.Wo, Cout)).astype(np.float32))
    ref_t = dev(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
    add = dev(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
    if len(geo.dgrad) > 4:          # LOANS_MAX_CLASSES: stride 3 has nine classes, no convolution of this path has
        from loans_amd._lib import HipKernelError
        with pytest.raises(HipKernelError):
            ops.conv_dgrad(gy, w, geo, tile=tile | ops.TILE_CLASSES)
        return
    kinds = [dict(), dict(addend=add)]
    if not geo.dgrad_has_empty_class:
        kinds += [dict(mask_ref=ref_t), dict(mask_ref=ref_t, addend=add), dict(addend=add, addend_mask_ref=ref_t)]
    for kw in kinds:
        one = ops.conv_dgrad(gy, w, geo, tile=tile | ops.TILE_CLASSES, **kw)
        per = ops.conv_dgrad(gy, w, geo, tile=tile, **kw)
        assert torch.equal(one, per), (case, tile, sorted(kw))
    acc_a, acc_b = add.clone(), add.clone()
    ops.conv_dgrad(gy, w, geo, out=acc_a, addend=acc_a, tile=tile | ops.TILE_CLASSES)         # in-place accumulate
    ops.conv_dgrad(gy, w, geo, out=acc_b, addend=acc_b, tile=tile)
    assert torch.equal(acc_a, acc_b)


def test_conv_dgrad_class_launch_rejects():
    """classes that disagree in a shared field, too many taps, a tile shape without a class form: LOANS_EINVAL, no launch"""
    import ctypes as Ct
    from loans_amd import _lib, ops
    lib = _lib.load()
    geo = ops.ConvGeometry(2, 16, 16, 64, 64, 3, 2, 1)
    gy = torch.zeros((2, 8, 8, 64), device='cuda')
    wp = torch.zeros(geo.dgrad_weight_floats, device='cuda')
    out = torch.zeros((2, 16, 16, 64), device='cuda')
    n = len(geo.dgrad)

    def call(mut=None, tile=3):
        descs = (_lib.IgemmDesc * n)()
        ws = (Ct.c_void_p * n)()
        for i, (d, _, off) in enumerate(geo.dgrad):
            Ct.memmove(Ct.byref(descs[i]), Ct.byref(d), Ct.sizeof(_lib.IgemmDesc))
            descs[i].flags, descs[i].tile = 0, tile
            ws[i] = wp[off:].data_ptr()
        if mut:
            mut(descs, ws)
        return lib.loans_igemm_classes_f32(gy.data_ptr(), ws, out.data_ptr(), None, None, descs, n, None)

    assert call() == 0
    assert call(tile=6) == -1 and call(tile=8) == -1 and call(tile=3 | (2 << 8)) == -1

    def other_cout(descs, ws):
        descs[1].Cout = 32
    assert call(other_cout) == -1

    def other_flags(descs, ws):
        descs[2].flags = _lib.F_RELU_IN
    assert call(other_flags) == -1

    def null_weight(descs, ws):
        ws[1] = None
    assert call(null_weight) == -1

    def stats_flag(descs, ws):
        for i in range(n):
            descs[i].flags = _lib.F_STATS
    assert call(stats_flag) == -1


@pytest.mark.parametrize("case", [(2, 224, 224), (3, 64, 64), (2, 34, 50), (1, 130, 66), (5, 32, 96), (1, 320, 304)])
def test_stem_wgrad_direct(case):
    """LOANS_TILE_STEM of loans_wgrad_f32 (csrc/stem.hip, stem7_wgrad_kernel): conv1's weight gradient as a persistent direct
    kernel, against the oracle and the implicit-GEMM kernel; even and odd output widths, more units than blocks and fewer,
    accumulation into a non-zero dw, the window-padding columns of the dense layout untouched"""
    from loans_amd import ops
    B, H, W = case
    rng = np.random.RandomState(H + 3 * W)
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, 3, 64, 7, 2, 3, dense=True)
    assert ops.stem_wgrad_ok(geo)
    gy = rng.standard_normal((B, 64, geo.Ho, geo.Wo)).astype(np.float32)
    xp = np.zeros((B, geo.Hp, geo.Wp, 3), np.float32)
    xp[:, 3:3 + H, 3:3 + W] = x.transpose(0, 2, 3, 1)
    w0 = np.zeros((64, 3, 7, 7))
    _, col = C.conv2d_fwd(x.astype(np.float64), w0, None, 2, 3)
    _, gw_ref, _ = C.conv2d_bwd(x.shape, col, w0, gy.astype(np.float64), 2, 3, False)
    xd, gyd = dev(xp), dev(_nhwc(gy))
    dw = torch.zeros((64, 7, geo.kwp, 3), device='cuda')
    ops._conv_wgrad(xd, gyd, dw, geo, False, 0, ops.TILE_STEM)
    got = dw.cpu().numpy()
    assert not got[:, :, 7:].any()
    assert rel_err(got[:, :, :7].transpose(0, 3, 1, 2), gw_ref) < 5e-6
    dw3 = torch.zeros_like(dw)
    ops._conv_wgrad(xd, gyd, dw3, geo, False, 0, 3)
    assert torch.allclose(dw, dw3, rtol=1e-4, atol=1e-3 * float(np.abs(gw_ref).max()))
    ops._conv_wgrad(xd, gyd, dw, geo, False, 0, ops.TILE_STEM)           # accumulates
    assert rel_err(dw.cpu().numpy()[:, :, :7].transpose(0, 3, 1, 2), 2 * gw_ref) < 5e-6


def test_stem_weight_gradient_leaves_the_calling_stream():
    """conv_wgrad sends a weight gradient to the weight-gradient stream once its shape is tuned.  The stem's table entry is kept under
    a key of its own (the direct kernel is among its candidates); rounds 2-5 looked it up under the plain key, found nothing and left
    conv1's weight gradient on the calling stream for good: +0.16 ms per fp32 step (profiles/r5_issue_orders_ab.txt)."""
    from loans_amd import ops
    geo = ops.ConvGeometry(2, 64, 64, 3, 64, 7, 2, 3, dense=True)
    x = ops.prep_images(torch.rand((2, 3, 64, 64), device='cuda'), geo)
    gy = torch.randn((2, geo.Ho, geo.Wo, 64), device='cuda')
    ref, dw = torch.zeros((64, 7, geo.kwp, 3), device='cuda'), torch.zeros((64, 7, geo.kwp, 3), device='cuda')
    assert ops.ASYNC_WGRAD and ops.stem_wgrad_ok(geo)
    ops.conv_wgrad(x, gy, ref, geo)                 # the first call times the candidates (on the calling stream)
    ops.join_side_stream()
    assert x.device.index not in ops._side_dirty
    ops.conv_wgrad(x, gy, dw, geo)
    assert x.device.index in ops._side_dirty        # ... the second one was handed to the weight-gradient stream
    ops.join_side_stream()
    torch.cuda.synchronize()
    assert torch.allclose(dw, ref, rtol=1e-4, atol=1e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("n", [4, 8, 12, 1020, (1 << 20) + 4, 5_000_004])
def test_cast_bf16_is_round_to_nearest_even_at_every_length(n):
    """loans_cast_bf16 (the arena's bf16 shadow) is exactly torch's RNE cast, ties included.  (A form with eight elements per thread
    and 16-byte stores measured the same 3.4-4.1 TB/s and was not kept.)"""
    from loans_amd import ops
    x = torch.randn(n, device='cuda') * 3
    x[:4] = torch.tensor([1.00390625, 1.01171875, -1.00390625, 3.0e38], device='cuda')     # two ties (to even: down / up), a large value
    out = ops.cast_bf16(x)
    assert out.dtype == torch.bfloat16 and torch.equal(out, x.to(torch.bfloat16))


def test_prep_images_exact():
    from loans_amd import ops
    rng = np.random.RandomState(0)
    k = rng.randint(0, 256, size=(3, 3, 20, 24)).astype(np.float32)
    x = (k / np.float32(255)).astype(np.float32)
    x[0, 0, 0, :8] = np.clip((k[0, 0, 0, :8] + 0.5) / 255, 0, 1)        # truncation, not rounding
    ref = C.prepare_images(x)
    out = ops.prep_images(dev(x)).cpu().numpy()
    np.testing.assert_array_equal(out[..., :3].transpose(0, 3, 1, 2), ref)
    assert not out[..., 3].any()


def test_conv_split_tile():
    """LOANS_TILE_SPLIT: 128x128 tiles over the rows that fill whole rounds of the machine, 64x64 over the rest --
    needs more than 2 x CUs big tiles to take the two-launch path, so a wide, shallow problem"""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p = 24, 8, 56, 56, 32, 1, 1, 0
    rng = np.random.RandomState(11)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin)).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    assert B * H * W > 2 * torch.cuda.get_device_properties(0).multi_processor_count * 128
    xd, wd = dev(_nhwc(x)), dev(_ohwi(w))
    y_ref = np.einsum('bchw,oc->bohw', x.astype(np.float64), w[:, :, 0, 0].astype(np.float64))
    add = rng.standard_normal(y_ref.shape).astype(np.float32)
    stats_r = ops.stats_buffer(Cout, 'cuda')
    y = ops.conv_fprop(xd, wd, geo, stats=stats_r, addend=dev(_nhwc(add)), tile=6)
    assert rel_err(_nchw(y), y_ref + add) < 2e-6
    stats = stats_r.sum(dim=0)
    np.testing.assert_allclose(stats[0].cpu().numpy(), y_ref.sum(axis=(0, 2, 3)), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(stats[1].cpu().numpy(), (y_ref ** 2).sum(axis=(0, 2, 3)), rtol=1e-5)
    y1 = ops.conv_fprop(xd, wd, geo, addend=dev(_nhwc(add)), tile=3)
    assert torch.equal(y, y1)                       # K is accumulated in the same order by every tile shape
    for t in (17, 19, 22):                          # ... and by the LDS-DMA staged variants
        assert torch.equal(y, ops.conv_fprop(xd, wd, geo, addend=dev(_nhwc(add)), tile=t)), t
    gy = rng.standard_normal(y_ref.shape).astype(np.float32)
    gx_ref = np.einsum('bohw,oc->bchw', gy.astype(np.float64), w[:, :, 0, 0].astype(np.float64))
    ref_t = rng.standard_normal(x.shape).astype(np.float32)
    gx = ops.conv_dgrad(dev(_nhwc(gy)), wd, geo, mask_ref=dev(_nhwc(ref_t)), tile=6)
    assert rel_err(_nchw(gx), gx_ref * (ref_t > 0)) < 2e-6


@pytest.mark.parametrize("case", [(40, 32, 29, 29, 128, 3, 1, 1, 0), (37, 256, 31, 31, 128, 1, 1, 0, 1), (66, 64, 56, 56, 128, 3, 2, 1, 0)])
def test_conv_fine_tail_tile(case):
    """LOANS_TILE_FINETAIL: the 64x64 tiles that share out evenly over the CUs at full K, the rest as K-slices behind them in
    the same launch (atomics into rows zeroed by the launcher) + loans_igemm_finalize_f32 over those rows; against the
    oracle convolution and the plain 64x64 tile, with bias / BN statistics / relu(in)"""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p, relu_in = case
    rng = np.random.RandomState(13)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    M_, nch = B * geo.Ho * geo.Wo, (k * k * Cin + 31) // 32
    rows_head, slices = ops._finetail_plan(M_, Cout, nch, torch.device('cuda', 0))
    assert 0 < rows_head < M_ and slices >= 2, 'the case must take the sliced path on this machine'
    xin = np.maximum(x, 0) if relu_in else x
    y_ref, _ = C.conv2d_fwd(xin.astype(np.float64), w.astype(np.float64), bias.astype(np.float64), s, p)
    xd, wd, bd = dev(_nhwc(x)), dev(_ohwi(w)), dev(bias)
    y3 = ops.conv_fprop(xd, wd, geo, bias=bd, relu_in=bool(relu_in), tile=3)
    nimg = rows_head // (geo.Ho * geo.Wo)
    for tile in (ops.TILE_FINETAIL, ops.TILE_FINETAIL | 16):
        st = ops.stats_buffer(Cout, 'cuda')
        y = ops.conv_fprop(xd, wd, geo, bias=bd, stats=st, relu_in=bool(relu_in), tile=tile)
        assert rel_err(_nchw(y), y_ref) < 2e-6, tile
        assert torch.equal(y[:nimg], y3[:nimg])               # full-K tiles: the plain kernel's arithmetic
        assert torch.allclose(y, y3, rtol=1e-5, atol=1e-5)
        stats = st.sum(dim=0)
        np.testing.assert_allclose(stats[0].cpu().numpy(), y_ref.sum(axis=(0, 2, 3)), rtol=1e-5, atol=1e-2)
        np.testing.assert_allclose(stats[1].cpu().numpy(), (y_ref ** 2).sum(axis=(0, 2, 3)), rtol=1e-5)
    assert ops._igemm_launches(M_, Cout, ops.TILE_FINETAIL, torch.device('cuda', 0), nch) == 2


@pytest.mark.parametrize("case", [(2, 64, 14, 14, 64, 3, 1, 1), (2, 64, 15, 13, 128, 3, 2, 1), (3, 256, 7, 7, 64, 1, 1, 0),
                                  (1, 128, 9, 9, 128, 4, 2, 1)])
@pytest.mark.parametrize("splits", [2, 4, 16])
def test_conv_split_k(case, splits):
    """LOANS_TILE_SPLITK: blocks contract slices of K and add raw partial tiles; loans_igemm_finalize_f32 applies bias /
    statistics / mask / addend to the finished sums.  Same results as the one-pass kernel up to the fp32 summation order
    (1e-6), for the forward conv (+ bias + stats, relu_in + aliased addend) and the data gradient (mask + addend, masked
    addend, in-place accumulation)."""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    rng = np.random.RandomState(17)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    xd, wd, bd = dev(_nhwc(x)), dev(_ohwi(w)), dev(b)
    tile = 3 | (splits << 8)
    st1, stk = ops.stats_buffer(Cout, 'cuda'), ops.stats_buffer(Cout, 'cuda')
    y1 = ops.conv_fprop(xd, wd, geo, bias=bd, stats=st1, tile=3)
    yk = ops.conv_fprop(xd, wd, geo, bias=bd, stats=stk, tile=tile)
    assert rel_err(yk.cpu().numpy(), y1.cpu().numpy()) < 2e-6
    np.testing.assert_allclose(stk.sum(0).cpu().numpy(), st1.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    add = dev(_nhwc(rng.standard_normal((B, Cout, geo.Ho, geo.Wo)).astype(np.float32)))
    a1, ak = add.clone(), add.clone()
    ops.conv_fprop(xd, wd, geo, out=a1, relu_in=True, addend=a1, tile=3)          # in place: out = conv(relu(x)) + out
    ops.conv_fprop(xd, wd, geo, out=ak, relu_in=True, addend=ak, tile=tile)
    assert rel_err(ak.cpu().numpy(), a1.cpu().numpy()) < 2e-6
    gy = dev(_nhwc(rng.standard_normal((B, Cout, geo.Ho, geo.Wo)).astype(np.float32)))
    ref_t = dev(_nhwc(rng.standard_normal(x.shape).astype(np.float32)))
    addx = dev(_nhwc(rng.standard_normal(x.shape).astype(np.float32)))
    g1 = ops.conv_dgrad(gy, wd, geo, mask_ref=ref_t, addend=addx, tile=3)
    gk = ops.conv_dgrad(gy, wd, geo, mask_ref=ref_t, addend=addx, tile=tile)
    assert rel_err(gk.cpu().numpy(), g1.cpu().numpy()) < 2e-6
    g1 = ops.conv_dgrad(gy, wd, geo, addend=addx, addend_mask_ref=ref_t, tile=3)
    gk = ops.conv_dgrad(gy, wd, geo, addend=addx, addend_mask_ref=ref_t, tile=tile)
    assert rel_err(gk.cpu().numpy(), g1.cpu().numpy()) < 2e-6
    a1, ak = addx.clone(), addx.clone()
    ops.conv_dgrad(gy, wd, geo, out=a1, addend=a1, tile=3)
    ops.conv_dgrad(gy, wd, geo, out=ak, addend=ak, tile=tile)
    assert rel_err(ak.cpu().numpy(), a1.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("case", [(3, 64, 14, 14, 128, 128, 3, 2, 1), (2, 256, 9, 9, 64, 256, 1, 2, 0), (2, 64, 12, 12, 64, 200, 3, 1, 1)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 17, 18, 19])
def test_conv_fprop_pair(case, tile):
    """loans_igemm_pair_f32: two convolutions of one input (BasicA's conv1 + conv shortcut; a bottleneck's conv1 + conv4,
    different Cout) in one launch give bit-identical outputs and the same statistics as two launches"""
    from loans_amd import ops
    B, Cin, H, W, Ca, Cb, k, s, p = case
    rng = np.random.RandomState(23)
    x = dev(_nhwc(rng.standard_normal((B, Cin, H, W)).astype(np.float32)))
    wa = dev(_ohwi((rng.standard_normal((Ca, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)))
    wb = dev(_ohwi((rng.standard_normal((Cb, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)))
    ga, gb = ops.ConvGeometry(B, H, W, Cin, Ca, k, s, p), ops.ConvGeometry(B, H, W, Cin, Cb, k, s, p)
    assert ops.fprop_pair_ok(x, ga, gb)
    sa, sb, sa1, sb1 = (ops.stats_buffer(c, 'cuda') for c in (Ca, Cb, Ca, Cb))
    ya, yb = ops.conv_fprop_pair(x, wa, wb, ga, gb, sa, sb, tile=tile)
    ya1 = ops.conv_fprop(x, wa, ga, stats=sa1, tile=tile if tile else 3)
    yb1 = ops.conv_fprop(x, wb, gb, stats=sb1, tile=tile if tile else 3)
    assert torch.equal(ya, ya1) and torch.equal(yb, yb1)
    np.testing.assert_allclose(sa.sum(0).cpu().numpy(), sa1.sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(sb.sum(0).cpu().numpy(), sb1.sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)


def test_prep_images_dense_exact():
    """the padded packed-RGB buffer conv1 reads: same arithmetic, zero border, every element written"""
    from loans_amd import ops
    rng = np.random.RandomState(1)
    for (B, H, W) in [(3, 20, 24), (2, 9, 11)]:
        k = rng.randint(0, 256, size=(B, 3, H, W)).astype(np.float32)
        x = (k / np.float32(255)).astype(np.float32)
        x[0, 0, 0, :8] = np.clip((k[0, 0, 0, :8] + 0.5) / 255, 0, 1)
        ref = C.prepare_images(x)
        geo = ops.ConvGeometry(B, H, W, 3, 64, 7, 2, 3, dense=True)
        out = ops.prep_images(dev(x), geo)
        assert out.frame_hw == (H, W) and out.shape == (B, geo.Hp, geo.Wp, 3) and geo.Wp % 2 == 0
        out = out.cpu().numpy()
        np.testing.assert_array_equal(out[:, 3:3 + H, 3:3 + W].transpose(0, 3, 1, 2), ref)
        border = out.copy()
        border[:, 3:3 + H, 3:3 + W] = 0
        assert not border.any()


@pytest.mark.parametrize("case", [(2, 32, 32, 64, 7, 2, 3), (3, 17, 23, 64, 7, 2, 3), (2, 12, 12, 128, 3, 1, 1),
                                  (1, 40, 36, 64, 7, 2, 3)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 17, 18, 19, 20])
def test_conv_dense_rows(case, tile):
    """LOANS_F_DENSE: the RGB stem on packed 3-channel rows of a zero-padded frame (K = kh x 24 instead of kh*kw x 4)"""
    from loans_amd import ops
    B, H, W, Cout, k, s, p = case
    rng = np.random.RandomState(sum(case))
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, 3, k, k)) / np.sqrt(3 * k * k)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, 3, Cout, k, s, p, dense=True)
    xp = np.zeros((B, geo.Hp, geo.Wp, 3), np.float32)
    xp[:, p:p + H, p:p + W] = x.transpose(0, 2, 3, 1)
    wp = np.zeros((Cout, k, geo.kwp, 3), np.float32)
    wp[:, :, :k] = w.transpose(0, 2, 3, 1)
    xd, wd, bd = dev(xp), dev(wp), dev(b)
    y_ref, col = C.conv2d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s, p)
    stats_r = ops.stats_buffer(Cout, 'cuda')
    itile = tile if tile != 5 else 0
    y = ops.conv_fprop(xd, wd, geo, bias=bd, stats=stats_r, tile=itile)
    stats = stats_r.sum(dim=0)
    assert rel_err(_nchw(y), y_ref) < 2e-6
    np.testing.assert_allclose(stats[0].cpu().numpy(), y_ref.sum(axis=(0, 2, 3)), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(stats[1].cpu().numpy(), (y_ref ** 2).sum(axis=(0, 2, 3)), rtol=1e-5)
    if tile in (0, 1, 3, 5):
        gy = rng.standard_normal(y_ref.shape).astype(np.float32)
        _, gw_ref, _ = C.conv2d_bwd(x.shape, col, w.astype(np.float64), gy.astype(np.float64), s, p, False)
        dw = torch.zeros_like(wd)
        ops.conv_wgrad(xd, dev(_nhwc(gy)), dw, geo, tile=tile)
        ops.conv_wgrad(xd, dev(_nhwc(gy)), dw, geo, splits=3, tile=tile)          # accumulates
        ops.join_side_stream()
        got = dw.cpu().numpy()
        assert rel_err(got[:, :, :k].transpose(0, 3, 1, 2), 2 * gw_ref) < 5e-6
        assert not got[:, :, k:].any()          # window-padding columns carry no gradient


@pytest.mark.parametrize("case", [(2, 224, 224), (3, 64, 64), (2, 32, 32), (1, 128, 128), (2, 64, 96), (1, 256, 64)])
def test_conv_stem_direct(case):
    """LOANS_TILE_STEM (csrc/stem.hip): conv1 7x7 / 2, 3 -> 64 with bias and BN statistics as a direct convolution from an
    LDS-staged image, against the oracle convolution and the implicit-GEMM dense-row kernel; frame sizes with 1 ... 7
    pixel tiles per wave"""
    from loans_amd import ops
    B, H, W = case
    rng = np.random.RandomState(H + W)
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    w = (rng.standard_normal((64, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, 3, 64, 7, 2, 3, dense=True)
    assert ops.stem_tile_rows(geo) > 0
    xp = np.zeros((B, geo.Hp, geo.Wp, 3), np.float32)
    xp[:, 3:3 + H, 3:3 + W] = x.transpose(0, 2, 3, 1)
    wp = np.zeros((64, 7, geo.kwp, 3), np.float32)
    wp[:, :, :7] = w.transpose(0, 2, 3, 1)
    xd, wd, bd = dev(xp), dev(wp), dev(b)
    y_ref, _ = C.conv2d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 2, 3)
    st, st3 = ops.stats_buffer(64, 'cuda'), ops.stats_buffer(64, 'cuda')
    y = ops.conv_fprop(xd, wd, geo, bias=bd, stats=st, tile=ops.TILE_STEM)
    y3 = ops.conv_fprop(xd, wd, geo, bias=bd, stats=st3, tile=3)
    assert rel_err(_nchw(y), y_ref) < 2e-6
    assert torch.allclose(y, y3, rtol=1e-5, atol=1e-5)
    stats = st.sum(dim=0)
    np.testing.assert_allclose(stats[0].cpu().numpy(), y_ref.sum(axis=(0, 2, 3)), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(stats[1].cpu().numpy(), (y_ref ** 2).sum(axis=(0, 2, 3)), rtol=1e-5)
    y_nb = ops.conv_fprop(xd, wd, geo, tile=ops.TILE_STEM)                    # no bias, no statistics
    assert torch.allclose(y_nb, y - bd, rtol=1e-5, atol=1e-5)
    # sizes the kernel does not cover are refused (the autotuner never offers it there)
    odd = ops.ConvGeometry(1, 17, 23, 3, 64, 7, 2, 3, dense=True)
    assert ops.stem_tile_rows(odd) == 0
    with pytest.raises(RuntimeError):
        ops.conv_fprop(torch.zeros((1, odd.Hp, odd.Wp, 3), device='cuda'), wd, odd, tile=ops.TILE_STEM)


@pytest.mark.parametrize("C_", [64, 128, 512, 2048])
def test_bn_forward_backward(C_):
    from loans_amd import ops
    rng = np.random.RandomState(1)
    B, H, W = 3, 9, 7
    x = (rng.standard_normal((B, C_, H, W)) * 2 + 0.5).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(C_)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C_)).astype(np.float32)
    rm, rv = np.zeros(C_), np.ones(C_)
    y_ref, ctx = C.bn_fwd_train(x.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64), rm, rv)
    xd = dev(_nhwc(x))
    stats = ops.stats_buffer(C_, 'cuda')
    stats[3, 0] = xd.double().sum(dim=(0, 1, 2)); stats[7, 1] = (xd.double() ** 2).sum(dim=(0, 1, 2))
    rmd, rvd = torch.zeros(C_, device='cuda'), torch.ones(C_, device='cuda')
    st = ops.bn_finalize(stats, B * H * W, dev(gamma), dev(beta), rmd, rvd)
    np.testing.assert_allclose(rmd.cpu().numpy(), rm, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv, rtol=1e-5)
    y = ops.bn_apply(xd, st, relu=False)
    assert rel_err(_nchw(y), y_ref) < 1e-5
    res = rng.standard_normal(x.shape).astype(np.float32)
    y = ops.bn_apply(xd, st, relu=True, residual=dev(_nhwc(res)))
    assert rel_err(_nchw(y), np.maximum(y_ref + res, 0)) < 1e-5
    y = ops.bn_apply(xd, st, relu=True, x2=xd, st2=st)
    assert rel_err(_nchw(y), np.maximum(2 * y_ref, 0)) < 1e-5

    gy = rng.standard_normal(x.shape).astype(np.float32)
    mask = rng.standard_normal(x.shape).astype(np.float32)
    g_eff = gy * (mask > 0)
    gx_ref, gg_ref, gb_ref = C.bn_bwd(ctx, gamma.astype(np.float64), g_eff.astype(np.float64))
    gg, gb = torch.zeros(C_, device='cuda'), torch.zeros(C_, device='cuda')
    gx = ops.bn_backward(dev(_nhwc(gy)), dev(_nhwc(mask)), xd, st, dev(gamma), gg, gb)
    assert rel_err(_nchw(gx), gx_ref) < 2e-5
    assert rel_err(gg.cpu().numpy(), gg_ref) < 1e-5 and rel_err(gb.cpu().numpy(), gb_ref) < 1e-5
    # the BN's own ReLU as the mask: sign recomputed from x (loans_bn_bwd_*_xmask) == mask tensor read from memory
    for tdt in (torch.float32, torch.bfloat16):
        xs, gys = xd.to(tdt), dev(_nhwc(gy)).to(tdt)
        h = ops.bn_apply(xs, st, relu=True)
        outs = []
        for own in (False, True):
            gga, gba = torch.zeros(C_, device='cuda'), torch.zeros(C_, device='cuda')
            outs.append((ops.bn_backward(gys, h, xs, st, dev(gamma), gga, gba, mask_is_own_relu=own), gga, gba))
        assert rel_err(outs[1][0].float().cpu().numpy(), outs[0][0].float().cpu().numpy()) < (1e-5 if tdt == torch.float32 else 2 ** -7)
        assert rel_err(outs[1][1].cpu().numpy(), outs[0][1].cpu().numpy()) < 1e-6
        assert rel_err(outs[1][2].cpu().numpy(), outs[0][2].cpu().numpy()) < 1e-6
    # the unit's output as the mask, handed over as sign bits (ops.bn_apply(want_bits=True) -> loans_bn_bwd_*_bits_*):
    # same results as reading the mask tensor, single and dual form
    for tdt in (torch.float32, torch.bfloat16):
        xs, gys, rs = xd.to(tdt), dev(_nhwc(gy)).to(tdt), dev(_nhwc(res)).to(tdt)
        y_plain = ops.bn_apply(xs, st, relu=True, residual=rs)
        y_bits = ops.bn_apply(xs, st, relu=True, residual=rs, want_bits=True)
        assert torch.equal(y_plain, y_bits) and not hasattr(y_plain, 'relu_bits')
        assert y_bits.relu_bits.dtype == torch.uint8 and y_bits.relu_bits.numel() == xs.numel() // 4
        ref_bits = ((y_plain.float().reshape(-1, 4) > 0).to(torch.int32) * torch.tensor([1, 2, 4, 8], dtype=torch.int32, device='cuda')).sum(1).to(torch.uint8)
        assert torch.equal(y_bits.relu_bits, ref_bits)
        for dual in (False, True):
            outs = []
            for m in (y_plain, y_bits):
                g1, b1, g2, b2 = (torch.zeros(C_, device='cuda') for _ in range(4))
                kw = dict(x2=xs, st2=st, gamma2=dev(gamma), ggamma2=g2, gbeta2=b2) if dual else {}
                r = ops.bn_backward(gys, m, xs, st, dev(gamma), g1, b1, **kw)
                outs.append((r if dual else (r,)) + (g1, b1, g2, b2))
            # (the fp64 atomics of the reduction land in a different order from run to run: equal up to the last bit of
            # the fp32 coefficients, one bf16 ulp on rounded outputs)
            tol = 1e-5 if tdt == torch.float32 else 2 ** -7
            for a, b in zip(*outs):
                assert rel_err(a.float().cpu().numpy(), b.float().cpu().numpy()) < tol
    # dual form (BasicA's output feeds bn2 and bn3)
    gg2, gb2 = torch.zeros(C_, device='cuda'), torch.zeros(C_, device='cuda')
    gg.zero_(); gb.zero_()
    gxa, gxb = ops.bn_backward(dev(_nhwc(gy)), dev(_nhwc(mask)), xd, st, dev(gamma), gg, gb,
                               x2=xd, st2=st, gamma2=dev(gamma), ggamma2=gg2, gbeta2=gb2)
    assert rel_err(_nchw(gxa), gx_ref) < 2e-5 and rel_err(_nchw(gxb), gx_ref) < 2e-5
    assert rel_err(gg2.cpu().numpy(), gg_ref) < 1e-5


def test_stem_pool_forward_backward():
    from loans_amd import ops
    rng = np.random.RandomState(2)
    for (B, H, W) in [(2, 16, 16), (1, 9, 12), (2, 8, 7)]:
        C_ = 64
        x = rng.standard_normal((B, C_, H, W)).astype(np.float32)
        scale = (1 + 0.1 * rng.standard_normal(C_)).astype(np.float32)
        shift = (0.1 * rng.standard_normal(C_)).astype(np.float32)
        pre = np.maximum(x * scale[None, :, None, None] + shift[None, :, None, None], 0)
        y_ref, idx_ref = C.max_pool_fwd(pre)
        st = ops.BNState(C_, 'cuda')
        st.scale.copy_(dev(scale)); st.shift.copy_(dev(shift))
        xd = dev(_nhwc(x))
        y, idx = ops.bn_relu_maxpool(xd, st)
        np.testing.assert_allclose(_nchw(y), y_ref, rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(idx.cpu().numpy().transpose(0, 3, 1, 2), idx_ref)
        gy = rng.standard_normal(y_ref.shape).astype(np.float32)
        g_ref = C.max_pool_bwd(pre.shape, idx_ref, gy) * (pre > 0)
        g = ops.maxpool_relu_bwd(dev(_nhwc(gy)), idx, xd, st)
        np.testing.assert_allclose(_nchw(g), g_ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (1, 9, 12, 64), (2, 8, 7, 128), (3, 13, 11, 64), (2, 3, 3, 64)])
def test_stem_tail_fused_backward(shape, dtype):
    """loans_pool_bn_bwd_reduce / _apply (max-pool backward + ReLU mask + BN backward without the dense gradient in
    between) against the oracle's max_pool_bwd -> relu mask -> bn_bwd, and against the three-pass kernels"""
    from loans_amd import ops
    B, H, W, C_ = shape
    rng = np.random.RandomState(5)
    x = (rng.standard_normal((B, C_, H, W)) * 2 + 0.5).astype(np.float32)
    if dtype == "bf16":
        x = torch.from_numpy(x).bfloat16().float().numpy()
    gamma = (1 + 0.1 * rng.standard_normal(C_)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C_)).astype(np.float32)
    y_bn, ctx = C.bn_fwd_train(x.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64),
                               np.zeros(C_), np.ones(C_))
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    xd = dev(_nhwc(x)).to(tdt)
    stats = ops.stats_buffer(C_, 'cuda')
    stats[0, 0] = xd.double().sum(dim=(0, 1, 2)); stats[0, 1] = (xd.double() ** 2).sum(dim=(0, 1, 2))
    st = ops.bn_finalize(stats, B * H * W, dev(gamma), dev(beta), torch.zeros(C_, device='cuda'), torch.ones(C_, device='cuda'))
    y, idx = ops.bn_relu_maxpool(xd, st)
    # (round 5) the pool's third output: the RAW x at every pooled element's argmax -- exactly those elements, and y / idx unchanged
    y2, idx2, xsel = ops.bn_relu_maxpool(xd, st, want_sel=True)
    assert torch.equal(y2, y) and torch.equal(idx2, idx) and xsel is not None and xsel.dtype == xd.dtype
    OH, OW = y.shape[1], y.shape[2]
    ii = idx.long()
    oh = torch.arange(OH, device='cuda').view(1, OH, 1, 1)
    ow = torch.arange(OW, device='cuda').view(1, 1, OW, 1)
    src = ((oh * 2 + ii // 3) * W + (ow * 2 + ii % 3))                          # argmax pixel inside its image
    want = torch.gather(xd.reshape(B, H * W, C_), 1, src.reshape(B, OH * OW, C_)).reshape(B, OH, OW, C_)
    assert torch.equal(xsel, want)
    gy = rng.standard_normal(tuple(y.shape)).astype(np.float32)           # NHWC
    if dtype == "bf16":
        gy = torch.from_numpy(gy).bfloat16().float().numpy()
    gyd = dev(gy).to(tdt)
    # oracle on the GPU's own argmax (ties / the last bits of the pre-activation may pick another equal maximum)
    pre = np.maximum(y_bn, 0)
    idx_np = idx.cpu().numpy().transpose(0, 3, 1, 2)
    g_ref = C.max_pool_bwd(pre.shape, idx_np, gy.transpose(0, 3, 1, 2).astype(np.float64)) * (y_bn > 0)
    gx_ref, gg_ref, gb_ref = C.bn_bwd(ctx, gamma.astype(np.float64), g_ref)
    gg, gb = torch.zeros(C_, device='cuda'), torch.zeros(C_, device='cuda')
    gbias = torch.zeros(C_, device='cuda')
    gx = ops.pool_bn_backward(gyd, idx, xd, st, dev(gamma), gg, gb, gbias=gbias)
    tol = 1e-2 if dtype == "bf16" else 2e-5
    assert rel_err(_nchw(gx.float()), gx_ref) < tol
    # the bias gradient of the conv in front of a train-mode BN is analytically zero: compare on the scale of sum |gx|
    assert np.abs(gbias.cpu().numpy() - gx_ref.sum(axis=(0, 2, 3))).max() < 1e-5 * np.abs(gx_ref).sum(axis=(0, 2, 3)).max() + 1e-6
    assert rel_err(gg.cpu().numpy(), gg_ref) < 1e-4 and rel_err(gb.cpu().numpy(), gb_ref) < 1e-4
    # the sums taken from (gy, xsel) instead of gathered out of x: the same terms in another order
    ggs, gbs, gbias_s = torch.zeros(C_, device='cuda'), torch.zeros(C_, device='cuda'), torch.zeros(C_, device='cuda')
    gxs = ops.pool_bn_backward(gyd, idx, xd, st, dev(gamma), ggs, gbs, gbias=gbias_s, xsel=xsel)
    assert rel_err(_nchw(gxs.float()), gx_ref) < tol
    assert rel_err(ggs.cpu().numpy(), gg_ref) < 1e-4 and rel_err(gbs.cpu().numpy(), gb_ref) < 1e-4
    assert rel_err(ggs.cpu().numpy(), gg.cpu().numpy()) < 1e-5 and rel_err(gbs.cpu().numpy(), gb.cpu().numpy()) < 1e-5
    if dtype == "f32":
        assert rel_err(gxs.cpu().numpy(), gx.cpu().numpy()) < 1e-5
    # the three-pass form
    gg3, gb3 = torch.zeros(C_, device='cuda'), torch.zeros(C_, device='cuda')
    gx3 = ops.bn_backward(ops.maxpool_relu_bwd(gyd, idx, xd, st), None, xd, st, dev(gamma), gg3, gb3)
    if dtype == "f32":
        assert rel_err(gx.cpu().numpy(), gx3.cpu().numpy()) < 1e-6
        assert rel_err(gg.cpu().numpy(), gg3.cpu().numpy()) < 1e-6 and rel_err(gb.cpu().numpy(), gb3.cpu().numpy()) < 1e-6
    else:
        assert rel_err(gx.float().cpu().numpy(), gx3.float().cpu().numpy()) < 1e-2


def test_spatial_transformer_forward_backward():
    from loans_amd import ops
    rng = np.random.RandomState(3)
    B, H, W, th, tw = 4, 20, 28, 9, 11
    img = rng.rand(B, 3, H, W).astype(np.float32)
    theta = np.tile(np.array([[.8, 0, 0], [0, .8, 0]], np.float32), (B, 1, 1))
    theta += (0.3 * rng.standard_normal(theta.shape)).astype(np.float32)
    theta[1] = [[1.6, 0, 0.2], [0, 1.7, -0.1]]       # samples well outside the frame
    grid_ref, coords = C.st_grid_fwd(theta, (th, tw))
    grid = ops.st_grid_fwd(dev(theta), (th, tw))
    np.testing.assert_allclose(grid.cpu().numpy(), grid_ref, atol=2e-7)
    rois_ref = C.st_sampler_fwd(img, grid_ref)
    rois = ops.st_sampler_fwd(dev(img), dev(grid_ref))
    np.testing.assert_allclose(rois.cpu().numpy()[..., :3].transpose(0, 3, 1, 2), rois_ref, atol=1e-6)
    gy = rng.standard_normal(rois_ref.shape).astype(np.float32)
    gg_ref = C.st_sampler_bwd_grid(img, grid_ref, gy)
    gg = ops.st_sampler_bwd_grid(dev(img), dev(grid_ref), dev(_nhwc(gy, 4)))
    np.testing.assert_allclose(gg.cpu().numpy(), gg_ref, rtol=1e-4, atol=1e-5)
    gt_ref = C.st_grid_bwd(coords, gg_ref)
    gt = ops.st_grid_bwd(dev(gg_ref))
    np.testing.assert_allclose(gt.cpu().numpy(), gt_ref, rtol=1e-4, atol=1e-5)


def test_regulariser_gradients_at_exact_ties():
    """SURVEY 8c-style known answer for common/utils.py:165-178,303-316 at an exact tie (tests/test_oracle_kat.py::test_kat9 is
    the oracle's twin): Chainer's Maximum hands the gradient to its first argument where x1 >= x2, Absolute's backward is
    sign(x) gy.  theta = identity: TR_x - 1 == 0 and BL_y - 1 == 0 -> gradient 1 there, TL_x + 1 == 0 and TL_y + 1 == 0 ->
    nothing.  Expected values are written out, not taken from the oracle."""
    from loans_amd import ops
    th, tw, B = 5, 6, 3
    one = torch.ones((), device='cuda')
    theta = torch.tensor([[1., 0, 0], [0, 1., 0]], device='cuda').expand(B, 2, 3).contiguous()
    grid = ops.st_grid_fwd(theta, (th, tw))
    g = grid.cpu().numpy()
    assert (g[:, 0, 0, tw - 1] == 1.0).all() and (g[:, 1, th - 1, 0] == 1.0).all()      # the kernel's linspace ends exactly on 1
    assert (g[:, 0, 0, 0] == -1.0).all() and (g[:, 1, 0, 0] == -1.0).all()
    assert float(ops.grid_loss_fwd(grid, 1)) == 0.0
    want = np.zeros_like(g)
    want[:, 0, 0, tw - 1] = 1.0
    want[:, 1, th - 1, 0] = 1.0
    np.testing.assert_array_equal(ops.grid_loss_bwd(grid, one, 1).cpu().numpy(), want)
    np.testing.assert_array_equal(C.out_of_image_loss(g)[1], want)
    # data-parallel scale of the batch SUM (oob_scale = world size) multiplies the tie's gradient like any other
    np.testing.assert_array_equal(ops.grid_loss_bwd(grid, one, 1, oob_scale=4.0).cpu().numpy(), 4.0 * want)
    # zero-height box: TL_y - BL_y == 0 exactly
    theta = torch.tensor([[1., 0, 0], [0, 0, 0.25]], device='cuda').expand(B, 2, 3).contiguous()
    grid = ops.st_grid_fwd(theta, (th, tw))
    assert float(ops.grid_loss_fwd(grid, 0, 48, 64)) == 0.0
    want = np.zeros_like(g)
    want[:, 1, 0, 0] = 48 / 2.0 / B
    want[:, 1, th - 1, 0] = -48 / 2.0 / B
    np.testing.assert_array_equal(ops.grid_loss_bwd(grid, one, 0, 48, 64).cpu().numpy(), want)
    np.testing.assert_array_equal(C.direction_loss(grid.cpu().numpy(), (48, 64))[1], want)


def test_tie_gradient_reaches_param_predictor_through_the_model():
    """The same tie through the product's call surface: a localizer whose param_predictor says theta = identity exactly
    (W = 0, b = [1,0,0,0,1,0]); OutOfImageLossCalculator.calc_loss(points).backward() leaves d loss / d b = sum over the batch of
    [TR_x's xs = 1, (masked), 1, (masked), BL_y's ys = 1, 1] = [B, 0, B, 0, B, B]."""
    import loans_amd
    B, crop = 3, (16, 16)
    np.random.seed(0)
    loc = loans_amd.SheepLocalizer(crop)
    loc.param_predictor.b.set_logical(np.array([1, 0, 0, 0, 1, 0], np.float32))
    from loans_amd.datasets import synthetic
    rois, points = loc(dev(synthetic.make_frames(1, B, 64, 64)))
    assert torch.equal(loc.last_transform_params.data, torch.tensor([[1., 0, 0], [0, 1., 0]], device='cuda').expand(B, 2, 3))
    loss = loans_amd.OutOfImageLossCalculator(torch).calc_loss(points, loans_amd.Size(64, 64))
    assert float(loss.data) == 0.0
    loc.cleargrads()
    loss.backward()
    np.testing.assert_array_equal(loc.param_predictor.b.grad_logical(), np.array([B, 0, B, 0, B, B], np.float32))


def test_losses_and_heads():
    from loans_amd import ops
    rng = np.random.RandomState(4)
    B, th, tw = 5, 6, 7
    theta = (rng.standard_normal((B, 2, 3)) * 0.9).astype(np.float32)
    grid, _ = C.st_grid_fwd(theta, (th, tw))
    gd = dev(grid)
    one = torch.ones((), device='cuda')
    l_ref, g_ref = C.direction_loss(grid, (48, 64))
    np.testing.assert_allclose(float(ops.grid_loss_fwd(gd, 0, 48, 64)), l_ref, rtol=1e-5)
    np.testing.assert_allclose(ops.grid_loss_bwd(gd, one, 0, 48, 64).cpu().numpy(), g_ref, rtol=1e-5, atol=1e-7)
    l_ref, g_ref = C.out_of_image_loss(grid)
    np.testing.assert_allclose(float(ops.grid_loss_fwd(gd, 1)), l_ref, rtol=1e-5)
    np.testing.assert_allclose(ops.grid_loss_bwd(gd, one, 1).cpu().numpy(), g_ref, rtol=1e-5, atol=1e-7)
    # mse
    y = rng.rand(B, 1).astype(np.float32); t = rng.rand(B, 1).astype(np.float32)
    np.testing.assert_allclose(float(ops.mse_fwd(dev(y), dev(t))), C.mse_fwd(y, t), rtol=1e-6)
    np.testing.assert_allclose(ops.mse_bwd(dev(y), one, dev(t)).cpu().numpy(), C.mse_bwd(y, t), rtol=1e-6)
    np.testing.assert_allclose(float(ops.mse_fwd(dev(y), None, 1.0)), C.mse_fwd(y, np.ones_like(y)), rtol=1e-6)
    # gap + linear (+ relu-in / sigmoid-out head)
    x = rng.standard_normal((B, 512, 3, 3)).astype(np.float32)
    np.testing.assert_allclose(ops.gap_fwd(dev(_nhwc(x))).cpu().numpy(), C.gap_fwd(x), rtol=1e-5, atol=1e-6)
    gp = rng.standard_normal((B, 512)).astype(np.float32)
    np.testing.assert_allclose(_nchw(ops.gap_bwd(dev(gp), (B, 3, 3, 512))), C.gap_bwd(x.shape, gp), rtol=1e-6)
    Wl = rng.standard_normal((6, 512)).astype(np.float32); bl = rng.standard_normal(6).astype(np.float32)
    h = rng.standard_normal((B, 512)).astype(np.float32)
    np.testing.assert_allclose(ops.linear_fwd(dev(h), dev(Wl), dev(bl)).cpu().numpy(), C.linear_fwd(h, Wl, bl), rtol=1e-4, atol=1e-5)
    gy = rng.standard_normal((B, 6)).astype(np.float32)
    gx_ref, gW_ref, gb_ref = C.linear_bwd(h, Wl, gy, True)
    gW, gb = torch.zeros(6, 512, device='cuda'), torch.zeros(6, device='cuda')
    gx = ops.linear_bwd(dev(h), dev(Wl), None, dev(gy), gW=gW, gb=gb)
    np.testing.assert_allclose(gx.cpu().numpy(), gx_ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gW.cpu().numpy(), gW_ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gb.cpu().numpy(), gb_ref, rtol=1e-4, atol=1e-5)
    K = 128 * 5 * 5
    hh = rng.standard_normal((B, K)).astype(np.float32); W4 = (0.02 * rng.standard_normal((1, K))).astype(np.float32)
    y_ref = C.sigmoid(C.linear_fwd(np.maximum(hh, 0), W4, None))
    yd = ops.linear_fwd(dev(hh), dev(W4), None, act_in=True, act_out=True)
    np.testing.assert_allclose(yd.cpu().numpy(), y_ref, rtol=1e-5)
    gy = rng.standard_normal((B, 1)).astype(np.float32)
    gz = C.sigmoid_bwd(y_ref, gy)
    gx_ref, gW_ref, _ = C.linear_bwd(np.maximum(hh, 0), W4, gz, False)
    gW = torch.zeros(1, K, device='cuda')
    gx = ops.linear_bwd(dev(hh), dev(W4), yd, dev(gy), gW=gW, act_in=True, act_out=True)
    np.testing.assert_allclose(gx.cpu().numpy(), gx_ref * (hh > 0), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gW.cpu().numpy(), gW_ref, rtol=1e-4, atol=1e-6)


def test_adam_amsgrad_matches_chainer_placement():
    from loans_amd import ops
    rng = np.random.RandomState(5)
    n = 1003
    p = rng.standard_normal(n).astype(np.float32)
    pr, m, v, vh = p.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    pd, md, vd, vhd = dev(p), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for t in range(1, 5):
        g = (rng.standard_normal(n) * 10.0 ** rng.uniform(-9, 0, n)).astype(np.float32)
        C.adam_amsgrad_update(pr, g, m, v, vh, t, alpha=1e-3)
        ops.adam_amsgrad(pd, dev(g), md, vd, vhd, C.adam_lr(1e-3, .9, .999, t), .9, .999, 1e-8, 1.0, 0.0)
        np.testing.assert_allclose(pd.cpu().numpy(), pr, rtol=0, atol=2e-7)
        np.testing.assert_allclose(vhd.cpu().numpy(), vh, rtol=1e-5, atol=1e-30)


def test_adam_without_amsgrad_and_update_with_a_loss_function():
    """The rest of chainer.optimizers.Adam's surface (stubs): amsgrad=False (Chainer's default) steps with
    sqrt(v) itself -- v falls when the gradients shrink, vhat would not --, and update(lossfun, *args) evaluates the loss,
    clears the gradients, runs the backward and steps."""
    import loans_amd
    from loans_amd import ops
    rng = np.random.RandomState(6)
    n = 1003
    p = rng.standard_normal(n).astype(np.float32)
    pr, m, v = p.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    pa, ma, va, vha = p.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    pd, md, vd = dev(p), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for t in range(1, 6):
        g = (rng.standard_normal(n) * 10.0 ** (-t)).astype(np.float32)          # shrinking gradients: v falls below its maximum
        # beta2 = 0.5: v follows the shrinking gradients within a step or two, vhat would stay at its first value
        C.adam_amsgrad_update(pr, g, m, v, None, t, alpha=1e-3, beta2=.5, amsgrad=False)
        C.adam_amsgrad_update(pa, g, ma, va, vha, t, alpha=1e-3, beta2=.5)
        ops.adam_amsgrad(pd, dev(g), md, vd, None, C.adam_lr(1e-3, .9, .5, t), .9, .5, 1e-8, 1.0, 0.0)
        np.testing.assert_allclose(pd.cpu().numpy(), pr, rtol=0, atol=2e-7)
        np.testing.assert_allclose(vd.cpu().numpy(), v, rtol=1e-5, atol=1e-30)
    assert np.abs(pr - pa).max() > 1e-4                                          # ... and that is not the AMSGrad trajectory
    # through the optimiser object, with a loss function: theta of a fresh localizer pulled towards zero
    np.random.seed(0)
    crop = (16, 16)
    loc = loans_amd.SheepLocalizer(crop)
    from loans_amd.datasets import synthetic
    frames = dev(synthetic.make_frames(1, 2, 64, 64))
    opt = loans_amd.Adam(alpha=1e-2).setup(loc)
    assert opt.hyperparam.amsgrad is False
    loc.finalize(torch.device('cuda', 0))
    loc.param_predictor.b.grad_view.fill_(123.0)                                # a stale gradient: update(lossfun) clears it

    def lossfun(images):
        rois, points = loc(images)
        return loans_amd.functions.mean_squared_error(loc.last_transform_params, torch.zeros(2, 2, 3, device='cuda'))
    b0 = loc.param_predictor.b.get_logical().copy()
    opt.update(lossfun, frames)
    b1 = loc.param_predictor.b.get_logical()
    # d mse / d b = 2 theta / 12 summed over the batch = [.8, 0, 0, 0, .8, 0] * 2 * 2 / 12 (the stale 123 was cleared);
    # step 1 of Adam moves every entry with a gradient by alpha against its sign, the masked entries not at all
    np.testing.assert_allclose(loc.param_predictor.b.grad_logical(), np.array([.8, 0, 0, 0, .8, 0]) * 4 / 12, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(b1 - b0, [-1e-2, 0, 0, 0, -1e-2, 0], rtol=1e-4, atol=1e-8)
    assert opt.t == 1


def _bf16_round(a):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (the conversion v_cvt_pk_bf16_f32 performs)."""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize("case", [(2, 64, 15, 13, 128, 3, 2, 1), (3, 128, 9, 9, 128, 4, 2, 1), (2, 3, 32, 32, 64, 7, 2, 3),
                                  (2, 256, 7, 7, 512, 3, 1, 1), (2, 64, 9, 10, 128, 1, 2, 0)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
def test_conv_bf16_compute_fprop_dgrad(case, tile):
    """bf16 compute arm: operands rounded to bf16 (RNE), products exact in fp32, fp32 accumulation -- so the
    result equals an fp64 convolution of the bf16-rounded tensors to fp32 accumulation accuracy."""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    rng = np.random.RandomState(7)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    cp = (Cin + 3) // 4 * 4
    geo = ops.ConvGeometry(B, H, W, cp, Cout, k, s, p)
    xd, wd = dev(_nhwc(x, cp)), dev(_ohwi(w, cp))
    xr, wr = _bf16_round(x).astype(np.float64), _bf16_round(w).astype(np.float64)
    y_ref, col = C.conv2d_fwd(xr, wr, None, s, p)
    ops.set_compute_dtype('bf16')
    try:
        stats_r = ops.stats_buffer(Cout, 'cuda')
        y = ops.conv_fprop(xd, wd, geo, stats=stats_r, tile=tile)
        assert rel_err(_nchw(y), y_ref) < 3e-6
        np.testing.assert_allclose(stats_r.sum(dim=0)[0].cpu().numpy(), y_ref.sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-3)
        assert rel_err(_nchw(y), C.conv2d_fwd(x.astype(np.float64), w.astype(np.float64), None, s, p)[0]) < 2e-2   # vs unrounded: bf16-level
        gy = rng.standard_normal(y_ref.shape).astype(np.float32)
        gx_ref, _, _ = C.conv2d_bwd(x.shape, col, wr, _bf16_round(gy).astype(np.float64), s, p, False)
        if Cin >= 32:
            gx = ops.conv_dgrad(dev(_nhwc(gy)), wd, geo, tile=tile)
            assert rel_err(_nchw(gx, Cin), gx_ref) < 3e-6
        if tile in (0, 1, 3):
            _, gw_ref, _ = C.conv2d_bwd(x.shape, col, wr, _bf16_round(gy).astype(np.float64), s, p, False, need_gx=False)
            dw = torch.zeros_like(wd)
            ops._conv_wgrad(xd, dev(_nhwc(gy)), dw, geo, False, 0, tile)
            assert rel_err(dw.cpu().numpy().transpose(0, 3, 1, 2)[:, :Cin], gw_ref) < 5e-6
    finally:
        ops.set_compute_dtype('f32')


@pytest.mark.parametrize("shape", [(2, 75, 75), (3, 16, 16), (2, 19, 23), (1, 33, 17), (2, 28, 24)])
@pytest.mark.parametrize("g16", [False, True])
def test_crop_dgrad_one_launch(shape, g16):
    """loans_crop_dgrad (csrc/cropgrad.hip): the gradient w.r.t. the 4-channel crops through r0.c0 (3x3 / 1) AND r0.cs
    (4x4 / 2) of the assessor (common/net.py:15,17,22-25) in one launch, against the oracle's two conv2d_bwd calls; the single
    convolution form; agreement with the per-class VALU kernel it replaces; fp32 and bf16 gradient tensors."""
    from loans_amd import ops
    B, H, W = shape
    Cc = 128
    rng = np.random.RandomState(H * 100 + W)
    ga, gb = ops.ConvGeometry(B, H, W, 4, Cc, 3, 1, 1), ops.ConvGeometry(B, H, W, 4, Cc, 4, 2, 1)
    wa = (rng.standard_normal((Cc, 3, 3, 3)) * 0.05).astype(np.float32)
    wb = (rng.standard_normal((Cc, 3, 4, 4)) * 0.05).astype(np.float32)
    gya = rng.standard_normal((B, Cc, ga.Ho, ga.Wo)).astype(np.float32)
    gyb = rng.standard_normal((B, Cc, gb.Ho, gb.Wo)).astype(np.float32)
    war, wbr = wa, wb
    if g16:         # bf16 gradients are contracted on bf16 MFMAs: the pre-pass rounds the weights (RNE), fp32 accumulation
        gya, gyb = _bf16_round(gya), _bf16_round(gyb)
        war, wbr = _bf16_round(wa), _bf16_round(wb)
    cast = (lambda t: t.to(torch.bfloat16)) if g16 else (lambda t: t)
    x_shape = (B, 3, H, W)
    _, col_a = C.conv2d_fwd(np.zeros(x_shape), war.astype(np.float64), None, 1, 1)
    _, col_b = C.conv2d_fwd(np.zeros(x_shape), wbr.astype(np.float64), None, 2, 1)
    ra = C.conv2d_bwd(x_shape, col_a, war.astype(np.float64), gya.astype(np.float64), 1, 1, False)[0]
    rb = C.conv2d_bwd(x_shape, col_b, wbr.astype(np.float64), gyb.astype(np.float64), 2, 1, False)[0]
    wad, wbd = dev(_ohwi(wa, 4)), dev(_ohwi(wb, 4))
    gad, gbd = cast(dev(_nhwc(gya))), cast(dev(_nhwc(gyb)))
    assert ops.crop_dgrad_ok(ga, gb)
    out = ops.crop_dgrad(gad, wad, ga, gbd, wbd, gb)
    assert out.dtype == torch.float32 and tuple(out.shape) == (B, H, W, 4)
    assert rel_err(_nchw(out, 3), ra + rb) < 5e-6
    assert not out[..., 3].any()                                   # the padding channel is written as zero
    # one convolution, with an addend that aliases nothing
    add = rng.standard_normal((B, H, W, 4)).astype(np.float32)
    add[..., 3] = 0
    one = ops.crop_dgrad(gbd, wbd, gb, addend=dev(add))
    assert rel_err(_nchw(one, 3), rb + add.transpose(0, 3, 1, 2)[:, :3]) < 5e-6
    # the kernel it replaces (one VALU launch per stride-parity class): same sums in another order
    old = ops.CROP_DGRAD
    try:
        ops.CROP_DGRAD = False
        ref = ops.conv_dgrad(gbd, wbd, gb)
        ops.conv_dgrad(gad, wad, ga, out=ref, addend=ref)
    finally:
        ops.CROP_DGRAD = old
    # (on bf16 gradients the VALU kernel reads the fp32 master weights, this one their bf16 roundings: 2^-9 per weight)
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < (4e-3 if g16 else 5e-6)
