"""-m gpu: the bf16 STORAGE arm (BASELINE configs 3 / 5) -- every kernel that reads or writes bf16 tensors, through the
C ABI, against the CPU oracle evaluated on the bf16-rounded operands.  A bf16 result is the fp32 result rounded once
(relative 2^-9 at most), so comparisons of bf16 outputs use 2^-8; fp32 outputs (weight gradients, statistics) keep
fp32-accumulation tolerances."""
import numpy as np
import pytest
import torch

from oracle import chainer_ops as C
from tests.gpu_util import dev, rel_err

pytestmark = pytest.mark.gpu

BF16_EPS = 2.0 ** -8


def _nhwc(x):
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)))


def _nchw(t):
    return t.detach().float().cpu().numpy().transpose(0, 3, 1, 2)


def _r(a):
    """round an fp32 array to bf16 (RNE) and return it as fp32"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def d16(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda().to(torch.bfloat16).contiguous()


CASES = [
    # B, Cin, H, W, Cout, k, stride, pad
    (3, 64, 14, 14, 64, 3, 1, 1),      # res2-like
    (2, 64, 15, 13, 128, 3, 2, 1),     # strided, odd sizes
    (5, 128, 6, 6, 200, 3, 1, 1),      # Cout not a multiple of the tile
    (2, 256, 7, 7, 512, 3, 2, 1),      # deep, strided
    (3, 256, 7, 7, 64, 1, 1, 0),       # 1x1 bottleneck reduce
    (2, 64, 9, 9, 256, 1, 2, 0),       # 1x1 strided (ResNet-50 shortcut)
    (2, 8, 11, 11, 16, 3, 1, 1),       # a single 16-byte unit per tap, K tail
    (3, 128, 20, 20, 512, 3, 1, 1),    # more than one 256-row tile, two 256-column tiles
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 7, 9, 43, 44, 33, 34, 35, 5])     # 5: a weight-gradient tile; 7, 9: 512 threads; 43 / 44: 9 with a ping-pong K loop (44: 16 x 16 x 32 MFMAs); +32: deep ring
def test_conv_bf16_storage(case, tile):
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    rng = np.random.RandomState(3)
    x = _r(rng.standard_normal((B, Cin, H, W)))
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    wr = _r(w)
    b = rng.standard_normal(Cout).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    xd, wd = d16(_nhwc(x)), dev(_nhwc(w))
    y_ref, col = C.conv2d_fwd(x.astype(np.float64), wr.astype(np.float64), b.astype(np.float64), s, p)
    stats_r = ops.stats_buffer(Cout, 'cuda')
    wtile, tile = tile, (tile if tile != 5 else 0)
    y = ops.conv_fprop(xd, wd, geo, bias=dev(b), stats=stats_r, tile=tile)
    assert y.dtype == torch.bfloat16
    assert np.abs(_nchw(y) - y_ref).max() <= BF16_EPS * np.abs(y_ref).max()
    assert rel_err(_nchw(y), y_ref) < BF16_EPS
    # statistics come from the fp32 accumulators, not from the rounded output
    st = stats_r.sum(dim=0).cpu().numpy()
    np.testing.assert_allclose(st[0], y_ref.sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1], (y_ref ** 2).sum(axis=(0, 2, 3)), rtol=1e-4)
    # + addend
    add = _r(rng.standard_normal(y_ref.shape))
    y2 = ops.conv_fprop(xd, wd, geo, addend=d16(_nhwc(add)), tile=tile)
    y2_ref = C.conv2d_fwd(x.astype(np.float64), wr.astype(np.float64), None, s, p)[0] + add
    assert rel_err(_nchw(y2), y2_ref) < BF16_EPS

    # pre-activation form (the assessor's blocks): conv(relu(x)) + addend
    xr_ = np.maximum(x, 0)
    y3 = ops.conv_fprop(xd, wd, geo, relu_in=True, addend=d16(_nhwc(add)), tile=tile)
    y3_ref, col_relu = C.conv2d_fwd(xr_.astype(np.float64), wr.astype(np.float64), None, s, p)
    assert rel_err(_nchw(y3), y3_ref + add) < BF16_EPS

    # data gradient and its fused epilogues
    gy = _r(rng.standard_normal(y_ref.shape))
    gx_ref, gw_ref, _ = C.conv2d_bwd(x.shape, col, wr.astype(np.float64), gy.astype(np.float64), s, p, False)
    gyd = d16(_nhwc(gy))
    gx = ops.conv_dgrad(gyd, wd, geo, tile=tile)
    assert gx.dtype == torch.bfloat16
    assert rel_err(_nchw(gx), gx_ref) < BF16_EPS
    ref_t = _r(rng.standard_normal(x.shape))
    addx = _r(rng.standard_normal(x.shape))
    if geo.dgrad_has_empty_class:       # strided 1x1: tap-less pixels only take the plain addend (as in fp32 storage)
        gx2 = ops.conv_dgrad(gyd, wd, geo, addend=d16(_nhwc(addx)), tile=tile)
        assert rel_err(_nchw(gx2), gx_ref + addx) < BF16_EPS
    else:
        gx2 = ops.conv_dgrad(gyd, wd, geo, mask_ref=d16(_nhwc(ref_t)), addend=d16(_nhwc(addx)), tile=tile)
        assert rel_err(_nchw(gx2), gx_ref * (ref_t > 0) + addx) < BF16_EPS
        gx3 = ops.conv_dgrad(gyd, wd, geo, addend=d16(_nhwc(addx)), addend_mask_ref=d16(_nhwc(ref_t)), tile=tile)
        assert rel_err(_nchw(gx3), gx_ref + addx * (ref_t > 0)) < BF16_EPS

    # weight gradient: bf16 operands, fp32 accumulation into the fp32 gradient
    if wtile in (0, 1, 3, 5, 9):
        dw = torch.zeros_like(wd)
        ops._conv_wgrad(xd, gyd, dw, geo, False, 0, wtile)
        ops._conv_wgrad(xd, gyd, dw, geo, False, 3, wtile)          # accumulates; explicit split count
        assert dw.dtype == torch.float32
        assert rel_err(dw.cpu().numpy().transpose(0, 3, 1, 2), 2 * gw_ref) < 1e-5
        _, gwr_ref, _ = C.conv2d_bwd(x.shape, col_relu, wr.astype(np.float64), gy.astype(np.float64), s, p, False, need_gx=False)
        dw.zero_()
        ops._conv_wgrad(xd, gyd, dw, geo, True, 0, wtile)           # weight gradient w.r.t. conv(relu(x))
        assert rel_err(dw.cpu().numpy().transpose(0, 3, 1, 2), gwr_ref) < 1e-5


@pytest.mark.parametrize("case", [(3, 64, 14, 14, 64), (2, 128, 20, 33, 128), (2, 64, 9, 50, 256), (1, 192, 8, 16, 64),
                                  (5, 64, 17, 12, 128), (2, 256, 32, 32, 256)])
@pytest.mark.parametrize("tile", [38, 39])
def test_weight_gradient_halo_tiles(case, tile):
    """LOANS_TILE_WGHALO_64 / _128 (csrc/wgrad_halo_bf16.hip): a block holds all nine taps of a stride-1 3x3 weight gradient
    and walks 8 x 16 pixel tiles with a staged halo image -- against the oracle on the bf16-rounded operands (fp32
    accumulation: 1e-5) and against the plain GEMM form; image sizes that are no multiples of the tile, several channel-tile
    pairs, explicit block counts, accumulation into a non-zero gradient, the relu(x) form of the assessor's blocks"""
    from loans_amd import ops
    B, Cin, H, W, Cout = case
    if tile == 39 and Cout % 128:
        pytest.skip('128 output channels per block')
    rng = np.random.RandomState(11)
    x = _r(rng.standard_normal((B, Cin, H, W)))
    gy = _r(rng.standard_normal((B, Cout, H, W)))
    w = np.zeros((Cout, Cin, 3, 3))
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, 3, 1, 1)
    assert tile in ops.wghalo_tiles(geo) or min(H, W) < 12
    _, col = C.conv2d_fwd(x.astype(np.float64), w, None, 1, 1)
    _, gw_ref, _ = C.conv2d_bwd(x.shape, col, w, gy.astype(np.float64), 1, 1, False, need_gx=False)
    xd, gyd = d16(_nhwc(x)), d16(_nhwc(gy))
    start = rng.standard_normal((Cout, 3, 3, Cin)).astype(np.float32)
    for splits in (0, 1, 5):
        dw = dev(start)
        ops._conv_wgrad(xd, gyd, dw, geo, False, splits, tile)
        got = (dw.cpu().numpy() - start).transpose(0, 3, 1, 2)
        assert rel_err(got, gw_ref) < 1e-5, (splits, rel_err(got, gw_ref))
    plain = torch.zeros((Cout, 3, 3, Cin), device='cuda')
    ops._conv_wgrad(xd, gyd, plain, geo, False, 0, 1)
    assert rel_err(got, plain.cpu().numpy().transpose(0, 3, 1, 2)) < 1e-5
    _, col_relu = C.conv2d_fwd(np.maximum(x, 0).astype(np.float64), w, None, 1, 1)
    _, gwr_ref, _ = C.conv2d_bwd(x.shape, col_relu, w, gy.astype(np.float64), 1, 1, False, need_gx=False)
    dw = torch.zeros((Cout, 3, 3, Cin), device='cuda')
    ops._conv_wgrad(xd, gyd, dw, geo, True, 0, tile)
    assert rel_err(dw.cpu().numpy().transpose(0, 3, 1, 2), gwr_ref) < 1e-5



@pytest.mark.parametrize("case", [(3, 64, 14, 14, 64, 3, 1, 1, 38), (2, 128, 20, 33, 128, 3, 1, 1, 39), (2, 128, 20, 33, 128, 3, 1, 1, 1),
                                  (4, 64, 16, 16, 256, 1, 1, 0, 1), (2, 64, 15, 13, 128, 3, 2, 1, 3), (2, 256, 8, 8, 256, 3, 1, 1, 9),
                                  (3, 128, 9, 9, 128, 4, 2, 1, 5)])
@pytest.mark.parametrize("splits", [0, 1, 7])
def test_weight_gradient_slabs_are_deterministic_and_equal_the_atomic_form(case, splits, monkeypatch):
    """loans_wgrad_bf16s_ws (round 5): blocks store raw partial tiles into workspace slabs, loans_fold_slabs_f32 adds them to dw in a
    fixed order.  Every tile form (plain GEMM tiles incl. the 512-thread one, the halo tiles), 1x1 / 3x3 / 4x4, strided: (a) two runs
    give BIT-IDENTICAL gradients; (b) the result equals the oracle on the bf16-rounded operands to 1e-5 like the atomic form,
    which it agrees with to fp32 rounding; (c) accumulation into a non-zero dw."""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, stride, pad, tile = case
    rng = np.random.RandomState(23)
    x = _r(rng.standard_normal((B, Cin, H, W)))
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, stride, pad)
    gy = _r(rng.standard_normal((B, Cout, geo.Ho, geo.Wo)))
    w = np.zeros((Cout, Cin, k, k))
    _, col = C.conv2d_fwd(x.astype(np.float64), w, None, stride, pad)
    _, gw_ref, _ = C.conv2d_bwd(x.shape, col, w, gy.astype(np.float64), stride, pad, False, need_gx=False)
    xd, gyd = d16(_nhwc(x)), d16(_nhwc(gy))
    assert ops.WGRAD_SLABS
    runs = []
    for _ in range(2):
        dw = torch.zeros((Cout, k, k, Cin), device='cuda')
        ops._conv_wgrad(xd, gyd, dw, geo, False, splits, tile)
        runs.append(dw)
    assert torch.equal(runs[0], runs[1])
    got = runs[0].cpu().numpy().transpose(0, 3, 1, 2)
    assert rel_err(got, gw_ref) < 1e-5, rel_err(got, gw_ref)
    start = rng.standard_normal((Cout, k, k, Cin)).astype(np.float32)
    dw = dev(start)
    ops._conv_wgrad(xd, gyd, dw, geo, False, splits, tile)
    assert rel_err((dw.cpu().numpy() - start).transpose(0, 3, 1, 2), gw_ref) < 1e-5
    monkeypatch.setattr(ops, 'WGRAD_SLABS', False)
    atomic = torch.zeros((Cout, k, k, Cin), device='cuda')
    ops._conv_wgrad(xd, gyd, atomic, geo, False, splits, tile)
    assert rel_err(got, atomic.cpu().numpy().transpose(0, 3, 1, 2)) < 1e-5


def test_fold_slabs_every_thread_shape():
    """loans_fold_slabs_f32 on every (units, slabs) regime of its thread map (1 .. 32 threads across the slabs of a unit), against a
    float64 sum; dst accumulates"""
    from loans_amd import _lib, ops
    import ctypes
    lib = _lib.load()
    rng = np.random.RandomState(3)
    for n, slabs in [(64, 1), (4096, 3), (36864, 512), (147456, 128), (10752, 341), (2359296, 8), (1024, 40), (4, 700)]:
        ws = rng.standard_normal((slabs, n)).astype(np.float32)
        base = rng.standard_normal(n).astype(np.float32)
        dst = dev(base)
        _lib.check(lib.loans_fold_slabs_f32(ops._ptr(dev(ws)), ops._ptr(dst), n, slabs, ops._stream()), 'loans_fold_slabs_f32')
        want = base.astype(np.float64) + ws.astype(np.float64).sum(0)
        np.testing.assert_allclose(dst.cpu().numpy(), want, rtol=0, atol=2e-6 * max(1.0, np.sqrt(slabs)) * 4)


def test_bn_passes_bf16_storage():
    """bn_apply (3 modes), bn_backward (single / dual, with and without the ReLU mask) on bf16 tensors against the
    fp32 kernels fed the same (already rounded) values"""
    from loans_amd import ops
    rng = np.random.RandomState(5)
    B, H, W, Cc = 4, 9, 7, 64
    mk = lambda: _r(rng.standard_normal((B, H, W, Cc)))      # noqa: E731
    x, x2, res, gy, out = mk(), mk(), mk(), mk(), mk()
    gamma = (1 + 0.1 * rng.standard_normal(Cc)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(Cc)).astype(np.float32)

    def state(a):
        stats = torch.zeros((ops.STATS_REPLICAS, 2, Cc), device='cuda', dtype=torch.float64)
        flat = torch.from_numpy(a.reshape(-1, Cc).astype(np.float64)).cuda()
        stats[0, 0], stats[0, 1] = flat.sum(0), (flat * flat).sum(0)
        return ops.bn_finalize(stats, B * H * W, dev(gamma), dev(beta), dev(np.zeros(Cc, np.float32)), dev(np.ones(Cc, np.float32)))
    st, st2 = state(x), state(x2)
    for kw32, kw16 in [({}, {}),
                       ({'residual': dev(res)}, {'residual': d16(res)}),
                       ({'x2': dev(x2), 'st2': st2}, {'x2': d16(x2), 'st2': st2})]:
        y32 = ops.bn_apply(dev(x), st, relu=True, **kw32)
        y16 = ops.bn_apply(d16(x), st, relu=True, **kw16)
        assert y16.dtype == torch.bfloat16
        assert rel_err(y16.float().cpu().numpy(), y32.cpu().numpy()) < BF16_EPS

    for mask32, mask16 in [(None, None), (dev(out), d16(out))]:
        gg32, gb32 = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
        gg16, gb16 = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
        g32 = ops.bn_backward(dev(gy), mask32, dev(x), st, dev(gamma), gg32, gb32)
        g16 = ops.bn_backward(d16(gy), mask16, d16(x), st, dev(gamma), gg16, gb16)
        assert g16.dtype == torch.bfloat16
        assert rel_err(g16.float().cpu().numpy(), g32.cpu().numpy()) < BF16_EPS
        np.testing.assert_allclose(gg16.cpu().numpy(), gg32.cpu().numpy(), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(gb16.cpu().numpy(), gb32.cpu().numpy(), rtol=1e-5, atol=1e-4)
    z = lambda: torch.zeros(Cc, device='cuda')      # noqa: E731
    a32, b32 = ops.bn_backward(dev(gy), dev(out), dev(x), st, dev(gamma), z(), z(),
                               x2=dev(x2), st2=st2, gamma2=dev(gamma), ggamma2=z(), gbeta2=z())
    a16, b16 = ops.bn_backward(d16(gy), d16(out), d16(x), st, dev(gamma), z(), z(),
                               x2=d16(x2), st2=st2, gamma2=dev(gamma), ggamma2=z(), gbeta2=z())
    assert rel_err(a16.float().cpu().numpy(), a32.cpu().numpy()) < BF16_EPS
    assert rel_err(b16.float().cpu().numpy(), b32.cpu().numpy()) < BF16_EPS


def test_region_boundaries_bf16_storage():
    """where the bf16 region begins and ends: the stem (fp32 frames -> bf16 conv output with fp32 statistics; pool and
    its backward on bf16; weight gradient from fp32 frames and a bf16 gradient; bias gradient) and global average
    pooling (bf16 -> fp32 features, fp32 -> bf16 gradient)"""
    from loans_amd import ops
    rng = np.random.RandomState(6)
    # --- pool forward / backward on bf16 against the fp32 kernels on the same (rounded) values
    B, H, W, Cc = 2, 13, 12, 64
    x = _r(rng.standard_normal((B, H, W, Cc)))
    stats = torch.zeros((ops.STATS_REPLICAS, 2, Cc), device='cuda', dtype=torch.float64)
    flat = torch.from_numpy(x.reshape(-1, Cc).astype(np.float64)).cuda()
    stats[0, 0], stats[0, 1] = flat.sum(0), (flat * flat).sum(0)
    st = ops.bn_finalize(stats, B * H * W, dev(np.ones(Cc, np.float32)), dev(np.zeros(Cc, np.float32)),
                         dev(np.zeros(Cc, np.float32)), dev(np.ones(Cc, np.float32)))
    y32, idx32 = ops.bn_relu_maxpool(dev(x), st)
    y16, idx16 = ops.bn_relu_maxpool(d16(x), st)
    assert y16.dtype == torch.bfloat16 and torch.equal(idx16, idx32)
    assert torch.equal(y16, y32.to(torch.bfloat16))                 # the fp32 result, rounded once
    gy = _r(rng.standard_normal(tuple(y32.shape)))
    g32 = ops.maxpool_relu_bwd(dev(gy), idx32, dev(x), st)
    g16 = ops.maxpool_relu_bwd(d16(gy), idx32, d16(x), st)
    assert g16.dtype == torch.bfloat16 and torch.equal(g16, g32.to(torch.bfloat16))
    cs32, cs16 = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    ops.colsum_acc(dev(x), cs32)
    ops.colsum_acc(d16(x), cs16)
    assert torch.allclose(cs16, cs32, rtol=1e-5, atol=1e-4)
    # --- global average pooling
    f = _r(rng.standard_normal((B, 7, 7, 512)))
    assert torch.allclose(ops.gap_fwd(d16(f)), ops.gap_fwd(dev(f)), rtol=1e-6, atol=1e-6)
    gf = rng.standard_normal((B, 512)).astype(np.float32)
    g16 = ops.gap_bwd(dev(gf), (B, 7, 7, 512), torch.bfloat16)
    assert torch.equal(g16, ops.gap_bwd(dev(gf), (B, 7, 7, 512)).to(torch.bfloat16))
    # --- stem conv (dense-row fp32 frames) with a bf16 output, and its weight gradient from a bf16 gradient
    Bs, Hs = 2, 32
    frames = rng.uniform(0, 1, (Bs, 3, Hs, Hs)).astype(np.float32)
    geo = ops.ConvGeometry(Bs, Hs, Hs, 3, 64, 7, 2, 3, dense=True)
    xp = ops.prep_images(dev(frames), geo)
    wd = dev((0.05 * rng.standard_normal((64, 7, geo.kwp, 3))).astype(np.float32))
    bias = dev(rng.standard_normal(64).astype(np.float32))
    ops.set_compute_dtype('bf16')
    try:
        s32, s16 = ops.stats_buffer(64, 'cuda'), ops.stats_buffer(64, 'cuda')
        c32 = ops.conv_fprop(xp, wd, geo, bias=bias, stats=s32, tile=3)
        c16 = ops.conv_fprop(xp, wd, geo, bias=bias, stats=s16, tile=3, out_bf16=True)
        assert c16.dtype == torch.bfloat16 and torch.equal(c16, c32.to(torch.bfloat16))
        assert torch.equal(s16, s32)
        gc = _r(rng.standard_normal(tuple(c32.shape)))
        dw32, dw16 = torch.zeros_like(wd), torch.zeros_like(wd)
        ops._conv_wgrad(xp, dev(gc), dw32, geo, False, 0, 3)
        ops._conv_wgrad(xp, d16(gc), dw16, geo, False, 0, 3)
        assert rel_err(dw16.cpu().numpy(), dw32.cpu().numpy()) < 1e-5
        # --- the same stem on bf16 FRAMES (loans_prep_images_dense_bf16 -> loans_igemm_bf16s / loans_wgrad_bf16s with
        #     LOANS_F_DENSE): the operands are the ones the fp32-frame kernels round while staging, only the summation
        #     order differs
        ops.set_storage_dtype('bf16')
        xp16 = ops.prep_images(dev(frames), geo)
        assert xp16.dtype == torch.bfloat16 and xp16.frame_hw == (Hs, Hs) and torch.equal(xp16, xp.to(torch.bfloat16))
        for t in (1, 2, 3, 4, 7, 9):
            sb = ops.stats_buffer(64, 'cuda')
            cb = ops.conv_fprop(xp16, wd, geo, bias=bias, stats=sb, tile=t)
            assert cb.dtype == torch.bfloat16
            assert rel_err(cb.float().cpu().numpy(), c32.cpu().numpy()) < 2 ** -7, t
            assert rel_err(sb.sum(0).cpu().numpy(), s32.sum(0).cpu().numpy()) < 1e-5, t
        assert ops.stem16_wgrad_ok(geo)
        for t in (1, 3, 5, ops.TILE_STEM):
            dwb = torch.zeros_like(wd)
            ops._conv_wgrad(xp16, d16(gc), dwb, geo, False, 0, t)
            assert rel_err(dwb.cpu().numpy(), dw32.cpu().numpy()) < 1e-5, t
    finally:
        ops.set_compute_dtype('f32')


@pytest.mark.parametrize("case", [(3, 96, 96), (2, 64, 160), (5, 32, 32), (1, 224, 224), (300, 32, 64)])
def test_stem_weight_gradient_direct_bf16(case, monkeypatch):
    """LOANS_TILE_STEM of loans_wgrad_bf16s (csrc/stem.hip: stem7_wgrad_bf16_kernel, round 5): conv1's weight gradient from the bf16
    frame buffer and a bf16 gradient as a direct, persistent kernel, against the implicit-GEMM tile on the same operands (fp32 sums of
    the same products: 1e-5) -- several units per block and fewer units than blocks, runs that start at every 4-byte alignment, more
    images than compute units; the slab form is bit-reproducible, accumulates into dw and leaves the window-padding columns alone;
    the atomic form (no workspace) agrees."""
    from loans_amd import ops
    B, H, W = case
    rng = np.random.RandomState(B + H)
    ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
    try:
        geo = ops.ConvGeometry(B, H, W, 3, 64, 7, 2, 3, dense=True)
        assert ops.stem16_wgrad_ok(geo)
        xp = ops.prep_images(dev(rng.uniform(-1, 1, (B, 3, H, W)).astype(np.float32)), geo)
        assert xp.dtype == torch.bfloat16
        gy = d16(rng.standard_normal((B, geo.Ho, geo.Wo, 64)).astype(np.float32))
        ref = torch.zeros((64, 7, geo.kwp, 3), device='cuda')
        ops._conv_wgrad(xp, gy, ref, geo, False, 0, 3)
        assert float(ref.abs().max()) > 0 and float(ref[:, :, 7:].abs().max()) == 0      # column 7 of a row is window padding
        runs = []
        for _ in range(2):
            dw = torch.zeros_like(ref)
            ops._conv_wgrad(xp, gy, dw, geo, False, 0, ops.TILE_STEM)
            runs.append(dw)
        assert torch.equal(runs[0], runs[1])
        assert rel_err(runs[0].cpu().numpy(), ref.cpu().numpy()) < 1e-5
        start = dev(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
        dw = start.clone()
        ops._conv_wgrad(xp, gy, dw, geo, False, 0, ops.TILE_STEM)
        assert torch.equal(dw[:, :, 7:], start[:, :, 7:])
        assert rel_err((dw - start)[:, :, :7].cpu().numpy(), ref[:, :, :7].cpu().numpy()) < 1e-5
        monkeypatch.setattr(ops, 'WGRAD_SLABS', False)
        atomic = torch.zeros_like(ref)
        ops._conv_wgrad(xp, gy, atomic, geo, False, 0, ops.TILE_STEM)
        assert rel_err(atomic.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    finally:
        ops.set_compute_dtype('f32'); ops.set_storage_dtype('f32')


def test_stem_weight_gradient_direct_bf16_refuses_what_it_does_not_cover():
    """frames whose output rows are not whole 16-pixel steps (200 px -> 100) stay on the implicit-GEMM tiles: the autotuner is not
    offered the direct kernel, and asking for it by name is an error, not a wrong answer"""
    from loans_amd import ops
    ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
    try:
        geo = ops.ConvGeometry(2, 200, 200, 3, 64, 7, 2, 3, dense=True)
        assert not ops.stem16_wgrad_ok(geo)
        xp = ops.prep_images(torch.rand((2, 3, 200, 200), device='cuda'), geo)
        gy = torch.randn((2, geo.Ho, geo.Wo, 64), device='cuda').to(torch.bfloat16)
        with pytest.raises(Exception):
            ops._conv_wgrad(xp, gy, torch.zeros((64, 7, geo.kwp, 3), device='cuda'), geo, False, 0, ops.TILE_STEM)
    finally:
        ops.set_compute_dtype('f32'); ops.set_storage_dtype('f32')


def test_assessor_edges_bf16_storage():
    """the assessor's ends in the bf16-storage arm: the gradient w.r.t. the 4-channel fp32 crops from a bf16 gradient
    (loans_dgrad_c4_bf16_f32), and the sigmoid(Linear(relu(h))) head on a bf16 feature map"""
    from loans_amd import ops
    rng = np.random.RandomState(9)
    B, H, W, Cout, k, s, p = 2, 19, 19, 128, 4, 2, 1
    geo = ops.ConvGeometry(B, H, W, 4, Cout, k, s, p)
    w = (rng.standard_normal((Cout, k, k, 4)) * 0.05).astype(np.float32)
    w[..., 3] = 0
    gy = _r(rng.standard_normal((B, geo.Ho, geo.Wo, Cout)))
    ref_t = rng.standard_normal((B, H, W, 4)).astype(np.float32)
    g32 = ops.conv_dgrad(dev(gy), dev(w), geo, mask_ref=dev(ref_t))
    g16 = ops.conv_dgrad(d16(gy), dev(w), geo, mask_ref=dev(ref_t))
    assert g16.dtype == torch.float32 and torch.allclose(g16, g32, rtol=1e-5, atol=1e-5)
    K = 18 * 18 * 128
    x = _r(rng.standard_normal((3, K)))
    Wl = (rng.standard_normal((1, K)) * 0.02).astype(np.float32)
    y32 = ops.linear_fwd(dev(x), dev(Wl), None, act_in=True, act_out=True)
    y16 = ops.linear_fwd(d16(x), dev(Wl), None, act_in=True, act_out=True)
    assert torch.allclose(y16, y32, rtol=1e-5, atol=1e-6)
    gyl = rng.standard_normal((3, 1)).astype(np.float32)
    gW32, gW16 = torch.zeros(1, K, device='cuda'), torch.zeros(1, K, device='cuda')
    gx32 = ops.linear_bwd(dev(x), dev(Wl), y32, dev(gyl), gW=gW32, act_in=True, act_out=True)
    gx16 = ops.linear_bwd(d16(x), dev(Wl), y16, dev(gyl), gW=gW16, act_in=True, act_out=True)
    assert gx16.dtype == torch.bfloat16 and torch.equal(gx16, gx32.to(torch.bfloat16))
    assert torch.allclose(gW16, gW32, rtol=1e-5, atol=1e-6)


HALO_CASES = [
    # B, Cin, H, W, Cout, k, pad
    (3, 64, 14, 14, 64, 3, 1),       # res2-like, one tile row ragged
    (2, 64, 33, 21, 64, 3, 1),       # several tiles per image, ragged both ways
    (2, 128, 16, 16, 128, 3, 1),     # two channel chunks, exact tiles
    (2, 256, 9, 19, 192, 3, 1),      # four chunks, Cout not a multiple of the tile
    (2, 128, 18, 18, 128, 3, 1),     # the assessor's r2 / r3 geometry
    (2, 64, 12, 17, 128, 1, 0),      # 1x1: no halo
    (1, 64, 40, 48, 64, 3, 1),       # 16 x 16 tiles, three per row
    (3, 64, 21, 35, 32, 3, 1),       # fewer output channels than a block holds
    (1, 256, 20, 33, 512, 3, 1),     # res4 / res5-like widths: two 256-column blocks, ragged 16 x 16 tiles, four chunks
]


@pytest.mark.parametrize("case", HALO_CASES)
@pytest.mark.parametrize("tile", [11, 12, 13, 14, 15, 36, 37, 42])
def test_conv_halo_tiles_bf16_storage(case, tile):
    """LOANS_TILE_HALO_* (csrc/halo_bf16.hip): stride-1 convolutions and their data gradients with the input tile staged once
    per 64-channel chunk -- against the oracle on the bf16-rounded operands and against the implicit-GEMM tile (bit for bit
    where K is one chunk per tap, Cin = 64; to the position of rare roundings otherwise), with every epilogue flag."""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, p = case
    if tile in (12, 14, 15, 37) and Cin != 64:
        pytest.skip('one-chunk (Cin = 64) forms')
    if tile in (15, 37) and (Cout > 64 or k != 3):
        pytest.skip('LOANS_TILE_WS64 / LOANS_TILE_WSW64: 3x3, Cin = 64, Cout <= 64')
    rng = np.random.RandomState(11)
    x = _r(rng.standard_normal((B, Cin, H, W)))
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    wr = _r(w)
    b = rng.standard_normal(Cout).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, 1, p)
    xd, wd = d16(_nhwc(x)), dev(_nhwc(w))
    y_ref, col = C.conv2d_fwd(x.astype(np.float64), wr.astype(np.float64), b.astype(np.float64), 1, p)

    def same(a, b_, exact):
        if exact:
            assert torch.equal(a, b_)
        else:       # another fp32 summation order under one bf16 rounding: a few results land on the neighbouring bf16 value
            assert float((a != b_).float().mean()) < 0.02
            assert rel_err(a.float().cpu().numpy(), b_.float().cpu().numpy()) < BF16_EPS
    exact_f = Cin == 64
    s_h, s_g = ops.stats_buffer(Cout, 'cuda'), ops.stats_buffer(Cout, 'cuda')
    y = ops.conv_fprop(xd, wd, geo, bias=dev(b), stats=s_h, tile=tile)
    y_g = ops.conv_fprop(xd, wd, geo, bias=dev(b), stats=s_g, tile=1)
    assert y.dtype == torch.bfloat16 and rel_err(_nchw(y), y_ref) < BF16_EPS
    same(y, y_g, exact_f)
    np.testing.assert_allclose(s_h.sum(0).cpu().numpy(), s_g.sum(0).cpu().numpy(), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(s_h.sum(0).cpu().numpy()[0], y_ref.sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    add = _r(rng.standard_normal(y_ref.shape))
    relu_in = tile not in (15, 37)                        # the weight-stationary kernels have no pre-activation form
    y3 = ops.conv_fprop(xd, wd, geo, relu_in=relu_in, addend=d16(_nhwc(add)), tile=tile)
    y3_ref = C.conv2d_fwd((np.maximum(x, 0) if relu_in else x).astype(np.float64), wr.astype(np.float64), None, 1, p)[0] + add
    assert rel_err(_nchw(y3), y3_ref) < BF16_EPS
    same(y3, ops.conv_fprop(xd, wd, geo, relu_in=relu_in, addend=d16(_nhwc(add)), tile=1), exact_f)

    # data gradient: the gathered tensor is gy (Cout channels), the "output channels" are Cin
    if Cout % 64 or (tile in (12, 14, 15, 37) and Cout != 64):
        return
    gy = _r(rng.standard_normal(y_ref.shape))
    gx_ref = C.conv2d_bwd(x.shape, col, wr.astype(np.float64), gy.astype(np.float64), 1, p, False)[0]
    gyd = d16(_nhwc(gy))
    exact_d = Cout == 64
    gx = ops.conv_dgrad(gyd, wd, geo, tile=tile)
    assert rel_err(_nchw(gx), gx_ref) < BF16_EPS
    same(gx, ops.conv_dgrad(gyd, wd, geo, tile=1), exact_d)
    ref_t, addx = _r(rng.standard_normal(x.shape)), _r(rng.standard_normal(x.shape))
    gx2 = ops.conv_dgrad(gyd, wd, geo, mask_ref=d16(_nhwc(ref_t)), addend=d16(_nhwc(addx)), tile=tile)
    assert rel_err(_nchw(gx2), gx_ref * (ref_t > 0) + addx) < BF16_EPS
    gx3 = ops.conv_dgrad(gyd, wd, geo, addend=d16(_nhwc(addx)), addend_mask_ref=d16(_nhwc(ref_t)), tile=tile)
    assert rel_err(_nchw(gx3), gx_ref + addx * (ref_t > 0)) < BF16_EPS
    same(gx3, ops.conv_dgrad(gyd, wd, geo, addend=d16(_nhwc(addx)), addend_mask_ref=d16(_nhwc(ref_t)), tile=1), exact_d)


@pytest.mark.parametrize("case", [(2, 64, 16, 16, 256), (3, 64, 7, 9, 256), (1, 64, 5, 5, 64), (2, 128, 12, 20, 512), (3, 128, 3, 11, 128),
                                  (1, 64, 1, 1, 512), (2, 64, 33, 47, 192), (4, 128, 64, 64, 512),
                                  (2, 256, 8, 8, 1024), (3, 256, 5, 7, 256), (1, 256, 3, 3, 128), (5, 256, 32, 32, 1024)])
def test_conv_pointwise_tile_bf16_storage(case):
    """LOANS_TILE_PW (csrc/pw_bf16.hip): the short-K 1 x 1 convolutions (ResNet-50's res2 / res3 expansions) with the operands
    fed global -> VGPR and the weights in fragment order -- against the oracle on the bf16-rounded operands, and BIT FOR BIT
    against the implicit-GEMM tile (same products, same order); the BN statistics to fp32 summation order.  Ragged last strips
    (pixel counts that are no multiple of 32), one-pixel inputs, more strips than resident waves."""
    from loans_amd import ops
    B, Cin, H, W, Cout = case
    rng = np.random.RandomState(5)
    x = _r(rng.standard_normal((B, Cin, H, W)))
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, 1, 1, 0)
    assert ops._pw_tiles(geo, True) == (ops.TILE_PW,) and ops._pw_tiles(geo, False) == ()
    xd, wd = d16(_nhwc(x)), dev(_nhwc(w))
    y_ref = C.conv2d_fwd(x.astype(np.float64), _r(w).astype(np.float64), None, 1, 0)[0]
    s_p, s_g = ops.stats_buffer(Cout, 'cuda'), ops.stats_buffer(Cout, 'cuda')
    sentinel = torch.full((B * H * W * Cout + 64,), 7.0, device='cuda', dtype=torch.bfloat16)
    y = ops.conv_fprop(xd, wd, geo, out=sentinel[:B * H * W * Cout].view(B, H, W, Cout), stats=s_p, tile=ops.TILE_PW)
    y_g = ops.conv_fprop(xd, wd, geo, stats=s_g, tile=1)
    assert y.dtype == torch.bfloat16 and rel_err(_nchw(y), y_ref) < BF16_EPS
    assert torch.equal(y, y_g)
    assert bool((sentinel[B * H * W * Cout:] == 7.0).all())            # nothing written past the last pixel
    sp, sg = s_p.sum(0).cpu().numpy(), s_g.sum(0).cpu().numpy()
    np.testing.assert_allclose(sp, sg, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(sp[0], y_ref.sum(axis=(0, 2, 3)), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(sp[1], (y_ref ** 2).sum(axis=(0, 2, 3)), rtol=1e-4, atol=2e-3)
    assert torch.equal(ops.conv_fprop(xd, wd, geo, tile=ops.TILE_PW), y_g)       # the form without statistics
    # the statistics do not depend on the order in which waves and blocks arrive (fp64 from the per-lane sums on): the BN
    # coefficients, hence the forward, are the same bit for bit in every run (the fixture deterministic_forward relies on it)
    for _ in range(3):
        s_r = ops.stats_buffer(Cout, 'cuda')
        ops.conv_fprop(xd, wd, geo, stats=s_r, tile=ops.TILE_PW)
        assert torch.equal(s_r.sum(0), s_p.sum(0))
    # an epilogue it does not have is refused, not ignored
    with pytest.raises(RuntimeError):
        ops.conv_fprop(xd, wd, geo, bias=dev(np.zeros(Cout, np.float32)), tile=ops.TILE_PW)


def test_pointwise_weights_are_packed_by_the_steps_preparation():
    """Inside a step the fragment-order weights of every LOANS_TILE_PW layer come from ONE launch at begin_step
    (loans_pw_pack_batch_f32, from the fp32 masters): a layer is packed per call only the first time it asks; the batch follows
    the weights from step to step, takes layers that join later, and forgets layers no step has used lately."""
    from loans_amd import ops
    device = torch.device('cuda', 0)
    g = torch.Generator(device='cuda').manual_seed(3)
    layers = []
    for B, H, Cin, Cout in ((2, 16, 64, 256), (3, 9, 128, 512), (1, 12, 256, 1024)):
        geo = ops.ConvGeometry(B, H, H, Cin, Cout, 1, 1, 0)
        x = torch.randn((B, H, H, Cin), device='cuda', generator=g).to(torch.bfloat16)
        w = torch.randn((Cout, 1, 1, Cin), device='cuda', generator=g) / Cin ** 0.5
        layers.append((geo, x, w))

    def step(active, expect_packs):
        refs = [ops.conv_fprop(x, w, geo, tile=1) for geo, x, w in active]           # outside the step: cast per call
        ops.begin_step(device)
        try:
            before = ops.PW_PACK_CALLS
            for (geo, x, w), ref in zip(active, refs):
                assert torch.equal(ops.conv_fprop(x, w, geo, tile=ops.TILE_PW), ref)
            assert ops.PW_PACK_CALLS - before == expect_packs
        finally:
            ops.end_step()

    step(layers[:1], 1)                     # first sight: registered, packed per call
    for _, _, w in layers:
        w.mul_(1.25)                        # "the optimisers have moved the weights"
    step(layers[:1], 0)                     # from begin_step's batch, with the new values
    step(layers[:2], 1)                     # a layer that joins later is packed per call once ...
    for _, _, w in layers:
        w.add_(0.01)
    step(layers[:2], 0)                     # ... and rides in the batch from then on
    step(layers[1:2], 0)
    step(layers, 1)                         # Cin = 256 is packed per call in every step (its packing launch is its L2 warm-up)
    step(layers, 1)
    wp = ops._weight_preps[0]
    assert len(wp.pw) == 2
    for _ in range(4):                      # entries no step has asked for lately go, with their buffers
        ops.begin_step(device)
        ops.end_step()
    assert not wp.pw and not wp.pw_order
    step(layers[:1], 1)


def test_conv_pointwise_tile_more_strips_than_waves():
    """LOANS_TILE_PW with more 32-pixel strips than the grid has waves (every wave walks several strips: the persistent loop, the
    prefetch of the next strip's pixels), at ResNet-50's three expansion shapes -- bit for bit against the implicit-GEMM tile."""
    from loans_amd import ops
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for Cin, Cout, per_cu in ((64, 256, 16), (128, 512, 8), (256, 1024, 8)):
        pixels = 32 * (cus * per_cu * 2 + 3) + 5                      # two strips and a bit per resident wave, a ragged last one
        geo = ops.ConvGeometry(1, 1, pixels, Cin, Cout, 1, 1, 0)
        g = torch.Generator(device='cuda').manual_seed(Cin)
        x = torch.randn((1, 1, pixels, Cin), device='cuda', generator=g).to(torch.bfloat16)
        w = torch.randn((Cout, 1, 1, Cin), device='cuda', generator=g) / Cin ** 0.5
        s_p, s_g = ops.stats_buffer(Cout, 'cuda'), ops.stats_buffer(Cout, 'cuda')
        y = ops.conv_fprop(x, w, geo, stats=s_p, tile=ops.TILE_PW)
        assert torch.equal(y, ops.conv_fprop(x, w, geo, stats=s_g, tile=1))
        np.testing.assert_allclose(s_p.sum(0).cpu().numpy(), s_g.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("case", [(2, 256, 7, 7, 512, 3, 2, 1), (3, 512, 4, 4, 512, 3, 1, 1), (2, 128, 9, 9, 128, 4, 2, 1),
                                  (2, 512, 8, 8, 64, 3, 1, 1)])
@pytest.mark.parametrize("tile", [3 | (2 << 8), 3 | (4 << 8), 2 | (8 << 8)])
def test_conv_split_k_bf16_storage(case, tile):
    """split-K on bf16 storage (loans_igemm_bf16s_splitk + loans_igemm_finalize_bf16): raw fp32 partial tiles added into a
    workspace, one finalize pass with bias / statistics / mask / addend and the single rounding to bf16 -- forward and data
    gradient (all stride-parity classes into one workspace) against the oracle on the rounded operands and against the
    unsplit kernel (another summation order under one rounding)."""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    rng = np.random.RandomState(17)
    x = _r(rng.standard_normal((B, Cin, H, W)))
    w = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    wr = _r(w)
    b = rng.standard_normal(Cout).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    xd, wd = d16(_nhwc(x)), dev(_nhwc(w))
    y_ref, col = C.conv2d_fwd(x.astype(np.float64), wr.astype(np.float64), b.astype(np.float64), s, p)
    s_k, s_1 = ops.stats_buffer(Cout, 'cuda'), ops.stats_buffer(Cout, 'cuda')
    y = ops.conv_fprop(xd, wd, geo, bias=dev(b), stats=s_k, tile=tile)
    y1 = ops.conv_fprop(xd, wd, geo, bias=dev(b), stats=s_1, tile=3)
    assert y.dtype == torch.bfloat16 and rel_err(_nchw(y), y_ref) < BF16_EPS
    assert float((y != y1).float().mean()) < 0.02
    np.testing.assert_allclose(s_k.sum(0).cpu().numpy(), s_1.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    add = _r(rng.standard_normal(y_ref.shape))
    y2 = ops.conv_fprop(xd, wd, geo, relu_in=True, addend=d16(_nhwc(add)), tile=tile)
    y2_ref = C.conv2d_fwd(np.maximum(x, 0).astype(np.float64), wr.astype(np.float64), None, s, p)[0] + add
    assert rel_err(_nchw(y2), y2_ref) < BF16_EPS
    gy = _r(rng.standard_normal(y_ref.shape))
    gx_ref = C.conv2d_bwd(x.shape, col, wr.astype(np.float64), gy.astype(np.float64), s, p, False)[0]
    gyd = d16(_nhwc(gy))
    gx = ops.conv_dgrad(gyd, wd, geo, tile=tile)
    assert gx.dtype == torch.bfloat16 and rel_err(_nchw(gx), gx_ref) < BF16_EPS
    ref_t, addx = _r(rng.standard_normal(x.shape)), _r(rng.standard_normal(x.shape))
    gx2 = ops.conv_dgrad(gyd, wd, geo, mask_ref=d16(_nhwc(ref_t)), addend=d16(_nhwc(addx)), tile=tile)
    assert rel_err(_nchw(gx2), gx_ref * (ref_t > 0) + addx) < BF16_EPS
    gx3 = ops.conv_dgrad(gyd, wd, geo, addend=d16(_nhwc(addx)), addend_mask_ref=d16(_nhwc(ref_t)), tile=tile)
    assert rel_err(_nchw(gx3), gx_ref + addx * (ref_t > 0)) < BF16_EPS


@pytest.mark.parametrize("wtile", [15, 37])
def test_ws64_persistent_blocks_walk_many_tiles(wtile):
    """LOANS_TILE_WS64 / LOANS_TILE_WSW64 with more tiles than CUs (and, for the wave-autonomous form, more units than waves): every block walks several tiles through its two image buffers; forward with
    statistics and bias, data gradient with the masked-addend epilogue -- bit for bit the implicit GEMM's results (same K order
    at Cin = 64), statistics to fp64-summation accuracy."""
    from loans_amd import ops
    g = torch.Generator(device='cuda').manual_seed(5)
    B, H, W = 24, 120, 136                                # 24 x 8 x 9 = 1728 tiles of 16 x 16, ragged on both edges
    geo = ops.ConvGeometry(B, H, W, 64, 64, 3, 1, 1)
    x = torch.randn(B, H, W, 64, device='cuda', generator=g).to(torch.bfloat16)
    w = torch.randn(64, 3, 3, 64, device='cuda', generator=g) * 0.05
    bias = torch.randn(64, device='cuda', generator=g)
    s1, s2 = ops.stats_buffer(64, 'cuda'), ops.stats_buffer(64, 'cuda')
    y_ws = ops.conv_fprop(x, w, geo, bias=bias, stats=s1, tile=wtile)
    y_ig = ops.conv_fprop(x, w, geo, bias=bias, stats=s2, tile=1)
    assert torch.equal(y_ws, y_ig)
    np.testing.assert_allclose(s1.sum(0).cpu().numpy(), s2.sum(0).cpu().numpy(), rtol=1e-6, atol=1e-3)
    gy = torch.randn(B, H, W, 64, device='cuda', generator=g).to(torch.bfloat16)
    ref = torch.randn(B, H, W, 64, device='cuda', generator=g).to(torch.bfloat16)
    add = torch.randn(B, H, W, 64, device='cuda', generator=g).to(torch.bfloat16)
    assert torch.equal(ops.conv_dgrad(gy, w, geo, addend=add, addend_mask_ref=ref, tile=wtile),
                       ops.conv_dgrad(gy, w, geo, addend=add, addend_mask_ref=ref, tile=1))
    assert torch.equal(ops.conv_dgrad(gy, w, geo, mask_ref=ref, tile=wtile), ops.conv_dgrad(gy, w, geo, mask_ref=ref, tile=1))


@pytest.mark.parametrize("case", [(2, 32, 32), (3, 64, 64), (2, 96, 80), (1, 130, 66), (2, 224, 224), (1, 512, 512), (2, 50, 34), (1, 22, 30)])
def test_stem_direct_bf16(case):
    """LOANS_TILE_STEM on the bf16 MFMA (csrc/stem.hip, stem7_bf16_kernel): conv1 with bias and BN statistics as a persistent
    direct convolution, from the fp32 frame buffer (loans_igemm_bf16_f32, operands rounded while staged) and from the bf16 one
    (loans_igemm_bf16s) -- against the oracle convolution of the bf16-rounded operands (fp64 sums; the kernel differs by its
    fp32 accumulation and the final rounding to bf16) and against the implicit-GEMM arm.  Frame sizes with R = 4, 2, 1 rows
    per unit, ragged last pixel tiles, units that start 4-byte aligned only, more units than blocks (512 x 512 is config 3)."""
    from loans_amd import ops
    B, H, W = case
    rng = np.random.RandomState(H * 7 + W)
    x = _r(rng.uniform(-1, 1, (B, 3, H, W)))
    w = _r(rng.standard_normal((64, 3, 7, 7)) / np.sqrt(147))
    b = rng.standard_normal(64).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, 3, 64, 7, 2, 3, dense=True)
    assert ops.stem16_tile_rows(geo) > 0
    xp = np.zeros((B, geo.Hp, geo.Wp, 3), np.float32)
    xp[:, 3:3 + H, 3:3 + W] = x.transpose(0, 2, 3, 1)
    wp = np.zeros((64, 7, geo.kwp, 3), np.float32)
    wp[:, :, :7] = w.transpose(0, 2, 3, 1)
    y_ref, _ = C.conv2d_fwd(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), 2, 3)
    y_ref = y_ref.transpose(0, 2, 3, 1)
    xd, wd, bd = dev(xp), dev(wp), dev(b)
    ops.set_compute_dtype('bf16')
    try:
        outs = []
        for src, wt in ((xd, wd), (d16(xp), d16(wp))):
            for tile in (ops.TILE_STEM, 3):
                st = ops.stats_buffer(64, 'cuda')
                y = ops.conv_fprop(src, wt, geo, bias=bd, stats=st, tile=tile, **({} if src.dtype == torch.bfloat16 else dict(out_bf16=True)))
                assert y.dtype == torch.bfloat16
                outs.append((y, st.sum(dim=0)))
        for y, st in outs:
            # bf16 output: half an ulp of the rounding (2^-9 relative) on top of the fp32 sums
            err = np.abs(y.float().cpu().numpy() - y_ref)
            assert (err <= 2.0 ** -8 * np.abs(y_ref) + 1e-5).all(), err.max()
            np.testing.assert_allclose(st[0].cpu().numpy(), y_ref.sum(axis=(0, 1, 2)), rtol=1e-5, atol=2e-3)
            np.testing.assert_allclose(st[1].cpu().numpy(), (y_ref ** 2).sum(axis=(0, 1, 2)), rtol=2e-5)
        # same operands, same arithmetic, another summation order: the direct and the implicit-GEMM results differ in a few
        # last-place roundings at most; the two input forms of the direct kernel are the SAME computation
        assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
        flips = (outs[0][0] != outs[1][0]).float().mean().item()
        assert flips < 0.02, flips
        y_nb = ops.conv_fprop(d16(xp), d16(wp), geo, tile=ops.TILE_STEM)            # no bias, no statistics
        ref_nb = torch.from_numpy((y_ref - b).astype(np.float32)).cuda()
        assert (y_nb.float() - ref_nb).abs().max().item() <= 2.0 ** -8 * ref_nb.abs().max().item() + 1e-5
    finally:
        ops.set_compute_dtype('f32')



@pytest.mark.parametrize("case", [(2, 64, 15, 13, 128, 3, 2, 1), (3, 128, 12, 12, 256, 3, 2, 1), (2, 64, 9, 9, 64, 1, 2, 0), (5, 32, 8, 8, 96, 3, 2, 1)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 7, 9, 43, 44, 33, 35])
def test_conv_pair_bf16_storage(case, tile):
    """loans_igemm_pair_bf16s: a unit's first convolution and its conv shortcut as ONE GEMM with the weights stacked along N.
    Every output element is contracted in the order of the single launch with the same tile, so the two tensors are
    bit-identical to two loans_igemm_bf16s calls; the BN statistics agree to the fp64 atomics' summation order."""
    from loans_amd import ops
    B, Cin, H, W, Cout, k, s, p = case
    rng = np.random.RandomState(11)
    geo_a = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    geo_b = ops.ConvGeometry(B, H, W, Cin, Cout, k, s, p)
    x = d16(rng.standard_normal((B, H, W, Cin)))
    wa = dev((rng.standard_normal((Cout, k, k, Cin)) / np.sqrt(Cin * k * k)).astype(np.float32))
    wb = dev((rng.standard_normal((Cout, k, k, Cin)) / np.sqrt(Cin * k * k)).astype(np.float32))
    assert ops.fprop_pair_ok(x, geo_a, geo_b)
    sa, sb = ops.stats_buffer(Cout, 'cuda'), ops.stats_buffer(Cout, 'cuda')
    ya, yb = ops.conv_fprop_pair(x, wa, wb, geo_a, geo_b, sa, sb, tile=tile)
    assert ya.dtype == torch.bfloat16 and ya.is_contiguous() and yb.is_contiguous()
    if tile:
        ra, rb = ops.stats_buffer(Cout, 'cuda'), ops.stats_buffer(Cout, 'cuda')
        ea = ops.conv_fprop(x, wa, geo_a, stats=ra, tile=tile)
        eb = ops.conv_fprop(x, wb, geo_b, stats=rb, tile=tile)
        assert torch.equal(ya, ea) and torch.equal(yb, eb)
        assert torch.allclose(sa.sum(0), ra.sum(0), rtol=1e-12, atol=1e-9) and torch.allclose(sb.sum(0), rb.sum(0), rtol=1e-12, atol=1e-9)
    # against the oracle on the bf16-rounded operands
    xr = x.float().cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)
    for y, w, st in ((ya, wa, sa), (yb, wb, sb)):
        wr = w.to(torch.bfloat16).float().cpu().numpy().transpose(0, 3, 1, 2).astype(np.float64)
        y_ref, _ = C.conv2d_fwd(xr, wr, None, s, p)
        y_ref = y_ref.transpose(0, 2, 3, 1)
        err = np.abs(y.float().cpu().numpy() - y_ref)
        assert (err <= 2.0 ** -8 * np.abs(y_ref) + 1e-5).all(), err.max()
        np.testing.assert_allclose(st.sum(0)[0].cpu().numpy(), y_ref.sum(axis=(0, 1, 2)), rtol=1e-5, atol=2e-3)
    y2a, y2b = ops.conv_fprop_pair(x, wa, wb, geo_a, geo_b, tile=tile)           # without statistics
    assert torch.equal(y2a, ya) and torch.equal(y2b, yb)


@pytest.mark.parametrize("storage,tiles", [('f32', [0, 1, 2, 3, 4, 6, 17, 18, 19, 20]),
                                           ('bf16', [0, 1, 2, 3, 4, 7, 9, 43, 44, 33, 34, 35, 11, 12, 13, 14, 36, 37, 42])])
@pytest.mark.parametrize("case", [(3, 64, 14, 14, 64, 3), (2, 128, 20, 33, 128, 3), (2, 256, 9, 12, 64, 3), (5, 64, 17, 12, 256, 1),
                                  (2, 512, 6, 7, 128, 1)])
def test_data_gradient_takes_the_bn_backward_sums(storage, tiles, case):
    """LOANS_F_BNSUMS (round 3): the data gradient that feeds a BatchNormalization followed by its own ReLU
    (sheep/resnet.py:137,157) takes that BN's two backward sums -- sum g m, sum g m (y - mean), m = (y scale + shift > 0) --
    in its epilogue, from the tile it is about to store, and the BN backward becomes ONE pass (`bn_backward_from_sums`).
    Every tile form that carries the flag, both storage types: the gradient tensor is bit-identical to the plain launch, the
    BN backward equals the two-pass form (`bn_backward(..., mask_is_own_relu=True)`) to the order of the sums, and both equal
    the oracle's bn_bwd of the masked gradient."""
    from loans_amd import ops
    B, C_, H, W, Cout, k = case          # C_ = channels of the gradient / of the BN; the conv maps C_ -> Cout
    s16 = storage == 'bf16'
    rng = np.random.RandomState(17)
    rnd = _r if s16 else (lambda a: a.astype(np.float32))
    up = d16 if s16 else dev
    gy = rnd(rng.standard_normal((B, H, W, Cout)))
    y = rnd(1.5 * rng.standard_normal((B, H, W, C_)) + 0.3 * rng.standard_normal(C_))
    w = (rng.standard_normal((Cout, k, k, C_)) / np.sqrt(C_ * k * k)).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(C_)).astype(np.float32)
    beta = (0.3 * rng.standard_normal(C_)).astype(np.float32)
    geo = ops.ConvGeometry(B, H, W, C_, Cout, k, 1, k // 2)
    stats = torch.zeros((ops.STATS_REPLICAS, 2, C_), device='cuda', dtype=torch.float64)
    flat = torch.from_numpy(y.reshape(-1, C_).astype(np.float64)).cuda()
    stats[0, 0], stats[0, 1] = flat.sum(0), (flat * flat).sum(0)
    st = ops.bn_finalize(stats, B * H * W, dev(gamma), dev(beta), dev(np.zeros(C_, np.float32)), dev(np.ones(C_, np.float32)))
    gyd, yd, wd = up(gy), up(y), dev(w)
    act = ops.bn_apply(yd, st, relu=True)
    assert ops.bn_sums_ok(geo, yd)
    old = (ops.COMPUTE, ops.STORAGE)
    if s16:
        ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
    try:
        ran = 0
        for tile in tiles:
            if s16 and tile in (11, 12, 13, 14, 36, 37, 42) and tile not in ops._halo_tiles(geo, geo.Cout, geo.Cin, (geo.H, geo.W)):
                continue
            if s16 and tile in (9, 43, 44) and C_ % 256:
                continue
            plain = ops.conv_dgrad(gyd, wd, geo, tile=tile)
            z = lambda: torch.zeros(C_, device='cuda')      # noqa: E731
            gg0, gb0, gg1, gb1 = z(), z(), z(), z()
            two_pass = ops.bn_backward(plain, act, yd, st, dev(gamma), gg0, gb0, mask_is_own_relu=True)
            fused, sums = ops.conv_dgrad(gyd, wd, geo, tile=tile, bn_sums=(yd, st))
            if tile == 0:       # the autotuner's picks, made separately for the two forms: a halo tile walks K chunk-major, the
                # implicit GEMM tap-major -- equal to fp32 rounding of the accumulators, bit for bit only on the SAME tile
                assert rel_err(fused.float().cpu().numpy(), plain.float().cpu().numpy()) < (BF16_EPS if s16 else 1e-6), tile
            else:
                assert torch.equal(fused, plain), tile
            one_pass = ops.bn_backward_from_sums(fused, yd, st, sums, dev(gamma), gg1, gb1)
            tol = 2 * BF16_EPS if s16 else 2e-5
            assert rel_err(one_pass.float().cpu().numpy(), two_pass.float().cpu().numpy()) < tol, tile
            np.testing.assert_allclose(gg1.cpu().numpy(), gg0.cpu().numpy(), rtol=2e-4, atol=2e-3 * float(gg0.abs().max()))
            np.testing.assert_allclose(gb1.cpu().numpy(), gb0.cpu().numpy(), rtol=2e-4, atol=2e-3 * float(gb0.abs().max()))
            ran += 1
        assert ran >= 6
        # ... and against the oracle: bn_bwd of the masked gradient
        g = plain.float().cpu().numpy().astype(np.float64)
        yy = yd.float().cpu().numpy().astype(np.float64)
        mean, var = yy.reshape(-1, C_).mean(0), yy.reshape(-1, C_).var(0)
        rstd = 1 / np.sqrt(var + ops.BN_EPS)
        xhat = (yy - mean) * rstd
        gm = g * ((xhat * gamma + beta) > 0)
        n = B * H * W
        ref = gamma * rstd * (gm - gm.reshape(-1, C_).sum(0) / n - xhat * (gm * xhat).reshape(-1, C_).sum(0) / n)
        assert rel_err(one_pass.float().cpu().numpy(), ref) < (2 * BF16_EPS if s16 else 1e-4)
        np.testing.assert_allclose(gg1.cpu().numpy(), (gm * xhat).reshape(-1, C_).sum(0), rtol=1e-3, atol=1e-3 * np.abs(gm).sum() / C_)
    finally:
        ops.set_compute_dtype(old[0])
        if old[1] == 'bf16':
            ops.set_storage_dtype('bf16')


@pytest.mark.parametrize("case", [(2, 64, 16, 16, 256), (3, 64, 7, 9, 256), (1, 64, 5, 5, 64), (2, 128, 12, 20, 512), (3, 128, 3, 11, 128),
                                  (2, 64, 33, 47, 192), (2, 256, 8, 8, 1024), (3, 256, 5, 7, 256), (4, 128, 64, 64, 512)])
def test_bn_relu_on_load_pointwise_forward_and_weight_gradient(case):
    """LOANS_F_AFFINE_IN (round 5): a 1 x 1 convolution that reads the INPUT of the BatchNormalization + ReLU in
    front of it and applies them on load -- forward (LOANS_TILE_PW) and weight gradient (every GEMM tile) -- is, BIT FOR BIT, the
    apply pass followed by the plain launch: outputs, BN statistics (to the order of the fp64 atomics), weight gradients.  Ragged last
    strips and pixel counts beyond the blocks' slices stay zero operands (relu(shift) of a row that does not exist must not enter)."""
    from loans_amd import ops
    B, Cin, H, W, Cout = case
    rng = np.random.RandomState(11)
    x = d16(rng.standard_normal((B, H, W, Cin)).astype(np.float32) * 2)
    w = dev((rng.standard_normal((Cout, 1, 1, Cin)) / np.sqrt(Cin)).astype(np.float32))
    gamma, beta = dev((1 + 0.3 * rng.standard_normal(Cin)).astype(np.float32)), dev((0.3 * rng.standard_normal(Cin)).astype(np.float32))
    stats = ops.stats_buffer(Cin, 'cuda')
    stats[0, 0] = x.double().sum(dim=(0, 1, 2)); stats[0, 1] = (x.double() ** 2).sum(dim=(0, 1, 2))
    st = ops.bn_finalize(stats, B * H * W, gamma, beta, torch.zeros(Cin, device='cuda'), torch.ones(Cin, device='cuda'))
    geo = ops.ConvGeometry(B, H, W, Cin, Cout, 1, 1, 0)
    assert ops.affine_in_ok(geo, x)
    a = ops.bn_apply(x, st, relu=True)
    assert float((a == 0).float().mean()) > 0.1            # the ReLU does something here
    s_ref, s_new = ops.stats_buffer(Cout, 'cuda'), ops.stats_buffer(Cout, 'cuda')
    y_ref = ops.conv_fprop(a, w, geo, stats=s_ref, tile=ops.TILE_PW)
    y_new = ops.conv_fprop_affine(x, st, w, geo, stats=s_new)
    assert torch.equal(y_new, y_ref)
    np.testing.assert_allclose(s_new.sum(0).cpu().numpy(), s_ref.sum(0).cpu().numpy(), rtol=1e-12, atol=1e-9)
    assert torch.equal(ops.conv_fprop_affine(x, st, w, geo), y_ref)          # the form without statistics
    gy = d16(rng.standard_normal((B, H, W, Cout)).astype(np.float32))
    tiles = [1, 3, 5] + ([9] if Cout % 256 == 0 else [])
    for tile in tiles:
        for splits in (0, 3):
            dw_ref, dw_new = torch.zeros((Cout, 1, 1, Cin), device='cuda'), torch.zeros((Cout, 1, 1, Cin), device='cuda')
            ops._conv_wgrad(a, gy, dw_ref, geo, False, splits, tile)
            ops._conv_wgrad(x, gy, dw_new, geo, False, splits, tile, in_affine=st)
            assert torch.equal(dw_new, dw_ref), (tile, splits)
    # the flag is refused where no kernel implements it (a 3 x 3 geometry, another tile), not ignored
    with pytest.raises(RuntimeError):
        d3 = ops.ConvGeometry(B, H, W, Cin, Cout, 3, 1, 1)
        ops._conv_wgrad(x, gy, torch.zeros((Cout, 3, 3, Cin), device='cuda'), d3, False, 0, 1, in_affine=st)


def test_bottleneck_with_bn_on_load_is_the_materialised_form(monkeypatch):
    """A ResNet-50 bottleneck whose bn2 -> relu -> conv3 never writes the activation between them (ops.BN_ON_LOAD) against the same
    unit with the apply pass: output, input gradient and every parameter gradient bit for bit."""
    import loans_amd
    from loans_amd import ops
    from loans_amd.iou.iou_regressor import BottleneckB
    from loans_amd.runtime.core import Variable
    rng = np.random.RandomState(13)
    B, H, W, cout, mid = 4, 12, 10, 256, 64
    x = d16(rng.standard_normal((B, H, W, cout)).astype(np.float32))
    gy = d16(rng.standard_normal((B, H, W, cout)).astype(np.float32))
    res = []
    loans_amd.set_compute_dtype('bf16')
    loans_amd.set_storage_dtype('bf16')
    try:
        for on_load in (False, True):
            monkeypatch.setattr(ops, 'BN_ON_LOAD', on_load)
            np.random.seed(14)
            blk = BottleneckB(cout, mid, loans_amd.links.HeNormal())
            r2 = np.random.RandomState(15)
            for key, p in blk.namedparams():
                if key.endswith('/gamma'):
                    p.set_logical((1 + 0.2 * r2.standard_normal(p.logical_shape)).astype(np.float32))
                elif key.endswith('/beta'):
                    p.set_logical((0.2 * r2.standard_normal(p.logical_shape)).astype(np.float32))
            blk.finalize(torch.device('cuda', 0))
            # (the materialised arm's conv3 on the SAME kernel: another tile gives the same outputs, but its BN statistics in
            # another fp32 order, and bn3's coefficients then differ in the last bit)
            g3 = blk.conv3.geometry(B, H, W)
            monkeypatch.setitem(g3.tuned, ops._variant(g3, 'fprop16', True, False, True, False), ops.TILE_PW)
            xv = Variable(x.clone(), requires_grad=True)
            out = blk(xv)
            fn = out.creator
            assert fn.onload == [False, False, on_load] and (fn.h[1] is None) == on_load
            out.grad = gy.clone()
            blk.cleargrads()
            out.backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            res.append((out.data.clone(), xv.grad.clone(), {k: p.grad_logical().copy() for k, p in blk.namedparams()}))
    finally:
        loans_amd.set_compute_dtype('f32')
    (o0, g0, p0), (o1, g1, p1) = res
    assert torch.equal(o1, o0) and torch.equal(g1, g0)
    for k in p0:
        np.testing.assert_array_equal(p1[k], p0[k], err_msg=k)
