"""-m gpu: host-runtime behaviour that only shows with device tensors -- gradient accumulation into shared buffers, the step
accumulator pool next to a captured hipGraph, Adam on parameters whose gradient disappears (frames no longer tall enough for
res6 / res7): Chainer steps them with zeros, their moments keep decaying (GradientMethod.reallocate_cleared_grads)."""
import numpy as np
import pytest
import torch

import loans_amd
from loans_amd import ops
from loans_amd.functions import reshape
from loans_amd.functions.basic import add
from loans_amd.runtime.core import Variable
from oracle import model as M
from tests.gpu_util import build_pair, dev, inputs, oracle_params
from tests.test_gpu_model import _updater

pytestmark = pytest.mark.gpu


def test_backward_fan_out_below_an_add_does_not_alias():
    """Add.backward hands the SAME tensor to both inputs and Reshape.backward returns a view of its own: a variable with a
    second consumer must not accumulate into that shared tensor in place."""
    g0 = torch.arange(12, dtype=torch.float32, device='cuda').reshape(3, 4) + 1
    p = Variable(torch.ones(12, device='cuda'), requires_grad=True)
    q = Variable(torch.full((12,), 2.0, device='cuda'), requires_grad=True)
    a, c = reshape(p, (3, 4)), reshape(q, (3, 4))
    z = add(a, c)
    w = add(a, z)                      # a feeds z and w
    w.grad = g0.clone()
    w.backward()
    torch.cuda.synchronize()
    assert torch.equal(w.grad, g0)                                    # the seed gradient was not written to
    assert torch.equal(q.grad.reshape(3, 4), g0)                      # c's branch sees g, not 2 g
    assert torch.equal(p.grad.reshape(3, 4), 2 * g0)                  # a = both paths
    np.testing.assert_array_equal(w.data.cpu().numpy(), np.full((3, 4), 1 + 1 + 2.0))


def test_zero_pool_never_moves_under_a_captured_graph():
    pool = ops._ZeroPool()
    dev_ = torch.device('cuda', 0)
    pool.begin(dev_)
    a = pool.take(100)
    assert a is not None and not pool.pinned
    ptr = pool.buf.data_ptr()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pool.begin(dev_)                                              # a captured step's memset
        b = pool.take(64)
        b.add_(1.0)
    assert pool.pinned and pool.buf.data_ptr() == ptr
    # an eager step that needs far more than the buffer holds (taller frames): what does not fit comes from torch.zeros
    pool.begin(dev_)
    assert pool.take(pool.buf.numel() * 4) is None
    pool.begin(dev_)                                                  # ... and the next begin() must NOT reallocate
    assert pool.buf.data_ptr() == ptr and pool.take(10) is not None
    g.replay()
    torch.cuda.synchronize()
    assert float(b.sum()) == 64.0                                     # the graph still owns live memory


def test_adam_keeps_stepping_parameters_that_lost_their_gradient(deterministic_forward):
    """One step on 232 px frames (res6 trains), one on 64 px frames (res6 has no gradient any more): Chainer updates res6 with
    a zero gradient in the second step -- its first moment decays, it keeps moving.  The oracle's AdamAMSGrad does exactly
    that; the HIP optimiser must too (and res7, never trained, must stay bit-identical)."""
    crop = (16, 16)
    loc, dis = build_pair(81, crop)
    tall, real, labels = inputs(82, 3, 232, 226, crop)
    small = inputs(83, 3, 64, 64, crop)[0]
    with loans_amd.using_config('enable_backprop', False):
        dis(dev(real))
    loc.finalize(torch.device('cuda', 0))
    lp, dp = oracle_params(loc, np.float64), oracle_params(dis, np.float64)
    init = {k: v.copy() for k, v in lp.items()}
    og, od = M.AdamAMSGrad(lp), M.AdamAMSGrad(dp)
    f64 = lambda a: a.astype(np.float64)            # noqa: E731
    M.update_core(lp, dp, og, od, f64(tall), f64(real), f64(labels), crop, rng=np.random.RandomState(0))
    after1 = {k: v.copy() for k, v in lp.items()}
    M.update_core(lp, dp, og, od, f64(small), f64(real), f64(labels), crop, rng=np.random.RandomState(0))

    upd = _updater(loc, dis, tall, real, labels)
    upd.get_iterator('main').batches = [dev(tall), dev(small)]
    upd.update()
    s1 = loc.state_dict_chainer()
    upd.update()
    s2 = loc.state_dict_chainer()
    k6, k7 = 'res6/0/conv1/W', 'res7/0/conv1/W'
    np.testing.assert_array_equal(s2[k7], init[k7].astype(np.float32))                    # never had a gradient: untouched
    assert np.abs(after1[k6] - init[k6]).max() > 5e-4                                     # step 1 trained res6
    d_ref = lp[k6] - after1[k6]                                                           # step 2: zero gradient, m decays
    assert np.abs(d_ref).max() > 1e-4
    d_hip = s2[k6].astype(np.float64) - s1[k6].astype(np.float64)
    assert np.abs(d_hip).max() > 1e-4, 'res6 froze when its gradient disappeared'
    # the second move is m / (sqrt(vhat) + eps) of step 1's gradient: sign-like again; same criterion as the update tests
    off = np.abs(d_hip - d_ref)
    assert off.max() < 1.5e-3 and np.mean(off > 5e-5) < 5e-3, (off.max(), np.mean(off > 5e-5))
    for key in ('res6/1/bn2/gamma', 'res6/0/bn3/beta'):
        assert np.abs(s2[key] - s1[key]).max() > 0


def test_weight_preparation_once_per_step():
    """ops._WeightPrep: inside a step the data gradients take their re-packed weights from ONE batched launch at the step's
    start (loans_repack_dgrad_batch; bf16 arm: forward weights from the arena's bf16 shadow) instead of per-class launches --
    same bytes as the per-call form, refreshed after the weights moved, per-call again outside a step"""
    import numpy as np
    import torch
    from loans_amd import _lib, ops
    from tests.gpu_util import dev
    rng = np.random.RandomState(2)
    dev0 = torch.device('cuda', 0)
    for bf16 in (False, True):
        geo = ops.ConvGeometry(2, 10, 12, 64, 128, 3, 2, 1)          # strided: four stride-parity classes
        geo1 = ops.ConvGeometry(2, 10, 12, 64, 128, 3, 1, 1)
        w = dev(rng.standard_normal((128, 3, 3, 64)).astype(np.float32))

        def per_call(g):
            out = torch.empty(g.dgrad_weight_floats, device='cuda', dtype=torch.bfloat16 if bf16 else torch.float32)
            fn = _lib.load().loans_repack_dgrad_bf16 if bf16 else _lib.load().loans_repack_dgrad_f32
            for d, tapsel, off in g.dgrad:
                ops.check(fn(w.data_ptr(), out[off:].data_ptr(), g.Cout, g.Cin, 9, tapsel, d.ntaps, ops._stream()), 'repack')
            return out
        assert ops._prepacked_dgrad_weights(w, geo, bf16) is None                      # outside a step
        ops.begin_step(dev0)
        first = [ops._prepacked_dgrad_weights(w, g, bf16) for g in (geo, geo1)]          # registered, filled per call
        assert all(torch.equal(a, per_call(g)) for a, g in zip(first, (geo, geo1)))
        ops.end_step()
        w.mul_(-0.5)                                                                     # "the optimiser's update"
        ops.begin_step(dev0)                                                             # one launch fills both
        wp = ops._weight_preps[0]
        assert {(w.data_ptr(), geo.key, bf16), (w.data_ptr(), geo1.key, bf16)} <= wp.prepared
        again = [ops._prepacked_dgrad_weights(w, g, bf16) for g in (geo, geo1)]
        assert all(a.data_ptr() == b.data_ptr() for a, b in zip(first, again))          # persistent buffers
        assert all(torch.equal(a, per_call(g)) for a, g in zip(again, (geo, geo1)))
        ops.end_step()
        assert ops._prepacked_dgrad_weights(w, geo, bf16) is None
        for _ in range(4):                                                               # unused for some steps: dropped
            ops.begin_step(dev0)
            ops.end_step()
        assert (w.data_ptr(), geo.key, bf16) not in ops._weight_preps[0].repacks


def test_weight_gradients_rotate_over_their_streams(monkeypatch):
    """LOANS_WGRAD_STREAMS (round 3; two streams by default on the fp32 kernels): consecutive weight gradients go to different
    streams, `join_side_stream` / the staged exchange's `_side_stream` wait for all of them, and the gradients are those of the
    single-stream order."""
    rng = np.random.RandomState(5)
    geo = ops.ConvGeometry(4, 20, 24, 64, 64, 3, 1, 1)
    x = dev(rng.standard_normal((4, 20, 24, 64)).astype(np.float32))
    gys = [dev(rng.standard_normal((4, 20, 24, 64)).astype(np.float32)) for _ in range(4)]
    ops.conv_wgrad(x, gys[0], torch.zeros(64, 3, 3, 64, device='cuda'), geo)        # autotune outside the comparison
    ops.join_side_stream()
    torch.cuda.synchronize()
    got = {}
    for n in ('1', '3'):
        monkeypatch.setattr(ops, '_WGRAD_STREAMS_ENV', n)
        used = []
        real = ops._conv_wgrad
        # (a tuned weight gradient is launched on its side stream by handle: `stream=`, not torch's current stream)
        monkeypatch.setattr(ops, '_conv_wgrad', lambda *a, **k: (used.append(k.get('stream') or torch.cuda.current_stream().cuda_stream), real(*a, **k))[1])
        dws = [torch.zeros(64, 3, 3, 64, device='cuda') for _ in gys]
        for gy, dw in zip(gys, dws):
            ops.conv_wgrad(x, gy, dw, geo)
        assert ops._side_dirty
        ops.join_side_stream()          # the current stream now waits for every one of them
        got[n] = [dw.clone() for dw in dws]
        assert not ops._side_dirty
        monkeypatch.setattr(ops, '_conv_wgrad', real)
        assert len(set(used)) == int(n) and torch.cuda.current_stream().cuda_stream not in used, used
        if n == '3':
            assert used[0] != used[1] != used[2] and used[3] == used[0]
    for a, b in zip(got['1'], got['3']):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-4)


def test_bf16_shadow_covers_stages_that_wake_up_between_steps(deterministic_forward):
    """ADVICE (round 3): begin_step cast only the arena's ACTIVE prefix -- the previous step's -- into the bf16 shadow, while the
    localizer sets this step's prefix afterwards, from the frame height.  A 64 px step followed by a 320 px step then read
    res6 / res7 from shadow memory no cast had written.  The shadow now covers the whole arena: the sequence short, tall,
    short, tall gives, bit for bit, what per-call casts (LOANS_WEIGHT_PREP=0) give."""
    crop = (16, 16)
    small = inputs(301, 2, 64, 64, crop)
    tall = inputs(302, 2, 320, 320, crop)
    loans_amd.set_compute_dtype('bf16')
    loans_amd.set_storage_dtype('bf16')
    old = ops.WEIGHT_PREP
    try:
        runs = []
        for prep in (True, False):
            ops.WEIGHT_PREP = prep
            loc, dis = build_pair(303, crop)
            with loans_amd.using_config('enable_backprop', False):
                dis(dev(small[1]))
            upd = _updater(loc, dis, *small, lr=1e-4)
            batches = [small, tall, small, tall]
            upd.get_iterator('main').batches = [dev(b[0]) for b in batches]
            upd.get_iterator('real').batches = [(dev(b[1]), dev(b[2])) for b in batches]
            losses = []
            for _ in batches:
                upd.update()
                obs = loans_amd.reporter.observation
                losses.append((float(obs['loss_localizer']), float(obs['loss_dis'])))
            assert loc.arena.active_numel == loc.arena.numel                  # the last step was a tall one
            runs.append((losses, loc.state_dict_chainer()))
        (l_prep, p_prep), (l_call, p_call) = runs
        assert np.isfinite(l_prep).all()
        assert l_prep[0] == l_call[0]                                          # same weights, deterministic forward
        # later steps start from weights whose gradients were summed with float atomics: last-bit differences between two runs
        np.testing.assert_allclose(np.array(l_prep), np.array(l_call), rtol=4e-2)
        # and directly: a step that begins with the SHORT prefix active serves res7's weights from its shadow all the same
        ops.WEIGHT_PREP = True
        loc.arena.set_active('res6')
        ops.begin_step(torch.device('cuda', 0))
        loc.arena.set_active(None)
        w = loc.res7[1].conv2.W.data
        sh = ops._bf16_shadow(w)
        ops.end_step()
        assert sh is not None and torch.equal(sh, w.to(torch.bfloat16))
        for k in ('res7/1/conv2/W', 'res6/0/conv1/W', 'feature_extractor/res5/1/conv2/W', 'param_predictor/W'):
            # (weight gradients are summed with float atomics: last-bit differences between two runs; Adam's step is sign-like,
            # so an entry whose gradient is rounding noise may walk the other way: at most both runs' four steps apart)
            d = np.abs(p_prep[k] - p_call[k])
            assert d.max() <= 3 * 4 * 1e-4, (k, float(d.max()))
    finally:
        ops.WEIGHT_PREP = old
        loans_amd.set_compute_dtype('f32')


def test_captured_step_survives_eager_steps_of_another_shape(deterministic_forward):
    """ADVICE (round 3): the captured step's loans_repack_dgrad_batch launch has the job table and the re-pack buffers baked in.
    Eager fallback steps on another input shape add entries (a NEW table) and, after three of them, used to evict the graph's
    entries as stale and free their buffers: the next replay then read a freed table and wrote into freed memory.  Tables and
    entries that were current during a capture are now kept.  Sequence: 2 warm-up + 1 captured + 1 replay at 64 px, four eager
    steps at 96 px, two replays at 64 px -- against the same sequence run eagerly."""
    crop = (16, 16)
    a = inputs(311, 4, 64, 64, crop)
    b = inputs(312, 4, 96, 96, crop)
    seq = [a, a, a, a, b, b, b, b, a, a]
    runs = []
    for use_graph in (False, True):
        # the well-conditioned start of test_graph_captured_step_matches_eager: a fresh localizer (theta = 0.8 x identity: no
        # corner near the image border, where the out-of-image term has its kink) with a small seeded param_predictor.W
        np.random.seed(313)
        loc, dis = loans_amd.SheepLocalizer(crop), loans_amd.ResnetAssessor()
        loc.param_predictor.W.set_logical((2e-3 * np.random.RandomState(3).standard_normal((6, 512))).astype(np.float32))
        with loans_amd.using_config('enable_backprop', False):
            dis(dev(a[1]))
        upd = _updater(loc, dis, *a, lr=1e-4, use_graph=use_graph)
        upd.get_iterator('main').batches = [dev(s[0]) for s in seq]
        upd.get_iterator('real').batches = [(dev(s[1]), dev(s[2])) for s in seq]
        losses = []
        for i, _ in enumerate(seq):
            upd.update()
            obs = loans_amd.reporter.observation
            losses.append((float(obs['loss_localizer']), float(obs['loss_dis'])))
            if use_graph and i == 7:
                wp = ops._weight_preps[0]
                assert wp.captured_tables and wp.captured_keys <= set(wp.repacks)      # the graph's entries are still there
                assert not any(t is wp.table for t in wp.captured_tables)              # ... beside a newer table
        assert (upd._graph is not None) == use_graph
        runs.append((losses, loc.state_dict_chainer()))
    (l0, p0), (l1, p1) = runs
    # (four frames per step: a ReLU / pooling decision that flips on the last bit of a weight -- both runs sum their weight
    # gradients with float atomics -- moves a later loss by 1e-3; what a freed table or buffer would do is garbage or a fault)
    np.testing.assert_allclose(np.array(l1)[0], np.array(l0)[0], rtol=1e-6)             # same weights, deterministic forward
    np.testing.assert_allclose(np.array(l1), np.array(l0), rtol=4e-2, atol=1e-6)       # (measured over tile assignments: <= 1.2e-2)
    # Adam's step is sign-like: an entry whose gradient is rounding noise may walk the other way in one run -- never further
    # apart than both runs' ten steps together (a step is at most ~1.5 lr)
    lr, steps = 1e-4, len(seq)
    for k in ('param_predictor/W', 'feature_extractor/conv1/W', 'feature_extractor/res5/1/conv2/W', 'feature_extractor/res3/0/conv1/W'):
        d = np.abs(p1[k] - p0[k])
        assert d.max() <= 3 * steps * lr, (k, float(d.max()))
