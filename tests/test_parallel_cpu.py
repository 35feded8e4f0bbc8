"""World-size-2 gloo tests (CPU) of the data-parallel path: the communicator sums the flat
gradient arena in buckets, the optimiser folds 1/world_size into the Adam kernel's grad_scale, the
batch-SUM regulariser (OutOfImageLoss, common/utils.py:315) is pre-scaled by the world size so that
averaged gradients equal the global-batch gradient, and rank 0's parameters are broadcast."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeArena:
    def __init__(self, n, rank):
        self.numel = n
        self.device = torch.device('cpu')
        self.data = torch.full((n,), float(rank + 1))
        self.grad = torch.arange(n, dtype=torch.float32) * (rank + 1)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from loans_amd import parallel
    comm = parallel.init_from_env(backend='gloo')
    assert (comm.size, comm.rank) == (world, rank)

    # 1. bucketed all-reduce of the gradient arena (bucket smaller than the arena -> several collectives)
    parallel.BUCKET_FLOATS = 1000
    arena = _FakeArena(4097, rank)
    comm.allreduce_grad(arena)
    expect = torch.arange(4097, dtype=torch.float32) * sum(r + 1 for r in range(world))
    assert torch.equal(arena.grad, expect)

    # 2. gradient averaging == global-batch gradient for the CPU oracle's joint step, with the SUM-type
    #    regulariser scaled by the world size on every rank and BN statistics local to the shard
    from oracle import model as M
    from oracle import chainer_ops as C
    from loans_amd.datasets import synthetic
    crop = (16, 16)
    rng = np.random.RandomState(0)                       # identical initial weights on every rank
    lp = M.cast_params(M.init_localizer_params(rng, predictor_w_std=5e-2), np.float64)
    frames = synthetic.make_frames(10 + rank, 2, 64, 64).astype(np.float64)        # this rank's shard
    loc = M.Localizer(lp, crop, train=True, rng=np.random.RandomState(0))
    rois, points = loc.forward(frames)
    l_dir, g_dir = C.direction_loss(points, (64, 64))
    l_oob, g_oob = C.out_of_image_loss(points)
    grads = {}
    loc.backward(None, g_dir + world * g_oob, grads)      # OutOfImageLossCalculator.batch_sum_scale = world
    keys = sorted(grads)
    flat = torch.from_numpy(np.concatenate([grads[k].ravel() for k in keys]))
    fake = _FakeArena(flat.numel(), rank)
    fake.grad = flat.clone()
    comm.allreduce_grad(fake)
    averaged = (fake.grad / world).numpy()
    # every rank can recompute all shards locally: mean of the mean-type loss + SUM of the sum-type loss
    total = None
    for r in range(world):
        lpr = M.cast_params(M.init_localizer_params(np.random.RandomState(0), predictor_w_std=5e-2), np.float64)
        fr = synthetic.make_frames(10 + r, 2, 64, 64).astype(np.float64)
        lr = M.Localizer(lpr, crop, train=True, rng=np.random.RandomState(0))
        _, pr = lr.forward(fr)
        gd = C.direction_loss(pr, (64, 64))[1]
        go = C.out_of_image_loss(pr)[1]
        gr = {}
        lr.backward(None, gd / world + go, gr)           # d/dtheta [ mean_r(dir_r) + sum_r(oob_r) ]
        v = np.concatenate([gr[k].ravel() for k in keys])
        total = v if total is None else total + v
    np.testing.assert_allclose(averaged, total, rtol=1e-9, atol=1e-12)

    # 2b. the STAGED exchange (SURVEY 8e "bucket + overlap with backward"): the stage boundaries of the backbone report to the
    #     optimiser during the backward (res5 first, then res4), which starts the all-reduce of that part of the arena at once;
    #     update_begin() adds the rest.  Bit-identical to ONE exchange of the whole active prefix, and nothing beyond it moves.
    import loans_amd
    np.random.seed(3)
    loc = loans_amd.SheepLocalizer((16, 16))
    arena = loc.finalize(torch.device('cpu'))
    arena.set_active('res6')                                   # 224 px frames: res6 / res7 are outside the active prefix
    g = torch.Generator().manual_seed(100 + rank)
    arena.grad.copy_(torch.randn(arena.numel, generator=g))
    mine = arena.grad.clone()
    opt = parallel.create_multi_node_optimizer(loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(loc), comm)
    fe = loc.feature_extractor
    assert fe.__dict__['_stage_hook'] == opt.stage_done and fe.exchange_stages == ('res4', 'res5')
    plan = parallel.exchange_plan(loc)
    assert 0 < plan['res4'] < plan['res5'] < arena.active_numel < arena.numel
    parallel.BUCKET_FLOATS = 1 << 20                           # several collectives per part
    fe.__dict__['_stage_hook']('res5')                         # what StageBoundary.backward calls
    assert opt._exchanged_from == plan['res5'] and len(opt._pending) == -(-(arena.active_numel - plan['res5']) // (1 << 20))
    fe.__dict__['_stage_hook']('res4')
    assert opt._exchanged_from == plan['res4']
    opt.update_begin()
    assert opt._exchanged_from == 0
    for w in opt._pending:
        w.wait()
    staged = arena.grad.clone()
    arena.grad.copy_(mine)
    comm.allreduce_grad(arena)
    assert torch.equal(staged, arena.grad)
    assert torch.equal(staged[arena.active_numel:], mine[arena.active_numel:])
    other = torch.randn(arena.numel, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    assert torch.equal(staged[:arena.active_numel], (mine + other)[:arena.active_numel])
    # the share of the bytes that travels while the backward of res4 .. stem is still being issued
    assert (arena.active_numel - plan['res5']) / arena.active_numel > 0.7
    opt._pending, opt._exchanged_from = [], None

    # 3. max-reduce used by bench.py's timing, and the barrier
    assert comm.allreduce_max(float(rank)) == float(world - 1)
    comm.barrier()
    if rank == 0:
        out.put('ok')
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == 'ok'



def _worker_r50(rank, world, port, out):
    """four ranks, the ResNet-50 localizer with res6 / res7 active (512 px frames): ~75 M gradient floats, several 64 MiB buckets
    per part of the staged exchange"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import loans_amd
    from loans_amd import parallel
    comm = parallel.init_from_env(backend='gloo')
    assert (comm.size, comm.rank) == (world, rank)
    np.random.seed(5)
    loc = loans_amd.Resnet50SheepLocalizer((16, 16))
    arena = loc.finalize(torch.device('cpu'))
    arena.set_active(None)                                      # frames taller than 300 px: everything is active
    n = arena.active_numel
    assert n == arena.numel and n > 60_000_000
    gen = lambda r: torch.randn(n, generator=torch.Generator().manual_seed(500 + r))      # noqa: E731
    arena.grad.copy_(gen(rank))
    opt = parallel.create_multi_node_optimizer(loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(loc), comm)
    plan = parallel.exchange_plan(loc)
    calls, plain = [], comm.allreduce_range
    comm.allreduce_range = lambda a, lo, hi, async_op=False: (calls.append((lo, hi)), plain(a, lo, hi, async_op))[1]
    hook = loc.feature_extractor.__dict__['_stage_hook']
    hook('res5')
    hook('res4')
    opt.update_begin()                                          # the rest: [0, res4)
    for w in opt._pending:
        w.wait()
    assert calls == [(plan['res5'], n), (plan['res4'], plan['res5']), (0, plan['res4'])], (calls, plan)
    B = parallel.BUCKET_FLOATS
    buckets = [-(-(hi - lo) // B) for lo, hi in calls]
    assert buckets[0] >= 3 and sum(buckets) >= 5, buckets       # [res5 .. end) alone is several 64 MiB collectives
    total = gen(0)
    for r in range(1, world):
        total += gen(r)
    # every float of every bucket exactly once: the sum of the four ranks' gradients (fp32 sums in gloo's ring order) -- a bucket
    # summed twice or skipped is off by O(1)
    err = (arena.grad - total).abs().max().item()
    assert err <= 1e-5, err
    opt._pending, opt._exchanged_from = [], None
    if rank == 0:
        out.put(('ok', buckets))
    comm.barrier()
    dist.destroy_process_group()


def test_four_rank_gloo_staged_exchange_resnet50_512px():
    """exchange_plan on the ResNet-50 localizer at 512 px (res6 / res7 active), four ranks, 64 MiB buckets: the
    three parts cover the arena exactly once, every bucket carries the four-rank sum, the kernel's scale is 1 / 4."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_r50, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    tag, buckets = q.get(timeout=5)
    assert tag == 'ok' and sum(buckets) >= 5


def test_updater_scales_batch_sum_regulariser_by_world_size():
    import loans_amd
    from loans_amd.runtime import training

    class Comm:
        size, rank = 4, 1

    np.random.seed(0)
    loc, dis = loans_amd.SheepLocalizer((16, 16)), loans_amd.ResnetAssessor()
    upd = loans_amd.SheepAssessor(models=[loc, dis], iterator={'main': iter(()), 'real': iter(())},
                                  optimizer={'opt_gen': loans_amd.Adam(amsgrad=True).setup(loc),
                                             'opt_dis': loans_amd.Adam(amsgrad=True).setup(dis)},
                                  converter=training.identity_converter, device=0, comm=Comm())
    assert upd.regularizers[1].batch_sum_scale == 4.0
    assert isinstance(upd.regularizers[0], loans_amd.DirectionLossCalculator)
    upd1 = loans_amd.SheepAssessor(models=[loc, dis], iterator={'main': iter(()), 'real': iter(())},
                                   optimizer={'opt_gen': loans_amd.Adam(amsgrad=True).setup(loc),
                                              'opt_dis': loans_amd.Adam(amsgrad=True).setup(dis)},
                                   converter=training.identity_converter, device=0)
    assert upd1.regularizers[1].batch_sum_scale == 1.0


def test_adam_lr_schedule_matches_chainer():
    import loans_amd
    from oracle import chainer_ops as C
    opt = loans_amd.Adam(alpha=1e-3, amsgrad=True)
    for t in (1, 2, 10, 1000):
        opt.t = t
        assert abs(opt.lr - C.adam_lr(1e-3, .9, .999, t)) < 1e-18


def test_exchange_plan_of_both_localizers():
    """the cut points of the staged gradient exchange: res4 / res5 of either backbone, in arena (= forward) order, the head and
    the cold res6 / res7 behind res5; without an active communicator no hook is attached and the graph is untouched"""
    import loans_amd
    from loans_amd import parallel
    np.random.seed(0)
    for cls in (loans_amd.SheepLocalizer, loans_amd.Resnet50SheepLocalizer):
        loc = cls((16, 16))
        arena = loc.finalize(torch.device('cpu'))
        plan = parallel.exchange_plan(loc)
        assert sorted(plan) == ['res4', 'res5'] and 0 < plan['res4'] < plan['res5'] < arena.numel
        off = {k: o for (k, p), o in zip([(k, p) for k, p in loc.namedparams() if not k.startswith(('/res6/', '/res7/'))], arena.offsets)}
        for k, o in off.items():
            stage = k.split('/')[2] if k.startswith('/feature_extractor/') else 'head'
            if stage in ('res5', 'fc6', 'head'):
                assert o >= plan['res5'], k
            elif stage == 'res4':
                assert plan['res4'] <= o < plan['res5'], k
            else:
                assert o < plan['res4'], k
        assert min(arena.cold_offsets.values()) > plan['res5']
        opt = parallel.create_multi_node_optimizer(loans_amd.Adam(amsgrad=True).setup(loc), parallel.Communicator())
        assert '_stage_hook' not in loc.feature_extractor.__dict__ and opt._exchanged_from is None
