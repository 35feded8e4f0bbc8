"""Data-parallel training over the GPUs of one node: one process per GPU,
``torch.distributed`` with the ``nccl`` backend (= RCCL over xGMI on ROCm).

The reference's LoANs trainer is single-GPU (train_sheep_localizer.py:59); its only
data-parallel call site is Chainer's ``MultiprocessParallelUpdater`` in the SSD
sub-project (schaaaafrichter/train.py:159-191: reduce-to-root + update + broadcast, local
BN statistics).  MI355X-first design instead: every rank holds the same parameters, all
gradients of a model live in ONE flat arena, and each optimiser step is
  all-reduce(sum) of that arena in a few large buckets  ->  identical fused Adam on every rank
(no parameter broadcast after the initial one).  xGMI is a point-to-point mesh, so few large
collectives beat many small ones; the 1/world_size factor is folded into the Adam kernel.
BN statistics stay local to each shard, like the reference's DP.
"""
import os

import torch
import torch.distributed as dist

BUCKET_FLOATS = 16 * 1024 * 1024      # 64 MiB buckets: the localizer's active gradients are one bucket


# LOANS_DIST_SELFTEST=1 issues every collective even at world size 1 (exercises the RCCL path on a 1-GPU box)
_SELFTEST = os.environ.get('LOANS_DIST_SELFTEST', '0') == '1'


class Communicator:
    def __init__(self, group=None):
        self.group = group
        self.size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.active = dist.is_initialized() and (self.size > 1 or _SELFTEST)
        # what actually carries the collectives: 'nccl' (= RCCL over xGMI on ROCm), 'gloo' (tests), None (single process)
        self.backend = dist.get_backend(group) if dist.is_initialized() else None

    def bcast_data(self, link):
        """Rank 0's parameters and persistents to everyone (once, after construction)."""
        if not self.active:
            return
        arena = link.arena or link.finalize()
        dist.broadcast(arena.data, src=0, group=self.group)
        for _, l, n in link.namedpersistents():
            v = getattr(l, n)
            if torch.is_tensor(v):
                dist.broadcast(v, src=0, group=self.group)

    def allreduce_grad(self, arena, async_op=False):
        """Sum the gradient arena over all ranks, in place, in large buckets.  ``async_op``: return the work handles
        instead of making the current stream wait (the collectives then overlap what is launched next)."""
        return self.allreduce_range(arena, 0, getattr(arena, 'active_numel', arena.numel), async_op)

    def allreduce_range(self, arena, lo, hi, async_op=False):
        """The same for the floats [lo, hi) of the gradient arena (one stage's share, see ``exchange_plan``)."""
        if not self.active or hi <= lo:
            return []
        g = arena.grad
        works = []
        for a in range(lo, hi, BUCKET_FLOATS):
            w = dist.all_reduce(g[a:min(a + BUCKET_FLOATS, hi)], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                works.append(w)
        return works

    def allreduce_max(self, value):
        t = torch.tensor([value], dtype=torch.float64, device='cuda' if torch.cuda.is_available() else 'cpu')
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.active:
            dist.barrier(group=self.group)


def init_from_env(backend=None):
    """Join the process group torchrun described (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1 and not (_SELFTEST and 'RANK' in os.environ):
        return Communicator()
    local_rank = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
    if backend is None:
        # LOANS_DIST_BACKEND=gloo lets several ranks share one GPU (gloo moves CUDA tensors through the host): the
        # whole multi-rank flow can then be exercised on a 1-GPU box; RCCL wants one device per rank
        backend = os.environ.get('LOANS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if not dist.is_initialized():
        # no `device_id=`: binding the communicator eagerly mutes torch's barrier() warning (the device is already
        # current, set above) but measured 6 % slower steps at world size 1 with the collectives on (4368 vs 4644
        # images/s): the lazily created communicator's streams overlap the backward better
        dist.init_process_group(backend=backend)
    return Communicator()


# the gradient arena is exchanged in parts during the backward (False: one exchange of the whole arena behind it, rounds 1-2)
STAGED_EXCHANGE = True


def exchange_plan(link):
    """Where the gradient arena of ``link`` is cut for the staged exchange: ``{stage name: first float of that stage}`` for the
    stages a backbone names in ``exchange_stages`` (sheep/resnet.py: res4, res5).  The arena is laid out in forward order
    (the head and the cold res6 / res7 behind the backbone), so when the backward has left stage S every gradient from S's
    first float to the end of the active prefix is complete: [res5 .. end) goes first -- 9.4 M of the 12.6 M floats at 224 px
    --, [res4, res5) second, [0, res4) when the backward ends."""
    arena = link.arena or link.finalize()
    offset = {id(p): o for p, o in zip(arena.params, arena.offsets)}
    plan = {}
    for path, l in link.namedlinks():
        for name in getattr(l, 'exchange_stages', ()):
            prefix = (path if path != '/' else '') + '/' + name + '/'
            inside = [offset[id(p)] for k, p in link.namedparams() if k.startswith(prefix)]
            if inside:
                plan[name] = min(inside)
    return plan


def create_multi_node_optimizer(optimizer, comm):
    """Attach a communicator to an optimiser (ChainerMN's name for the same thing).  With an active communicator the backbone's
    stage boundaries report to the optimiser (``Adam.stage_done``), which exchanges each part of the gradient arena as soon as
    it is complete, beside the backward of the stages in front of it."""
    optimizer.comm = comm
    if STAGED_EXCHANGE and comm is not None and comm.active and optimizer.target is not None:
        for l in optimizer.target.links():
            if getattr(l, 'exchange_stages', None):
                l.__dict__['_stage_hook'] = optimizer.stage_done
    return optimizer


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
