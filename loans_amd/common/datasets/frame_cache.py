"""Decode once (round 5; SURVEY 8f.2: "so 8 GPUs are not starved by 8 host cores").

An 8-GPU node gives each rank 8 host cores; Pillow decodes 650 frames/s per core (480 x 640 JPEG), the step consumes 5 100
(fp32) .. 6 700 (bf16) frames/s: a node is fed at 0.76 of its consumption (profiles/r4_feed_bench.txt).  The frames of the
reference's ``ImageDataset`` (common/datasets/image_dataset.py:47-98) do not change between epochs -- only the random draws
of the augmentation do -- so a frame is decoded the first time an epoch asks for it and kept as the uint8 HWC array the
resize would see:

* ``where='device'`` (default): in HBM, fixed-size ``[chunk][H][W][3]`` uint8 pools per source size (288 GB hold ~300 k frames
  of 480 x 640); what is ALLOCATED counts against the budget -- a pool is added only if it fits what is left -- so the resident
  bytes never exceed ``budget_bytes`` and nothing is ever copied to grow (the doubling tensor held up to 2-3 x it).  A batch whose frames are resident is a device-side gather in front of the augmentation / LANCZOS kernels:
  no decode, no pinned staging copy, no PCIe transfer.
* ``where='host'``: the decoded arrays in host memory; the batch is staged and uploaded as before, only the decode is skipped.
  (The naive crop / flip branch -- ``use_imgaug=False`` -- hands strided VIEWS of a frame to the resize and always uses this form.)

The bytes a batch is made of are the decoded arrays either way: ``device_batch == stack(get_example)`` holds as before.
"""
import threading

import numpy as np
import torch


class CachedFrame:
    """placeholder of a device-resident frame in ``decode_batch``'s result"""
    __slots__ = ('index', 'shape')

    def __init__(self, index, shape):
        self.index, self.shape = index, tuple(shape)


class FrameCache:
    def __init__(self, budget_bytes, where='device'):
        assert where in ('device', 'host')
        self.where, self.budget, self.bytes = where, int(budget_bytes), 0
        self.host = {}              # index -> uint8 HWC array                         (where == 'host')
        self.slots = {}             # index -> ((H, W), slot)                            (where == 'device')
        self.pools = {}             # (H, W) -> {'chunks': [uint8 tensor [cap][H][W][3], ...], 'cap': frames per chunk, 'n': frames stored}
        self.allocated = 0          # bytes of all pools (device mode): what the budget bounds
        self.lock = threading.Lock()        # the decode thread asks, the finish thread stores
        self.hits = self.misses = 0

    def __len__(self):
        return len(self.host) + len(self.slots)

    # ---- decode thread ----------------------------------------------------------------------------------------------------
    def lookup(self, i):
        """the cached form of frame i: a uint8 array (host), a CachedFrame (device) or None"""
        with self.lock:
            if self.where == 'host':
                f = self.host.get(i)
            else:
                s = self.slots.get(i)
                f = CachedFrame(i, s[0] + (3,)) if s is not None else None
            if f is None:
                self.misses += 1
            else:
                self.hits += 1
            return f

    def keep_host(self, i, image):
        """host mode: keep a freshly decoded frame (within the budget)"""
        if self.where != 'host' or not isinstance(image, np.ndarray):
            return
        with self.lock:
            if i not in self.host and self.bytes + image.nbytes <= self.budget:
                self.host[i] = image
                self.bytes += image.nbytes

    # ---- finish thread (device mode) ------------------------------------------------------------------------------------------
    POOL_BYTES = 64 << 20         # a pool chunk: about 64 MiB of frames (at least 8 frames)

    def _store(self, size, indices, frames_dev):
        """append freshly uploaded frames [m][H][W][3] of one size to that size's pools (as many as the budget admits)"""
        H, W = size
        per = H * W * 3
        pool = self.pools.get(size)
        if pool is None:
            pool = self.pools[size] = {'chunks': [], 'cap': max(8, self.POOL_BYTES // per), 'n': 0}
        cap, stored = pool['cap'], 0
        while stored < len(indices):
            room = len(pool['chunks']) * cap - pool['n']
            if room == 0:
                if self.allocated + cap * per > self.budget:
                    break                                   # the next chunk does not fit the budget: the rest stays uncached
                pool['chunks'].append(torch.empty((cap, H, W, 3), device=frames_dev.device, dtype=torch.uint8))
                self.allocated += cap * per
                room = cap
            m = min(room, len(indices) - stored)
            off = pool['n'] % cap
            pool['chunks'][-1][off:off + m].copy_(frames_dev[stored:stored + m])
            with self.lock:
                for k in range(m):
                    self.slots[indices[stored + k]] = (size, pool['n'] + k)
                self.bytes += m * per
            pool['n'] += m
            stored += m

    def _gather(self, size, slots, device):
        """the frames at `slots` (global slot numbers of one size) as one [len][H][W][3] tensor, in that order"""
        pool = self.pools[size]
        cap = pool['cap']
        if len(pool['chunks']) == 1:
            return pool['chunks'][0].index_select(0, torch.tensor(slots, dtype=torch.int64).to(device, non_blocking=True))
        out = torch.empty((len(slots),) + tuple(pool['chunks'][0].shape[1:]), device=device, dtype=torch.uint8)
        by_chunk = {}
        for pos, sl in enumerate(slots):
            by_chunk.setdefault(sl // cap, []).append(pos)
        for c, positions in by_chunk.items():
            inside = torch.tensor([slots[p] % cap for p in positions], dtype=torch.int64).to(device, non_blocking=True)
            out[torch.tensor(positions, dtype=torch.int64).to(device, non_blocking=True)] = pool['chunks'][c].index_select(0, inside)
        return out

    def assemble(self, frames, rows, indices, out_hw, device, map_fn=map):
        """``frames_to_device`` for a batch whose entries are decoded arrays (misses: uploaded and stored) or CachedFrame
        placeholders (gathered from HBM): [N][3][oh][ow] float32 in batch order"""
        from .augment import apply_device
        from .resample import resize_lanczos, upload_frames
        device = torch.device(device)
        groups = {}
        for pos, f in enumerate(frames):
            shape = f.shape
            if len(shape) != 3 or shape[2] != 3 or (isinstance(f, np.ndarray) and f.dtype != np.uint8):
                raise ValueError('frames must be uint8 HWC RGB arrays')
            groups.setdefault(tuple(shape[:2]), []).append(pos)
        out = None
        for size, positions in groups.items():
            hit = [p for p in positions if isinstance(frames[p], CachedFrame)]
            miss = [p for p in positions if not isinstance(frames[p], CachedFrame)]
            up = None
            if miss:
                up = upload_frames([frames[p] for p in miss], device, map_fn)           # [m][H][W][3] uint8
                new = [p for p in miss if indices[p] not in self.slots]
                if new and len(new) == len(miss):
                    self._store(size, [indices[p] for p in miss], up)
            if not hit:
                g, order = up, miss
            else:
                with self.lock:
                    sl = [self.slots[frames[p].index][1] for p in hit]
                cached = self._gather(size, sl, device)
                g, order = (cached, hit) if up is None else (torch.cat([cached, up], dim=0), hit + miss)
            if rows is not None:            # the imgaug branch (augment.py), per image, before the resize (reference :80-93)
                g = apply_device(g, [rows[p] for p in order])
            batch = resize_lanczos(g, out_hw)
            if len(groups) == 1 and order == list(range(len(frames))):
                return batch
            if out is None:
                out = torch.empty((len(frames), 3, int(out_hw[0]), int(out_hw[1])), device=device, dtype=torch.float32)
            out[torch.tensor(order, dtype=torch.int64).to(device, non_blocking=True)] = batch
        return out
