"""Minimal training-loop runtime with the surface the reference's trainer script uses
(chainer.training.StandardUpdater, chainer.dataset.concat_examples, iterators):
``update()``/``update_core()``, ``get_optimizer``, ``get_iterator``, ``iteration``,
``epoch``, ``is_new_epoch`` (train_sheep_localizer.py:113-164)."""
import numpy as np
import torch


def concat_examples(batch, device=None):
    """Stack a list of examples (arrays or tuples of arrays) and move them to ``device``.  A batch that an iterator already
    finished on the device (``MultithreadIterator(device=...)``: a tensor, or a tuple of tensors) passes through."""
    if torch.is_tensor(batch) or (isinstance(batch, tuple) and all(torch.is_tensor(b) for b in batch)):
        return batch
    first = batch[0]
    if isinstance(first, tuple):
        cols = [np.stack([np.asarray(ex[i]) for ex in batch]) for i in range(len(first))]
        return tuple(_to_device(c, device) for c in cols)
    return _to_device(np.stack([np.asarray(ex) for ex in batch]), device)


def _to_device(array, device):
    if torch.is_tensor(array):
        return array if device is None else array.to(torch.device('cuda', device) if isinstance(device, int) else device)
    t = torch.from_numpy(np.ascontiguousarray(array))
    if device is None or (isinstance(device, int) and device < 0):
        return t
    return t.to(torch.device('cuda', device) if isinstance(device, int) else device, non_blocking=True)


class SerialIterator:
    """Repeating, optionally shuffling batch iterator over an indexable dataset."""

    def __init__(self, dataset, batch_size, repeat=True, shuffle=True, seed=0):
        self.dataset, self.batch_size, self.repeat, self.shuffle = dataset, batch_size, repeat, shuffle
        self._rng = np.random.RandomState(seed)
        self.epoch = 0
        self.is_new_epoch = False
        self._order = self._new_order()
        self._pos = 0

    def _new_order(self):
        n = len(self.dataset)
        return self._rng.permutation(n) if self.shuffle else np.arange(n)

    def reset(self):
        """Back to the start of the first epoch (chainer iterators' ``reset``; the Evaluator runs a fresh pass each time)."""
        self.epoch, self.is_new_epoch, self._pos = 0, False, 0
        self._order = self._new_order()

    def __iter__(self):
        return self

    def __next__(self):
        """chainer.iterators.SerialIterator: ``is_new_epoch`` is True for the batch that CONSUMES the last example of an epoch
        (the reference's log / validation trigger reads it, train_sheep_localizer.py:188-190); with ``repeat`` a batch that
        straddles the boundary is completed from the next epoch's order, without it the last batch is the short remainder."""
        n = len(self.dataset)
        if not self.repeat and self._pos >= n:
            raise StopIteration
        i_end = self._pos + self.batch_size
        idx = list(self._order[self._pos:i_end])
        if i_end >= n:
            if self.repeat:
                rest = i_end - n
                self._order = self._new_order()
                idx += list(self._order[:rest])
                self._pos = rest
            else:
                self._pos = n
            self.epoch += 1
            self.is_new_epoch = True
        else:
            self.is_new_epoch = False
            self._pos = i_end
        return [self.dataset[int(i)] for i in idx]

    next = __next__

    @property
    def epoch_detail(self):
        return self.epoch + self._pos / max(len(self.dataset), 1)


class MultithreadIterator(SerialIterator):
    """``chainer.iterators.MultithreadIterator(dataset, batch_size, repeat, shuffle, n_threads)`` in the role the reference
    gives it (train_sheep_localizer.py:113-116): the examples of the NEXT batches are decoded by a pool of host threads while
    the current step runs.  Same order / epoch bookkeeping as ``SerialIterator`` (the values a batch is returned with are
    those of that batch, not of the prefetched ones).

    MI355X side: with ``device=`` and a dataset that offers ``decode_batch / finish_batch`` (``ImageDataset``) the producer
    thread also uploads the uint8 frames and runs augmentation + LANCZOS resize + ``/ 255`` on the GPU on a stream of its own
    (``loans_amd/common/datasets/resample.py``), so ``next()`` hands out a float32 NCHW batch that is already resident in HBM
    and the step's stream only waits for an event; other datasets are decoded by the pool and uploaded by the converter.
    ``device_stage='consumer'`` keeps every GPU call on the caller's thread (needed while a hipGraph is being captured: an
    allocation from another thread would invalidate the capture).

    Random draws of the datasets (augmentation) are taken on the producer thread in index order, each ``ImageDataset`` from a
    stream of its own (``augment_seed`` / ``reseed``; both augmentation branches), so a run is reproducible for a given seed
    whatever the pool size and however the producer threads of several iterators interleave.  Datasets without
    ``get_examples`` (``LabeledImageDataset``: no random draws in ``get_example``) go through ``pool.map(ds.__getitem__)``."""

    def __init__(self, dataset, batch_size, repeat=True, shuffle=True, n_threads=4, seed=0, device=None, n_prefetch=2,
                 device_stage='producer', n_processes=0):
        super().__init__(dataset, batch_size, repeat=repeat, shuffle=shuffle, seed=seed)
        # n_processes > 0: frames are decoded by that many worker PROCESSES (common/datasets/decode_farm.py; Pillow's decoders
        # hold the GIL, a thread pool stays near one core's rate); the pool threads then only move pixels over pipes
        self.n_processes, self._farm = int(n_processes), None
        if self.n_processes > 0:
            n_threads = max(n_threads, self.n_processes)
        self.n_threads, self.n_prefetch, self.device_stage = max(1, int(n_threads)), max(1, int(n_prefetch)), device_stage
        self.device = None if device is None or (isinstance(device, int) and device < 0) else \
            (torch.device('cuda', device) if isinstance(device, int) else torch.device(device))
        self._pool = self._pool2 = self._thread = self._thread2 = self._queue = self._mid = self._stream = None
        self._generation = 0
        self._host_state = (self.epoch, self.is_new_epoch, self._pos)

    # ---- producer side ----------------------------------------------------------------------------------------------
    def _indices(self):
        """the next batch's indices + the bookkeeping values that batch is handed out with (SerialIterator.__next__)"""
        n = len(self.dataset)
        if not self.repeat and self._pos >= n:
            return None
        i_end = self._pos + self.batch_size
        idx = list(self._order[self._pos:i_end])
        if i_end >= n:
            if self.repeat:
                rest = i_end - n
                self._order = self._new_order()
                idx += list(self._order[:rest])
                self._pos = rest
            else:
                self._pos = n
            epoch, new = self._p_epoch + 1, True
        else:
            epoch, new = self._p_epoch, False
            self._pos = i_end
        self._p_epoch = epoch
        return [int(i) for i in idx], (epoch, new, self._pos)

    def _decode(self, idx):
        """host half of a batch (runs on the decode thread): pooled / farmed decode, random draws in index order"""
        ds = self.dataset
        if self.device is not None and hasattr(ds, 'decode_batch'):
            if self._farm is not None:
                return ('decoded', ds.decode_batch(idx, self._pool.map, self._farm))
            return ('decoded', ds.decode_batch(idx, self._pool.map))
        if hasattr(ds, 'get_examples'):        # decode / resize pooled, random draws in index order
            return ('host', ds.get_examples(idx, self._pool.map))
        return ('host', list(self._pool.map(ds.__getitem__, idx)))

    def _finish(self, item):
        """device half (runs on the finish thread, so that batch k is uploaded and resampled while batch k + 1 is decoded)"""
        if item[0] != 'decoded' or self.device_stage != 'producer':
            return item
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
            batch = self.dataset.finish_batch(item[1], self.device, self._pool2.map)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return ('device', batch, ev)

    def _put(self, generation, q, item):
        """blocking put that gives up when the iterator has been reset / finalised meanwhile"""
        import queue
        while generation == self._generation:
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _produce(self, generation, q, q_mid):
        try:
            while generation == self._generation:
                nxt = self._indices()
                if nxt is None:
                    self._put(generation, q_mid, (generation, StopIteration, None))
                    return
                idx, state = nxt
                if not self._put(generation, q_mid, (generation, self._decode(idx), state)):
                    return
        except BaseException as e:          # handed to the consumer: a failing decode must fail the training loop
            self._put(generation, q_mid, (generation, e, None))

    def _produce_finish(self, generation, q, q_mid):
        import queue
        while generation == self._generation:
            try:
                g, item, state = q_mid.get(timeout=0.1)
            except queue.Empty:
                continue
            if item is StopIteration or isinstance(item, BaseException):
                self._put(generation, q, (g, item, state))
                return
            try:
                out = (g, self._finish(item), state)
            except BaseException as e:
                out = (g, e, None)
            if not self._put(generation, q, out) or out[2] is None:
                return

    def _start(self):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=self.n_threads, thread_name_prefix='loans-decode')
        if self.device is not None and self._stream is None and self.device_stage == 'producer':
            self._stream = torch.cuda.Stream(device=self.device)
        if self.n_processes > 0 and self._farm is None and self.device is not None and hasattr(self.dataset, 'decode_batch'):
            from ..common.datasets.decode_farm import DecodeFarm
            self._farm = DecodeFarm(self.n_processes)
        if self._pool2 is None:
            self._pool2 = ThreadPoolExecutor(max_workers=min(self.n_threads, 8), thread_name_prefix='loans-stage')
        self._p_epoch = self.epoch
        self._queue, self._mid = queue.Queue(maxsize=self.n_prefetch), queue.Queue(maxsize=1)
        args = (self._generation, self._queue, self._mid)
        self._thread = threading.Thread(target=self._produce, args=args, daemon=True, name='loans-feed-decode')
        self._thread2 = threading.Thread(target=self._produce_finish, args=args, daemon=True, name='loans-feed-finish')
        self._thread.start()
        self._thread2.start()

    def _stop(self):
        self._generation += 1              # both threads poll it (their queue operations time out every 0.1 s)
        threads = [t for t in (self._thread, self._thread2) if t is not None]
        self._queue = self._mid = self._thread = self._thread2 = None
        for t in threads:
            t.join()
        if self.device is not None and threads:         # the finish thread's pinned staging buffers go with it
            from ..common.datasets.resample import release_staging
            for t in threads:
                release_staging(t.ident)

    # ---- consumer side ----------------------------------------------------------------------------------------------
    def reset(self):
        self._stop()
        super().reset()

    def finalize(self):
        self._stop()
        for name in ('_pool', '_pool2'):
            if getattr(self, name) is not None:
                getattr(self, name).shutdown(wait=True)
                setattr(self, name, None)
        if self._farm is not None:
            self._farm.close()
            self._farm = None

    def __next__(self):
        if self._thread is None:
            self._start()
        generation, item, state = self._queue.get()
        assert generation == self._generation
        if item is StopIteration:
            self._queue.put((generation, StopIteration, None))       # keeps raising until reset()
            raise StopIteration
        if isinstance(item, BaseException):
            self._stop()
            raise item
        self.epoch, self.is_new_epoch, self._host_pos = state[0], state[1], state[2]
        kind = item[0]
        if kind == 'host':
            return item[1]
        if kind == 'decoded':
            return self.dataset.finish_batch(item[1], self.device)
        _, batch, ev = item
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        batch.record_stream(cur)
        return batch

    next = __next__

    @property
    def epoch_detail(self):
        return self.epoch + getattr(self, '_host_pos', 0) / max(len(self.dataset), 1)


class DeviceBatchIterator:
    """Yields pre-staged device batches (already resident in HBM) round-robin; used by
    bench.py so that the timed region starts with inputs in device memory."""

    def __init__(self, batches):
        self.batches, self.i = list(batches), 0
        self.epoch, self.is_new_epoch = 0, False

    def __iter__(self):
        return self

    def __next__(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        self.is_new_epoch = (self.i % len(self.batches)) == 0
        self.epoch += int(self.is_new_epoch)
        return b

    next = __next__


def identity_converter(batch, device=None):
    return batch


class Evaluator:
    """``chainer.training.extensions.Evaluator(iterator, target, device=..., eval_func=...)`` as the reference uses it
    (train_sheep_localizer.py:192-197): one pass over a ``repeat=False`` iterator, ``eval_func(*converter(batch, device))`` per
    batch, every value the function reports (or returns) averaged over the batches (Chainer's ``DictSummary.compute_mean``)
    and reported once.  Returns the averaged dict."""

    def __init__(self, iterator, target, converter=concat_examples, device=None, eval_func=None):
        self.iterator, self.target, self.converter, self.device = iterator, target, converter, device
        self.eval_func = eval_func or target

    def evaluate(self):
        from .core import reporter
        it = self.iterator
        it.reset()
        sums, n = {}, 0
        for batch in it:
            in_arrays = self.converter(batch, self.device)
            before = dict(reporter.observation)
            out = self.eval_func(*in_arrays) if isinstance(in_arrays, tuple) else self.eval_func(in_arrays)
            seen = {k: v for k, v in reporter.observation.items() if before.get(k) is not v}
            if isinstance(out, dict):
                seen.update(out)
            for k, v in seen.items():
                sums[k] = sums.get(k, 0.0) + float(v)
            n += 1
        return {k: v / max(n, 1) for k, v in sums.items()}

    def __call__(self, trainer=None):
        from .core import report
        result = self.evaluate()
        report(result)
        return result


class StandardUpdater:

    def __init__(self, iterator, optimizer, converter=concat_examples, device=None, comm=None):
        if not isinstance(iterator, dict):
            iterator = {'main': iterator}
        if not isinstance(optimizer, dict):
            optimizer = {'main': optimizer}
        self._iterators, self._optimizers = iterator, optimizer
        self.converter = converter
        if device is None or (isinstance(device, int) and device < 0):
            device = torch.cuda.current_device()
        self.device = device
        self.comm = comm
        self.iteration = 0

    @property
    def epoch(self):
        return self._iterators['main'].epoch

    @property
    def epoch_detail(self):
        return getattr(self._iterators['main'], 'epoch_detail', float(self.epoch))

    @property
    def is_new_epoch(self):
        return self._iterators['main'].is_new_epoch

    def get_optimizer(self, name):
        return self._optimizers[name]

    def get_all_optimizers(self):
        return dict(self._optimizers)

    def get_iterator(self, name):
        return self._iterators[name]

    def update(self):
        self.update_core()
        self.iteration += 1

    def update_core(self):
        raise NotImplementedError
