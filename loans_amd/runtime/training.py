"""Minimal training-loop runtime with the surface the reference's trainer script uses
(chainer.training.StandardUpdater, chainer.dataset.concat_examples, iterators):
``update()``/``update_core()``, ``get_optimizer``, ``get_iterator``, ``iteration``,
``epoch``, ``is_new_epoch`` (train_sheep_localizer.py:113-164)."""
import numpy as np
import torch


def concat_examples(batch, device=None):
    """Stack a list of examples (arrays or tuples of arrays) and move them to ``device``."""
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


MultithreadIterator = SerialIterator      # the decode work of the real datasets is out of scope


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
