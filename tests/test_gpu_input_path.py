"""-m gpu: the training loop fed from FILES through the GPU input path (reference train_sheep_localizer.py:84-116:
``ImageDataset(use_imgaug=..., transform_probability=0.5)`` behind ``MultithreadIterator``): frames decoded on host threads,
augmentation + LANCZOS resize + ``/ 255`` on the GPU on the feed's own stream, the step starting with its batch in HBM.

The resize is pinned by Pillow itself (tests/test_gpu_resample.py); the augmentation branch restates imgaug, which is not
installable: PARITY UNPINNED for that branch (its GPU form is checked against its NumPy twin and the documented semantics,
tests/test_augment_cpu.py)."""
import os
import random

import numpy as np
import pytest
import torch

from tests._input_files import write_files

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_imgaug,n_processes", [(True, 0), (False, 0), (True, 3)])
def test_device_feed_equals_stack_of_get_example(tmp_path, use_imgaug, n_processes):
    """what the loop receives from MultithreadIterator(device=...) is, bit for bit, ``stack(get_example(i))`` of a dataset with
    the same seed -- across an epoch boundary, mixed frame sizes, both augmentation branches, pool and prefetch on"""
    from loans_amd.common.datasets.image_dataset import ImageDataset
    from loans_amd.runtime import training
    train, _, _ = write_files(tmp_path)
    mk = lambda: ImageDataset(train, str(tmp_path), image_size=(64, 64), dtype=np.float32, use_imgaug=use_imgaug,   # noqa: E731
                              transform_probability=0.5, augment_seed=21)
    host_ds, dev_ds = mk(), mk()
    random.seed(8)
    order = training.SerialIterator(host_ds, 4, shuffle=True, seed=2)
    want = [np.stack(next(order)) for _ in range(5)]
    random.seed(8)
    feed = training.MultithreadIterator(dev_ds, 4, shuffle=True, seed=2, n_threads=4, n_prefetch=2, device=0,
                                       n_processes=n_processes)       # 3: frames decoded by worker processes (decode_farm.py)
    changed = 0
    plain = ImageDataset(train, str(tmp_path), image_size=(64, 64))
    for k, ref in enumerate(want):
        got = next(feed)
        assert torch.is_tensor(got) and got.is_cuda and got.dtype == torch.float32 and tuple(got.shape) == (4, 3, 64, 64)
        assert training.concat_examples(got, 0) is got                   # the converter passes a finished batch through
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
    feed.finalize()
    # ... and the augmentation is really on: about half of the examples differ from the un-augmented frames
    random.seed(8)
    idx = training.SerialIterator(plain, 4, shuffle=True, seed=2)
    for ref in want:
        changed += int((np.stack(next(idx)) != ref).any(axis=(1, 2, 3)).sum())
    assert 3 <= changed <= 17, changed



@pytest.mark.parametrize("use_imgaug,where", [(True, 'device'), (True, 'host'), (False, 'device')])
def test_decode_once_cache_keeps_the_bytes(tmp_path, use_imgaug, where):
    """frame_cache.py (round 5; not in the reference, whose ImageDataset decodes every frame every epoch): with the decoded frames
    kept in HBM (a batch = a device-side gather) or in host memory, the loop still receives ``stack(get_example(i))`` bit for bit
    -- over THREE epochs of a seeded run (the first fills the cache, the others are served from it), mixed frame sizes, both
    augmentation branches (the naive crop / flip branch caches on the host whatever was asked: it resizes strided views)."""
    from loans_amd.common.datasets.image_dataset import ImageDataset
    from loans_amd.runtime import training
    train, _, _ = write_files(tmp_path)
    mk = lambda **kw: ImageDataset(train, str(tmp_path), image_size=(64, 64), dtype=np.float32, use_imgaug=use_imgaug,   # noqa: E731
                                   transform_probability=0.5, augment_seed=21, **kw)
    host_ds, dev_ds = mk(), mk(frame_cache_gb=1, frame_cache_where=where)
    assert dev_ds._cache.where == ('host' if not use_imgaug else where)
    n_batches = 3 * ((len(dev_ds) + 3) // 4) + 1
    random.seed(8)
    order = training.SerialIterator(host_ds, 4, shuffle=True, seed=2)
    want = [np.stack(next(order)) for _ in range(n_batches)]
    random.seed(8)
    feed = training.MultithreadIterator(dev_ds, 4, shuffle=True, seed=2, n_threads=4, n_prefetch=2, device=0)
    for ref in want:
        np.testing.assert_array_equal(next(feed).cpu().numpy(), ref)
    feed.finalize()
    c = dev_ds._cache
    assert len(c) == len(dev_ds) and c.hits >= len(dev_ds), (len(c), c.hits, c.misses)        # every frame decoded once, then served
    assert c.misses <= len(dev_ds) + 8, c.misses                                              # (+ the batches in flight at the epoch's end)


def test_trainer_runs_from_files_with_augmentation(tmp_path, monkeypatch):
    """the reference's command line on generator-written files: 4 iterations, validation at the log interval, snapshots in the
    timestamped log directory with a JSON log whose first entry carries the configuration"""
    import json
    import train_sheep_localizer as T
    from loans_amd.common.datasets.image_dataset import ImageDataset
    train, val, ref = write_files(tmp_path / 'data')
    calls = {'finish': 0, 'host': 0}
    fin, ex = ImageDataset.finish_batch, ImageDataset.get_example
    monkeypatch.setattr(ImageDataset, 'finish_batch', lambda self, d, dev, m=map: (calls.__setitem__('finish', calls['finish'] + 1), fin(self, d, dev, m))[1])
    monkeypatch.setattr(ImageDataset, 'get_example', lambda self, i: (calls.__setitem__('host', calls['host'] + 1), ex(self, i))[1])
    args = T.parse_args([train, val, ref, '--use-resnet-18', '-b', '4', '--image-size', '64', '64', '--target-size', '16', '16',
                         '--iterations', '4', '--seed', '3', '--log-interval', '2', '--lr', '1e-4', '-g', '0',
                         '--no-snapshot-every-epoch', '--ln', 'files', '-l', str(tmp_path / 'logs')])
    lines = []
    history, localizer, _ = T.run(args, log=lines.append)
    assert calls['finish'] >= 4 and calls['host'] <= 1            # batches came through the GPU stages (one host frame for predict())
    assert [h['iteration'] for h in history] == [2, 3, 4] and all(      # log interval 2; 10 frames / 4: iteration 3 ends an epoch
        np.isfinite(h['loss_localizer']) and np.isfinite(h['loss_dis']) for h in history)
    assert all(set(h['validation']) >= {'mean_iou', 'map'} for h in history)
    # reference :158-162: <log-dir>/<iso time>_<log name>
    assert os.path.dirname(args.log_dir) == str(tmp_path / 'logs') and args.log_dir.endswith('_files')
    assert os.path.exists(os.path.join(args.log_dir, 'SheepLocalizer_4.npz'))
    log = json.load(open(os.path.join(args.log_dir, 'log')))
    assert [e['iteration'] for e in log] == [2, 3, 4] and log[0]['localizer'] == ['SheepLocalizer', 'localizer.py']
    assert log[0]['image_size'] == [64, 64] and log[0]['use_imgaug'] is True and 'mean_iou' in log[1] and 'loss_dis' in log[1]


def test_history_is_only_synchronised_at_log_intervals(tmp_path):
    """ADVICE (round 2): no per-iteration host sync, no unbounded history; theta entries are those of the TRAINING batch even
    when a validation pass ran in between (its test-mode forward rebinds last_transform_params)"""
    import train_sheep_localizer as T
    base = ['--use-resnet-18', '-b', '4', '--image-size', '64', '64', '--target-size', '16', '16', '--iterations', '6',
            '--dataset-size', '64', '--seed', '5', '--no-shuffle', '--log-interval', '3', '--validation-size', '6', '--lr', '1e-5',
            '--no-snapshot-every-epoch', '--flat-log-dir']
    h0, _, _ = T.run(T.parse_args(base + ['-l', str(tmp_path / 'a')]), log=lambda s: None)
    assert [h['iteration'] for h in h0] == [3, 6] and 'theta' not in h0[0]
    h1, _, _ = T.run(T.parse_args(base + ['--record-history', '-l', str(tmp_path / 'b')]), log=lambda s: None)
    h2, _, _ = T.run(T.parse_args(base + ['--record-history', '--no-validation', '-l', str(tmp_path / 'c')]), log=lambda s: None)
    assert [h['iteration'] for h in h1] == list(range(1, 7))
    for a, b in zip(h1, h2):                 # validation does not leak into the recorded training values
        # (to the last bits only: weight gradients are summed by fp32 atomics, whose order varies from run to run)
        np.testing.assert_allclose(a['theta'], b['theta'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(a['loss_localizer'], b['loss_localizer'], rtol=1e-5)
