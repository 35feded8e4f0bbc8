"""Input contract of the hot path (common/datasets/image_dataset.py:47-182, paste_and_crop_sheep.py:221-228):
generator -> images.csv -> LabeledImageDataset round trip, uint8/255 exactness, bbox rescaling."""
import csv
import os

import numpy as np
from PIL import Image

from loans_amd.common.datasets.image_dataset import ImageDataset, LabeledImageDataset, resize_bbox
from loans_amd.datasets.sheep import paste_and_crop_sheep


def test_paste_and_crop_roundtrip(tmp_path):
    dest = str(tmp_path / 'samples')
    assert paste_and_crop_sheep.main(['-', dest, '--synthetic', '3', '--zoom-mode', '--seed', '1', '--num-samples', '8',
                                      '--output-size', '40', '30', '--image-size', '96', '96']) == 0
    rows = list(csv.reader(open(os.path.join(dest, 'images.csv')), delimiter='\t'))
    assert len(rows) >= 6 and all(len(r) == 2 and len(r[1].split('.')[1]) == 4 for r in rows)
    ds = LabeledImageDataset(os.path.join(dest, 'images.csv'), dest, image_size=(30, 40), dtype=np.float32, label_dtype=np.float32)
    img, label, dummy = ds.get_example(2)
    assert img.shape == (3, 30, 40) and img.dtype == np.float32 and dummy.shape == (1,)
    assert 0.0 <= img.min() and img.max() <= 1.0
    k = np.round(img * 255)
    np.testing.assert_array_equal(img, (k / 255).astype(np.float32))        # exactly uint8 / 255 (lossless prep)
    np.testing.assert_allclose(label, float(rows[2][1]), atol=1e-6)
    assert len(ds) == len(rows) and ds[0][0].shape == (3, 30, 40)


def test_image_dataset_resize_and_bbox_scaling(tmp_path):
    rng = np.random.RandomState(0)
    paths = []
    for i in range(3):
        p = str(tmp_path / ('f%d.png' % i))
        Image.fromarray(rng.randint(0, 256, size=(50, 80, 3)).astype(np.uint8)).save(p)
        paths.append(os.path.basename(p))
    ds = ImageDataset(paths, str(tmp_path), image_size=(32, 48), dtype=np.float32, use_imgaug=False, transform_probability=0.5)
    for i in range(3):
        img = ds.get_example(i)
        assert img.shape == (3, 32, 48) and img.dtype == np.float32
    gray = str(tmp_path / 'g.png')
    Image.fromarray(rng.randint(0, 256, size=(20, 20)).astype(np.uint8)).save(gray)
    assert ImageDataset(['g.png'], str(tmp_path), image_size=(16, 16)).get_example(0).shape == (3, 16, 16)
    lds = LabeledImageDataset([(paths[0], [10, 20, 40, 60])], str(tmp_path), image_size=(25, 40), label_dtype=np.int32)
    img, label, _ = lds.get_example(0)
    np.testing.assert_array_equal(label, [[5, 10, 20, 30]])
    np.testing.assert_allclose(resize_bbox(np.array([[0., 0., 50., 80.]]), (50, 80), (100, 40)), [[0, 0, 100, 40]])


def test_imgaug_branch_restated(tmp_path):
    """common/datasets/image_dataset.py:57-70,80-83 in the restated form of augment.py: what the three operations do to known
    frames, the sampling structure (Sometimes / SomeOf / random order), and the dataset wiring (augment before the resize)."""
    import random
    from loans_amd.common.datasets import augment as A
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    none = [[0] * 8] * 3
    np.testing.assert_array_equal(A.apply_host(img, none), img)
    np.testing.assert_array_equal(A.apply_host(img, [[1] + [0] * 7] + none[:2]), img[:, ::-1])
    # hue / saturation: a zero shift is the identity up to the 8-bit HSV round trip (<= 2 levels), grey pixels keep their
    # value under any hue shift, a saturation of -255 removes the colour
    back = A.apply_host(img, [[2, 0, 0, 0, 0, 0, 0, 0]] + none[:2])
    assert np.abs(back.astype(int) - img.astype(int)).max() <= 6       # H has 180 levels, S 256: OpenCV's 8-bit HSV loses as much
    grey = np.full((4, 4, 3), 77, np.uint8)
    np.testing.assert_array_equal(A.apply_host(grey, [[2, 20, 0, 0, 0, 0, 0, 0]] + none[:2]), grey)
    flat = A.apply_host(img, [[2, 0, -255, 0, 0, 0, 0, 0]] + none[:2])
    assert (flat.max(axis=2) == flat.min(axis=2)).all() and (flat.max(axis=2) == img.max(axis=2)).all()
    red = np.zeros((2, 2, 3), np.uint8); red[..., 0] = 200
    shifted = A.apply_host(red, [[2, 60, 0, 0, 0, 0, 0, 0]] + none[:2])          # +60 of 180 = 120 degrees: red -> green
    np.testing.assert_array_equal(shifted[0, 0], [0, 200, 0])
    # crop-and-pad: zero sides = identity; padding a constant frame with edge fill keeps it constant, with constant fill the
    # border darkens; cropping a left-right gradient by 10 % on the left drops its darkest columns
    np.testing.assert_array_equal(A.apply_host(img, [[3, 0, 0, 0, 0, 0, 0, 0]] + none[:2]), img)
    const = np.full((16, 16, 3), 90, np.uint8)
    np.testing.assert_array_equal(A.apply_host(const, [[3, 2, 2, 2, 2, 1, 0, 0]] + none[:2]), const)
    padded = A.apply_host(const, [[3, 2, 2, 2, 2, 0, 0, 0]] + none[:2])
    assert padded[0, 0, 0] < 40 and padded[8, 8, 0] == 90
    ramp = np.tile(np.arange(40, dtype=np.uint8)[None, :, None] * 5, (8, 1, 3))
    cropped = A.apply_host(ramp, [[3, 0, 0, 0, -4, 0, 0, 0]] + none[:2])
    assert cropped[0, 0, 0] >= 18 and cropped[0, -1, 0] == ramp[0, -1, 0]
    # sampling: probability 0 -> nothing; probability 1 -> 0..3 distinct operations, every order occurs
    assert A.sample_params(random.Random(0), 20, 30, 0.0) == none
    seen = set()
    r = random.Random(5)
    for _ in range(400):
        rows = A.sample_params(r, 100, 200, 1.0)
        ops_ = tuple(row[0] for row in rows if row[0])
        assert len(set(ops_)) == len(ops_) <= 3
        seen.add(ops_)
        for row in rows:
            if row[0] == 2:
                assert -20 <= row[1] <= 20 and -20 <= row[2] <= 20
            if row[0] == 3:
                assert all(abs(v) <= 10 for v in (row[1], row[3])) and all(abs(v) <= 20 for v in (row[2], row[4])) and row[5] in (0, 1)
    assert len(seen) == 16                                              # 1 + 3 + 6 + 6 ordered subsets
    # dataset: the augmented uint8 frame goes through the same LANCZOS resize and / 255 as an un-augmented one
    paths = []
    for i in range(4):
        Image.fromarray(rng.randint(0, 256, (40, 60, 3)).astype(np.uint8)).save(str(tmp_path / ('a%d.png' % i)))
        paths.append('a%d.png' % i)
    ds = ImageDataset(paths, str(tmp_path), image_size=(32, 32), transform_probability=1.0, augment_seed=3)
    plain = ImageDataset(paths, str(tmp_path), image_size=(32, 32))
    a = np.stack([ds.get_example(i) for i in range(4)])
    assert a.shape == (4, 3, 32, 32) and a.dtype == np.float32 and np.array_equal(a, np.round(a * 255) / np.float32(255))
    assert not np.array_equal(a, np.stack([plain.get_example(i) for i in range(4)]))
    ds.reseed(3)
    np.testing.assert_array_equal(np.stack([ds.get_example(i) for i in range(4)]), a)          # same stream, same frames


def test_frame_cache_pools_stay_inside_the_budget():
    """the device frame cache counted stored frames against its budget while its doubling pool held up to 2-3 x that.
    Now fixed-size pool chunks are allocated only while they fit what is left of the budget, nothing is copied to grow, and a
    gather across chunks returns the stored bytes in the order asked (host logic on CPU tensors)."""
    import torch
    from loans_amd.common.datasets.frame_cache import FrameCache
    H, W = 6, 10
    per = H * W * 3
    cache = FrameCache(budget_bytes=20 * per, where='device')
    cache.POOL_BYTES = 8 * per                      # 8 frames per chunk: at most two chunks fit 20 frames of budget
    rng = np.random.RandomState(0)
    frames = torch.from_numpy(rng.randint(0, 256, (30, H, W, 3)).astype(np.uint8))
    cache._store((H, W), list(range(100, 113)), frames[:13])            # 13 frames: two chunks
    assert cache.allocated == 16 * per <= cache.budget and len(cache.pools[(H, W)]['chunks']) == 2
    cache._store((H, W), list(range(113, 130)), frames[13:30])          # only 3 more fit the two chunks; a third chunk would not
    assert cache.allocated == 16 * per and cache.bytes == 16 * per and len(cache) == 16
    assert set(cache.slots) == set(range(100, 116))
    order = [115, 100, 108, 107, 101]
    got = cache._gather((H, W), [cache.slots[i][1] for i in order], torch.device('cpu'))
    assert torch.equal(got, frames[[i - 100 for i in order]])
    assert cache.lookup(120) is None and cache.lookup(104) is not None
