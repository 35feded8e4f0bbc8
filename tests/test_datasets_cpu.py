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
    assert paste_and_crop_sheep.main([dest, '--num-samples', '6', '--output-size', '40', '30', '--image-size', '96']) == 0
    rows = list(csv.reader(open(os.path.join(dest, 'images.csv')), delimiter='\t'))
    assert len(rows) == 6 and all(len(r) == 2 and len(r[1].split('.')[1]) == 4 for r in rows)
    ds = LabeledImageDataset(os.path.join(dest, 'images.csv'), dest, image_size=(30, 40), dtype=np.float32, label_dtype=np.float32)
    img, label, dummy = ds.get_example(2)
    assert img.shape == (3, 30, 40) and img.dtype == np.float32 and dummy.shape == (1,)
    assert 0.0 <= img.min() and img.max() <= 1.0
    k = np.round(img * 255)
    np.testing.assert_array_equal(img, (k / 255).astype(np.float32))        # exactly uint8 / 255 (lossless prep)
    np.testing.assert_allclose(label, float(rows[2][1]), atol=1e-6)
    assert len(ds) == 6 and ds[0][0].shape == (3, 30, 40)


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
