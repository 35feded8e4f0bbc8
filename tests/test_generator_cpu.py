"""CPU: the file-based paste-and-crop generator (reference datasets/sheep/paste_and_crop_sheep.py:17-154,195-244) on seeded
backgrounds / stamps written to disk: csv format, output size, the IoU acceptance rule of the targeted crops (label = IoU of
the crop box with the paste box, inside (target - 0.05, target] for the cycling target), the plain template mode, the
round trip through LabeledImageDataset, reproducibility of a seeded run."""
import csv
import os

import numpy as np
from PIL import Image

from loans_amd.datasets.sheep import paste_and_crop_sheep as G
from loans_amd.sheep.sheep_evaluator import bbox_iou


def _run(tmp_path, name, extra):
    dest = str(tmp_path / name)
    args = G.build_parser().parse_args(['-', dest, '--synthetic', '6', '--num-samples', '40', '--seed', '7'] + extra)
    return dest, G.generate(args)


def test_zoom_mode_samples_and_labels(tmp_path):
    dest, (rows, infos) = _run(tmp_path, 'zoom', ['--zoom-mode', '--image-size', '224', '224', '--output-size', '75', '60'])
    assert len(rows) >= 30 and len(rows) == len(infos)
    with open(os.path.join(dest, 'images.csv')) as f:
        on_disk = list(csv.reader(f, delimiter='\t'))
    assert on_disk == rows and all(len(r) == 2 and r[0].startswith('images/') and r[0].endswith('.png') for r in rows)
    for (name, label), info in zip(rows, infos):
        with Image.open(os.path.join(dest, name)) as im:
            assert im.size == (75, 60)
        iou = bbox_iou(np.array([info['crop_box']], np.float64), np.array([info['paste_box']], np.float64))[0, 0]
        assert label == format(iou, '.4f')
        assert info['frame_size'] == (224, 224)
    labels = [float(r[1]) for r in rows]
    assert 0.0 < min(labels) < 0.3 and max(labels) > 0.8


def test_iou_targeted_crops_hit_their_window():
    """the acceptance rule of get_iou_crop (reference :45-81): the target cycles 0.20, 0.25 ... 1.00 and an accepted crop has
    target - 0.05 < IoU(crop, paste box) <= target; a target no crop can meet raises (the caller skips the sample)"""
    gen = G.SampleGenerator(seed=3, image_size=(224, 224), zoom_mode=True)
    image = Image.new('RGBA', (224, 224), (10, 20, 30, 255))
    stamp = Image.new('RGBA', (60, 44), (200, 100, 50, 255))
    seen = []
    for _ in range(2 * len(G.IOU_RANGES)):
        target = G.IOU_RANGES[(gen.iou_index + 1) % len(G.IOU_RANGES)] / 100
        try:
            crop, iou, box = gen.iou_crop(image, 80, 90, stamp)
        except ValueError:
            continue
        assert target - 0.05 < iou <= target, (target, iou)
        assert crop.size == (box[2] - box[0], box[3] - box[1])
        got = bbox_iou(np.array([box], np.float64), np.array([[80, 90, 140, 134]], np.float64))[0, 0]
        assert abs(got - iou) < 1e-12
        seen.append(target)
    assert len(set(seen)) >= 10, seen                      # most of the 17 targets are reachable for this stamp


def test_template_mode_and_dataset_round_trip(tmp_path):
    dest, (rows, infos) = _run(tmp_path, 'plain', ['--enlarge-region', '4', '2', '4', '2'])
    assert all(len(r) == 1 for r in rows)                                   # no label without zoom mode
    for info in infos:
        px0, py0, px1, py1 = info['paste_box']
        assert info['crop_box'] == (px0 - 4, py0 - 2, px1 + 4, py1 + 2)     # the stamp plus the enlarged margin
    dest, (rows, _) = _run(tmp_path, 'zoom2', ['--zoom-mode'])
    from loans_amd.common.datasets.image_dataset import LabeledImageDataset
    ds = LabeledImageDataset(os.path.join(dest, 'images.csv'), dest, image_size=(75, 75), dtype=np.float32, label_dtype=np.float32)
    assert len(ds) == len(rows)
    image, label, dummy = ds[3]
    assert image.shape == (3, 75, 75) and image.dtype == np.float32 and 0.0 <= image.min() and image.max() <= 1.0
    np.testing.assert_allclose(label, [float(rows[3][1])], rtol=1e-6)


def test_seeded_runs_are_reproducible(tmp_path):
    d1, (r1, i1) = _run(tmp_path, 'a', ['--zoom-mode'])
    d2, (r2, i2) = _run(tmp_path, 'b', ['--zoom-mode'])
    assert r1 == r2 and i1 == i2
    for name, _ in r1[:5]:
        with Image.open(os.path.join(d1, name)) as a, Image.open(os.path.join(d2, name)) as b:
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_base_bbox_sizes(tmp_path):
    import json
    path = str(tmp_path / 'gt.json')
    json.dump([{'bounding_boxes': [[10, 20, 50, 80], [5, 5, 5, 9]]}, {'bounding_boxes': [[0, 0, 30, 40]]}], open(path, 'w'))
    assert G.get_base_bbox_sizes(path) == [(40, 30), (60, 40)]               # (width, height); the degenerate box is dropped
