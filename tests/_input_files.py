"""Shared by the input-path tests: a small on-disk data set in the reference's file formats -- a path list of training
frames of mixed sizes (``ImageDataset``), a tab-separated validation csv ``<frame>\\t<top>\\t<left>\\t<bottom>\\t<right>``
(``LabeledImageDataset``, int labels) and the assessor's ``images.csv`` written by the paste-and-crop generator itself
(datasets/sheep/paste_and_crop_sheep.py: ``<crop>\\t<iou>``)."""
import json
import os

import numpy as np
from PIL import Image


def write_files(root, n_train=10, n_val=6, crop=16, as_json=False):
    from loans_amd.datasets import synthetic
    from loans_amd.datasets.sheep import paste_and_crop_sheep as G
    root = str(root)
    rng = np.random.Generator(np.random.PCG64(42))
    os.makedirs(os.path.join(root, 'frames'), exist_ok=True)
    sizes = [(96, 128), (80, 80), (120, 90)]
    names, boxes = [], []
    for i in range(n_train + n_val):
        h, w = sizes[i % len(sizes)]
        img, (x0, y0, x1, y1) = synthetic.make_composite(rng, h, w)
        name = 'frames/f%02d.png' % i
        Image.fromarray(np.asarray(img)[..., :3].astype(np.uint8)).save(os.path.join(root, name))
        names.append(name)
        boxes.append((int(y0), int(x0), int(y1), int(x1)))
    train = os.path.join(root, 'train.json' if as_json else 'train.txt')
    val = os.path.join(root, 'val.json' if as_json else 'val.csv')
    if as_json:
        json.dump([{"image": n} for n in names[:n_train]], open(train, 'w'))
        json.dump([{"image": n, "bounding_boxes": [list(b)]} for n, b in zip(names[n_train:], boxes[n_train:])], open(val, 'w'))
    else:
        open(train, 'w').write('\n'.join(names[:n_train]) + '\n')
        open(val, 'w').write(''.join('%s\t%d\t%d\t%d\t%d\n' % ((n,) + b) for n, b in zip(names[n_train:], boxes[n_train:])))
    ref_dir = os.path.join(root, 'reference')
    args = G.build_parser().parse_args(['-', ref_dir, '--synthetic', '4', '--num-samples', '24', '--seed', '5', '--zoom-mode',
                                        '--image-size', '64', '64', '--output-size', str(crop), str(crop)])
    rows, _ = G.generate(args)
    assert len(rows) >= 12
    return train, val, os.path.join(ref_dir, 'images.csv')
