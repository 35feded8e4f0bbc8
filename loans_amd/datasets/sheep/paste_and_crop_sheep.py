#!/usr/bin/env python
"""Offline generator of assessor training samples (reference datasets/sheep/paste_and_crop_sheep.py:17-154,
195-244): paste a stamp on a background, crop it with a target IoU cycling 0.20 ... 1.00 in steps of 0.05,
resize the crop to ``--output-size`` and write ``images.csv`` rows ``<file>\\t<iou with 4 decimals>``.

The reference needs background photos and RGBA stamps; here both come from the seeded synthetic generator
(loans_amd/datasets/synthetic.py), so the script is runnable offline and its output is read back by
``LabeledImageDataset`` exactly like the reference's.

    python -m loans_amd.datasets.sheep.paste_and_crop_sheep /tmp/assessor_samples --num-samples 64
"""
import argparse
import csv
import os

import numpy as np
from PIL import Image

from .. import synthetic


def main(argv=None):
    parser = argparse.ArgumentParser(description="create synthetic IoU-labelled crops for the assessor")
    parser.add_argument("destination")
    parser.add_argument("--num-samples", type=int, default=100)
    parser.add_argument("--output-size", type=int, nargs=2, default=(75, 75), help="(width, height) of the saved crops")
    parser.add_argument("--image-size", type=int, default=224, help="side of the synthetic composite frame")
    parser.add_argument("--seed", type=int, default=0)
    args = parser.parse_args(argv)

    os.makedirs(args.destination, exist_ok=True)
    crops, labels = synthetic.make_assessor_batch(args.seed, args.num_samples, args.output_size[1], args.output_size[0],
                                                  src=args.image_size)
    with open(os.path.join(args.destination, 'images.csv'), 'w', newline='') as handle:
        writer = csv.writer(handle, delimiter='\t')
        for i, (crop, iou) in enumerate(zip(crops, labels[:, 0])):
            name = '{}.png'.format(i)
            u8 = np.round(crop.transpose(1, 2, 0) * 255).astype(np.uint8)
            Image.fromarray(u8).save(os.path.join(args.destination, name))
            writer.writerow([name, '{:.4f}'.format(float(iou))])
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
