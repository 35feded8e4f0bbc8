#!/usr/bin/env python
"""Offline generator of the assessor's training samples (reference datasets/sheep/paste_and_crop_sheep.py:17-154,195-244):
paste an RGBA stamp at a random place of a background photo, cut a crop around it -- either an IoU-TARGETED crop (the target
cycles 0.20, 0.25 ... 1.00; a crop is accepted when target - 0.05 < IoU(crop, paste box) <= target) or a "naive zoom" -- resize
the crop to ``--output-size`` with Pillow's bilinear filter (the reference's ``Image.LINEAR``, today's ``Image.BILINEAR``) and
write ``images/<i>.png`` plus tab-separated ``images.csv`` rows ``images/<i>.png<TAB><iou with 4 decimals>`` that
``LabeledImageDataset`` reads back (common/datasets/image_dataset.py:102-182).

Same command line, file formats, sample recipe and acceptance rule as the reference; written against Pillow 12 and a private
``random.Random(seed)`` (reproducible runs; the reference draws from the global generator).  ``--device cuda`` does the final
bilinear resize of all samples on the GPU (loans_amd/common/datasets/resample.py: bit-identical to Pillow's, tested), the rest
is host-side file handling like the reference's.  Without real photos, ``--synthetic N`` first writes N seeded backgrounds and
RGBA stamps (loans_amd/datasets/synthetic.py) and then runs the same file-based pipeline on them.

    python -m loans_amd.datasets.sheep.paste_and_crop_sheep backgrounds/ out/ --stamps sheep1.png sheep2.png --zoom-mode
    python -m loans_amd.datasets.sheep.paste_and_crop_sheep - out/ --synthetic 8 --zoom-mode --num-samples 64
"""
import argparse
import csv
import json
import os
import random

import numpy as np
from PIL import Image

from ...sheep.sheep_evaluator import bbox_iou

IOU_RANGES = list(range(20, 105, 5))             # reference :13


class SampleGenerator:
    """One stream of samples: holds the cycling IoU target and the random generator (reference: module globals)."""

    def __init__(self, seed=None, image_size=(224, 224), enlarge_region=(0, 0, 0, 0), bbox_sizes=None, zoom_mode=False):
        self.rng = random.Random(seed)
        self.iou_index = -1
        self.image_size = None if image_size is None else tuple(image_size)
        self.crop_extra = tuple(enlarge_region)
        self.bbox_sizes = bbox_sizes
        self.zoom_mode = zoom_mode

    # ---- crops (reference :17-105) ----
    def _crop_box(self, image, paste_box, crop_w, crop_h, desired_iou):
        rng = self.rng
        if desired_iou < 0.0:
            x = rng.randint(0, image.width - crop_w)
            y = rng.randint(0, image.height - crop_h)
        else:
            dev_x = int(crop_w // 2 * (1.0 - desired_iou))
            dev_y = int(crop_h // 2 * (1.0 - desired_iou))
            x = rng.randint(max(paste_box[0] - dev_x, 0), min(paste_box[0] + dev_x, image.width - crop_w))
            y = rng.randint(max(paste_box[1] - dev_y, 0), min(paste_box[1] + dev_y, image.height - crop_h))
        # the reference clips the bottom edge with image.WIDTH as well (:40); kept, frames are square in its use
        return np.array([x, y, min(x + crop_w, image.width), min(y + crop_h, image.width)])

    def iou_crop(self, image, paste_x, paste_y, stamp):
        rng = self.rng
        self.iou_index = (self.iou_index + 1) % len(IOU_RANGES)
        desired = min(IOU_RANGES[self.iou_index] / 100, 1.0)
        paste_box = np.array([paste_x, paste_y, paste_x + stamp.width, paste_y + stamp.height])
        size = paste_box[2:] - paste_box[:2]
        spread = 1.0 - desired
        for _ in range(200):
            for _ in range(200):
                if desired < 0.3:
                    cw = int(min(stamp.width + (1 - desired) * 10 * stamp.width, image.width))
                    ch = int(min(stamp.height + (1 - desired) * 10 * stamp.height, image.height))
                else:
                    cw = rng.randint(max(int(size[0] - size[0] * spread), 1), int(size[0] + size[0] * spread))
                    ch = rng.randint(max(int(size[1] - size[1] * spread), 1), int(size[1] + size[1] * spread))
                try:
                    box = self._crop_box(image, paste_box.astype(np.int32), cw, ch, desired)
                except ValueError:              # empty randint range: this crop size does not fit beside the paste box
                    continue
                iou = float(abs(np.max(bbox_iou(box[None].astype(np.float64), paste_box[None].astype(np.float64)))))
                if desired - 0.05 < iou <= desired:
                    return image.crop(tuple(int(v) for v in box)), iou, tuple(int(v) for v in box)
        raise ValueError("No Good BBOX Found")

    def naive_zoom(self, image, paste_x, paste_y, stamp):
        rng = self.rng
        zoom = rng.random() * 10 + 0.3
        cw = min(stamp.width + zoom * stamp.width, image.width)
        ch = min(stamp.height + zoom * stamp.height, image.height)
        rx, ry = rng.random(), rng.random()
        hi = [min(paste_x, image.width - cw), min(paste_y, image.height - ch)]
        lo = [max(paste_x + stamp.width - cw, 0), max(paste_y + stamp.height - ch, 0)]
        hi = [max(h, l) for h, l in zip(hi, lo)]
        px, py = (int(l + r * (h - l)) for l, h, r in zip(lo, hi, (rx, ry)))
        box = [px, py, px + cw, py + ch]
        paste_box = np.array([paste_x, paste_y, paste_x + stamp.width, paste_y + stamp.height], np.float64)
        iou = float(bbox_iou(np.array(box, np.float64)[None], paste_box[None])[0, 0])
        return image.crop(box), iou, tuple(box)

    # ---- one sample (reference :108-154) ----
    def create_sample(self, image, stamp):
        """image: RGBA background, stamp: RGBA.  Returns (crop, label or None, info) -- info holds the paste and crop boxes."""
        rng = self.rng
        bbox_size = None
        if self.bbox_sizes is not None:
            bbox_size = rng.choice(self.bbox_sizes)
        else:
            if self.image_size is None:
                raise ValueError("without base bboxes the stamp size is drawn from the image size: give --image-size")
            stamp = stamp.resize((rng.randint(self.image_size[0] // 15, self.image_size[0] // 2),
                                  rng.randint(self.image_size[1] // 15, self.image_size[1] // 2)), Image.LANCZOS)
        if self.image_size:
            factors = [n / o for n, o in zip(self.image_size, image.size)]
            image = image.resize(self.image_size, Image.LANCZOS)
            if bbox_size is not None:
                bbox_size = [int(d * f) for d, f in zip(bbox_size, factors)]
        if bbox_size is not None:
            stamp = stamp.resize(tuple(bbox_size), Image.LANCZOS)

        ex = self.crop_extra
        paste_x = rng.randint(ex[0], image.width - stamp.width - ex[2])
        paste_y = rng.randint(ex[1], image.height - stamp.height - ex[3])
        layer = Image.new('RGBA', image.size)
        layer.paste(stamp, (paste_x, paste_y))
        image = Image.alpha_composite(image, layer)
        info = {'paste_box': (paste_x, paste_y, paste_x + stamp.width, paste_y + stamp.height), 'frame_size': image.size}

        if self.zoom_mode:
            if self.image_size is None:
                raise ValueError("if you are using zoom mode, image size can not be None")
            crop, iou, box = (self.iou_crop if rng.random() >= 0.3 else self.naive_zoom)(image, paste_x, paste_y, stamp)
            info['crop_box'] = box
            return crop, iou, info
        box = (paste_x - ex[0], paste_y - ex[1], paste_x + stamp.width + ex[2], paste_y + stamp.height + ex[3])
        info['crop_box'] = box
        return image.crop(box), None, info


def get_base_bbox_sizes(base_bbox_path):
    """(width, height) of every well-formed box of a ground-truth json (reference :157-175; boxes are y0, x0, y1, x1)"""
    with open(base_bbox_path) as handle:
        data = json.load(handle)
    sizes = set()
    for item in data:
        for box in item['bounding_boxes']:
            size = (box[3] - box[1], box[2] - box[0])
            if all(v > 0 for v in size):
                sizes.add(size)
    return sorted(sizes)


def write_synthetic_sources(directory, n, seed, size=(320, 320)):
    """seeded background photos and RGBA stamps for a run without real data; returns (background dir, stamp paths)"""
    from .. import synthetic
    rng = np.random.Generator(np.random.PCG64(seed))
    bg_dir = os.path.join(directory, 'backgrounds')
    os.makedirs(bg_dir, exist_ok=True)
    for i in range(n):
        Image.fromarray(synthetic._low_freq_noise(rng, size[1], size[0])).save(os.path.join(bg_dir, 'bg%d.png' % i))
    stamps = []
    for i in range(max(2, n // 4)):
        tex, alpha = synthetic._stamp(rng, 96, 128)
        rgba = np.concatenate([tex, alpha * 255], axis=2).round().clip(0, 255).astype(np.uint8)
        path = os.path.join(directory, 'stamp%d.png' % i)
        Image.fromarray(rgba, 'RGBA').save(path)
        stamps.append(path)
    return bg_dir, stamps


def resize_samples(samples, output_size, device=None):
    """the final ``sample.resize(output_size, Image.LINEAR)`` of every sample (reference :218): Pillow on the host, or -- device
    given -- the same integer resampler on the GPU (bit-identical), frames of equal size batched"""
    if device is None:
        return [s.resize(tuple(output_size), Image.BILINEAR) for s in samples]
    import torch
    from ...common.datasets.resample import resize_bilinear
    out = [None] * len(samples)
    groups = {}
    for i, s in enumerate(samples):
        groups.setdefault(s.size, []).append(i)
    for (w, h), idx in groups.items():
        rgba = np.stack([np.asarray(samples[i].convert('RGBA')) for i in idx])            # crops of an RGBA composite
        if (rgba[..., 3] != 255).any():
            # Pillow resizes RGBA through premultiplied alpha (Image.resize: RGBA -> RGBa -> resize -> RGBA); for the
            # generator's composites (an opaque photo under the stamp) alpha is 255 everywhere and that is the identity.
            # Anything else stays with Pillow.
            for i in idx:
                out[i] = samples[i].resize(tuple(output_size), Image.BILINEAR)
            continue
        t = torch.from_numpy(np.ascontiguousarray(rgba[..., :3])).to(device)
        rgb = resize_bilinear(t, (output_size[1], output_size[0]), as_float=False).cpu().numpy()
        alpha = np.full(rgb.shape[:3] + (1,), 255, np.uint8)
        for j, i in enumerate(idx):
            out[i] = Image.fromarray(np.concatenate([rgb[j], alpha[j]], axis=2), 'RGBA')
    return out


def generate(args):
    rng_seed = args.seed
    if args.synthetic:
        background_dir, stamp_paths = write_synthetic_sources(os.path.join(args.destination, '_synthetic'), args.synthetic,
                                                              0 if rng_seed is None else rng_seed)
    else:
        background_dir, stamp_paths = args.background_image_dir, args.stamps
    if not stamp_paths:
        raise SystemExit('give --stamps (RGBA images) or --synthetic N')
    all_images = sorted(os.listdir(background_dir))
    stamps = [Image.open(p).convert('RGBA') for p in stamp_paths]
    os.makedirs(os.path.join(args.destination, 'images'), exist_ok=True)
    bbox_sizes = get_base_bbox_sizes(args.base_bboxes) if args.base_bboxes is not None else None
    gen = SampleGenerator(rng_seed, args.image_size, args.enlarge_region, bbox_sizes, args.zoom_mode)

    crops, rows, infos = [], [], []
    for i in range(args.num_samples):
        image_path = gen.rng.choice(all_images)
        stamp = gen.rng.choice(stamps)
        if gen.rng.random() >= 0.5:                          # randomly flip stamps horizontally
            stamp = stamp.transpose(Image.FLIP_LEFT_RIGHT)
        try:
            with Image.open(os.path.join(background_dir, image_path)) as bg:
                crop, label, info = gen.create_sample(bg.convert('RGBA'), stamp)
        except ValueError:
            continue
        crops.append(crop)
        infos.append(info)
        name = 'images/{}.png'.format(i)
        rows.append([name] if label is None else [name, format(label, '.4f')])
    for row, sample in zip(rows, resize_samples(crops, args.output_size, args.device)):
        sample.save(os.path.join(args.destination, row[0]))
    with open(os.path.join(args.destination, 'images.csv'), 'w', newline='') as handle:
        csv.writer(handle, delimiter='\t').writerows(rows)
    return rows, infos


def build_parser():
    parser = argparse.ArgumentParser(description="Put the stamp on any place in the input image, save crops around it as "
                                                 "IoU-labelled samples for the assessor")
    parser.add_argument("background_image_dir", help="directory that contains all possible background images ('-' with --synthetic)")
    parser.add_argument("destination", help="destination directory (images/ and images.csv are created in it)")
    parser.add_argument("--stamps", nargs='+', help="RGBA stamp images")
    parser.add_argument("--num-samples", type=int, default=10000)
    parser.add_argument("--output-size", type=int, nargs=2, default=(75, 75), help="(width, height) of the saved crops")
    parser.add_argument("--image-size", type=int, nargs=2, default=(224, 224), help="size of image for network (important for zoom mode)")
    parser.add_argument("--enlarge-region", type=int, nargs=4, default=(0, 0, 0, 0), help="pixels the template is enlarged by (l, t, r, b)")
    parser.add_argument("--base-bboxes", help="json with ground-truth boxes whose sizes are used for the pasted stamps")
    parser.add_argument("--zoom-mode", action='store_true', default=False, help="IoU-targeted / zoomed crops with IoU labels")
    parser.add_argument("--seed", type=int, default=None, help="seed of the generator's private random stream")
    parser.add_argument("--synthetic", type=int, default=0, help="write N seeded backgrounds + stamps first and use those")
    parser.add_argument("--device", default=None, help="e.g. cuda: do the final bilinear resize on the GPU (same bytes)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    rows, _ = generate(args)
    print('wrote %d samples to %s' % (len(rows), args.destination))
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
