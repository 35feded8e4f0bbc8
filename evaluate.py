#!/usr/bin/env python
"""Snapshot sweep (reference evaluate.py:197-386) for the MI355X build: for every ``<prefix>*.npz`` in a log
directory (sorted by iteration, already-evaluated ones skipped unless ``--force-reset``), load the localizer,
run test-mode inference over a labelled set on the HIP path, and append
``{ap, recall, precision, h_mean, bad_iou_mean, snapshot_name}`` to ``<model_dir>/eval_results.json``.

Same metrics as the reference: axis-aligned boxes from the four grid corners clipped to the image
(train_utils/match_bbox.py:46-67), a prediction is a hit iff IoU >= ``--iou-threshold`` (evaluate.py:170-195),
VOC AP via ``eval_detection_voc`` (:283-291).  Not reproduced: model classes re-imported from the log
directory (the classes are fixed here), NMS / deteval / prediction plots.

    python evaluate.py --synthetic 64 sheep_logs SheepLocalizer_ --use-resnet-18
"""
import argparse
import json
import os
import re

import numpy as np
import torch

import loans_amd
from loans_amd.common.datasets.image_dataset import LabeledImageDataset
from loans_amd.datasets import synthetic
from loans_amd.sheep.sheep_evaluator import bbox_iou, eval_detection_voc


def get_aabb_corners(grids, image_size):
    """(top, left, bottom, right) of the axis-aligned box around the sampled quadrilateral, clipped to the image."""
    _, _, height, width = grids.shape
    grids = (grids + 1) / 2
    x_points = np.clip(grids[:, 0] * image_size.width, 0., float(image_size.width))
    y_points = np.clip(grids[:, 1] * image_size.height, 0., float(image_size.height))
    tlx, tly = x_points[:, 0, 0], y_points[:, 0, 0]
    trx, try_ = x_points[:, 0, width - 1], y_points[:, 0, width - 1]
    brx, bry = x_points[:, height - 1, width - 1], y_points[:, height - 1, width - 1]
    blx, bly = x_points[:, height - 1, 0], y_points[:, height - 1, 0]
    return np.minimum(tly, try_), np.minimum(tlx, blx), np.maximum(bly, bry), np.maximum(trx, brx)


class SyntheticLabeled:
    """Composite frames with their paste box as ground truth (y_min, x_min, y_max, x_max)."""

    def __init__(self, n, image_size, seed=123):
        rng = np.random.Generator(np.random.PCG64(seed))
        self.items = []
        for _ in range(n):
            img, (x0, y0, x1, y1) = synthetic.make_composite(rng, image_size[0], image_size[1])
            self.items.append((synthetic.to_chw_float(img), np.array([[y0, x0, y1, x1]], np.float32)))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class Evaluator:
    def __init__(self, args):
        self.args = args
        cls = loans_amd.SheepLocalizer if args.use_resnet_18 else loans_amd.Resnet50SheepLocalizer
        self.localizer = cls(tuple(args.target_size))
        self.results_path = os.path.join(args.model_dir, 'eval_results.json')
        if args.synthetic:
            self.dataset = SyntheticLabeled(args.synthetic, tuple(args.image_size))
        else:
            ds = LabeledImageDataset(args.eval_gt, os.path.dirname(args.eval_gt), image_size=tuple(args.image_size),
                                     label_dtype=np.float32)
            n = len(ds) if args.num_samples is None else min(args.num_samples, len(ds))
            self.dataset = [ds[i][:2] for i in range(n)]
        self.reset()

    def reset(self):
        self.num_hits = self.num_objects = self.num_predicted_objects = 0
        self.bad_ious = [0.0]

    def load_weights(self, snapshot_name):
        loans_amd.load_npz(os.path.join(self.args.model_dir, snapshot_name), self.localizer, strict=False)

    def calc_accuracy(self, predicted, gt_bboxes):
        self.num_objects += len(gt_bboxes)
        self.num_predicted_objects += len(predicted)
        for gt in gt_bboxes:
            ious = bbox_iou(np.tile(gt, (len(predicted), 1)), predicted)
            if not (ious[0] >= self.args.iou_threshold).any():
                self.bad_ious += [float(v) for v in ious[0][ious[0].nonzero()[0]]]
                continue
            self.num_hits += 1

    def evaluate(self, snapshot_name):
        size = loans_amd.Size(*self.args.image_size)
        predictions, gts = [], []
        bs = self.args.batchsize
        for lo in range(0, len(self.dataset), bs):
            batch = [self.dataset[i] for i in range(lo, min(lo + bs, len(self.dataset)))]
            images = np.stack([b[0] for b in batch])
            with loans_amd.using_config('train', False), loans_amd.using_config('enable_backprop', False):
                _, points = self.localizer(images)
            corners = np.stack(get_aabb_corners(points.data.cpu().numpy(), size), axis=1)
            for pred, (_, gt) in zip(corners, batch):
                pred = pred[None].astype(np.float32)
                self.calc_accuracy(pred, gt)
                predictions.append(pred.astype(np.int32))
                gts.append(gt)
        zeros = np.zeros((len(predictions), 1))
        result = eval_detection_voc(predictions, zeros, np.ones_like(zeros), gts, zeros)
        recall = self.num_hits / max(self.num_objects, 1)
        precision = self.num_hits / max(self.num_predicted_objects, 1)
        h_mean = 2 * precision * recall / (precision + recall) if precision + recall != 0 else 0.0
        data = json.load(open(self.results_path)) if os.path.exists(self.results_path) else []
        data.append({"ap": float(result["map"]), "recall": recall, "precision": precision, "h_mean": h_mean,
                     "bad_iou_mean": float(np.mean(self.bad_ious)), "snapshot_name": snapshot_name})
        with open(self.results_path, 'w') as f:
            json.dump(data, f, indent=4)
        return data[-1]


def main(argv=None):
    parser = argparse.ArgumentParser(description="evaluates trained localizer snapshots")
    parser.add_argument("model_dir", help="directory containing the snapshots")
    parser.add_argument("snapshot_prefix", help="prefix of snapshots to evaluate")
    parser.add_argument("--eval-gt", help="tab-separated file: image path, y_min, x_min, y_max, x_max")
    parser.add_argument("--synthetic", type=int, default=0, help="evaluate on N synthetic composites instead")
    parser.add_argument("--gpu", "-g", type=int, default=0)
    parser.add_argument("--num-samples", "-n", type=int)
    parser.add_argument("--batchsize", "-b", type=int, default=1)
    parser.add_argument("--iou-threshold", type=float, default=0.5)
    parser.add_argument("--image-size", type=int, nargs=2, default=(224, 224))
    parser.add_argument("--target-size", type=int, nargs=2, default=(75, 75))
    parser.add_argument("--use-resnet-18", action='store_true', default=False)
    parser.add_argument("--force-reset", action='store_true', default=False)
    args = parser.parse_args(argv)
    if not args.synthetic and not args.eval_gt:
        parser.error("give --eval-gt or --synthetic N")
    torch.cuda.set_device(args.gpu)

    evaluator = Evaluator(args)
    evaluated = []
    if os.path.exists(evaluator.results_path):
        if args.force_reset:
            os.unlink(evaluator.results_path)
        else:
            evaluated = [item['snapshot_name'] for item in json.load(open(evaluator.results_path))]
    num = lambda x: int(m.group(1)) if (m := re.search(r"(\d+).npz", x)) else 0     # noqa: E731
    snapshots = sorted((x for x in os.listdir(args.model_dir)
                        if x not in evaluated and args.snapshot_prefix in x and x.endswith('.npz')), key=num)
    for snapshot in snapshots:
        try:
            evaluator.load_weights(snapshot)
            evaluator.reset()
            print(evaluator.evaluate(snapshot), flush=True)
        except Exception as e:           # reference :375-381: keep sweeping
            print(f"Exception: {e} at snapshot: {snapshot}")
    if os.path.exists(evaluator.results_path):
        data = json.load(open(evaluator.results_path))
        if data:
            best = int(np.argmax([d['ap'] for d in data]))
            print(f"best ap: {data[best]['ap']}  Best Snapshot: {data[best]['snapshot_name']}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
