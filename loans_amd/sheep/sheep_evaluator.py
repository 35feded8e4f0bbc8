"""``SheepMAPEvaluator`` (reference sheep/sheep_evaluator.py:11-66): validation metrics of the
localizer -- mean IoU of the predicted box against the ground truth and the VOC detection AP.

Same constructor / ``__call__(images, labels)`` contract and reported keys (``mean_iou``, ``map``,
``ap/sheep``).  The forward runs on the HIP path in test mode; the box arithmetic that the
reference does on the host with chainercv (``bbox_iou``, ``eval_detection_voc``, both NumPy) is
restated here in NumPy from chainercv 0.9.0's published algorithm.
"""
import numpy as np
import torch

from ..common.utils import Size
from ..runtime.core import report, using_config


def bbox_iou(bbox_a, bbox_b):
    """chainercv.utils.bbox_iou: boxes are (y_min, x_min, y_max, x_max); returns (|a|, |b|)."""
    tl = np.maximum(bbox_a[:, None, :2], bbox_b[:, :2])
    br = np.minimum(bbox_a[:, None, 2:], bbox_b[:, 2:])
    area_i = np.prod(br - tl, axis=2) * (tl < br).all(axis=2)
    area_a = np.prod(bbox_a[:, 2:] - bbox_a[:, :2], axis=1)
    area_b = np.prod(bbox_b[:, 2:] - bbox_b[:, :2], axis=1)
    return area_i / (area_a[:, None] + area_b - area_i)


def eval_detection_voc(pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels, iou_thresh=0.5,
                       use_07_metric=False):
    """chainercv.evaluations.eval_detection_voc for lists of per-image arrays (no `difficult` flags)."""
    n_pos, score, match = {}, {}, {}
    for pb, pl, ps, gb, gl in zip(pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels):
        pb, gb = np.asarray(pb), np.asarray(gb)
        pl, ps, gl = np.asarray(pl).reshape(-1), np.asarray(ps).reshape(-1), np.asarray(gl).reshape(-1)
        for l in np.unique(np.concatenate((pl, gl)).astype(int)):
            pm = pl == l
            pbl, psl = pb[pm], ps[pm]
            order = psl.argsort()[::-1]
            pbl, psl = pbl[order], psl[order]
            gbl = gb[gl == l]
            n_pos[l] = n_pos.get(l, 0) + len(gbl)
            score.setdefault(l, []).extend(psl)
            if len(pbl) == 0:
                continue
            if len(gbl) == 0:
                match.setdefault(l, []).extend((0,) * len(pbl))
                continue
            # VOC evaluation follows integer typed bounding boxes
            pbl = pbl.copy().astype(np.float64)
            pbl[:, 2:] += 1
            gbl = gbl.copy().astype(np.float64)
            gbl[:, 2:] += 1
            iou = bbox_iou(pbl, gbl)
            gt_index = iou.argmax(axis=1)
            gt_index[iou.max(axis=1) < iou_thresh] = -1
            selec = np.zeros(len(gbl), dtype=bool)
            for gi in gt_index:
                if gi >= 0:
                    match.setdefault(l, []).append(0 if selec[gi] else 1)
                    selec[gi] = True
                else:
                    match.setdefault(l, []).append(0)
    n_class = max(n_pos.keys()) + 1 if n_pos else 0
    ap = np.full(n_class, np.nan)
    for l in range(n_class):
        if l not in n_pos:
            continue
        sc, mt = np.array(score.get(l, [])), np.array(match.get(l, []), dtype=np.int8)
        order = sc.argsort()[::-1]
        mt = mt[order]
        tp, fp = np.cumsum(mt == 1), np.cumsum(mt == 0)
        with np.errstate(divide='ignore', invalid='ignore'):
            prec = tp / (fp + tp)
        rec = tp / n_pos[l] if n_pos[l] > 0 else None
        if rec is None:
            continue
        if use_07_metric:
            a = 0.
            for t in np.arange(0., 1.1, 0.1):
                p = 0 if np.sum(rec >= t) == 0 else np.max(np.nan_to_num(prec)[rec >= t])
                a += p / 11
            ap[l] = a
        else:
            mpre = np.concatenate(([0], np.nan_to_num(prec), [0]))
            mrec = np.concatenate(([0], rec, [1]))
            mpre = np.maximum.accumulate(mpre[::-1])[::-1]
            i = np.where(mrec[1:] != mrec[:-1])[0]
            ap[l] = np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])
    return {'ap': ap, 'map': np.nanmean(ap) if len(ap) else np.nan}


class SheepMAPEvaluator:

    def __init__(self, link, device):
        self.link = link
        self.device = device

    def extract_corners(self, bboxes):
        """(top, left, bottom, right) of every sampling grid: channel 1 is y, channel 0 is x; first and last grid point
        (reference sheep/sheep_evaluator.py:19-24)"""
        first, last = bboxes[:, :, 0, 0], bboxes[:, :, -1, -1]          # (B, [x, y]) each
        return np.stack([first[:, 1], first[:, 0], last[:, 1], last[:, 0]], axis=1)

    def scale_bboxes(self, bboxes, image_size):
        bboxes = (bboxes + 1) / 2
        bboxes[:, ::2] *= image_size.height
        bboxes[:, 1::2] *= image_size.width
        return bboxes

    def __call__(self, *inputs):
        images, labels = inputs[:2]
        # chainer's Evaluator runs eval_func with train=False and without building a graph
        with using_config('train', False), using_config('enable_backprop', False):
            _, bboxes = self.link(images)
        bboxes = bboxes.data.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)

        bboxes = self.extract_corners(bboxes)
        bboxes = self.scale_bboxes(bboxes, Size._make(tuple(images.shape[-2:])))

        gt = np.squeeze(labels).reshape(len(bboxes), 4)
        ious = bbox_iou(bboxes.copy(), gt)[np.eye(len(bboxes)).astype(bool)]
        mean_iou = ious.mean()
        report({'mean_iou': mean_iou})

        pred_bboxes = [bbox[np.newaxis, ...].astype(np.int32) for bbox in bboxes]
        pred_scores = np.ones((len(bboxes), 1))
        pred_labels = np.zeros_like(pred_scores)
        gt_bboxes = [np.asarray(lb).reshape(-1, 4) for lb in labels]
        gt_labels = np.zeros_like(pred_scores)

        result = eval_detection_voc(pred_bboxes, pred_labels, pred_scores, gt_bboxes, gt_labels)
        report({'map': result['map']})
        report({'ap/sheep': result['ap'][0]})
        return {'mean_iou': mean_iou, 'map': result['map'], 'ap/sheep': result['ap'][0]}
