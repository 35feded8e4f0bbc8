"""Host-side validation metrics (sheep/sheep_evaluator.py:32-66 -> chainercv bbox_iou / eval_detection_voc)."""
import numpy as np

from loans_amd.sheep.sheep_evaluator import bbox_iou, eval_detection_voc


def test_bbox_iou_known_values():
    a = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30]], np.float32)
    b = np.array([[0, 0, 10, 10], [5, 5, 15, 15], [0, 0, 10, 10]], np.float32)
    iou = bbox_iou(a, b)
    assert iou.shape == (3, 3)
    np.testing.assert_allclose(np.diag(iou), [1.0, 25.0 / 175.0, 0.0])


def test_voc_ap_single_class_all_hits_and_half_hits():
    gt = [np.array([[10, 10, 50, 50]], np.float32) for _ in range(4)]
    hit = np.array([[12, 11, 49, 52]], np.int32)
    miss = np.array([[60, 60, 90, 90]], np.int32)
    ones, zeros = np.ones((4, 1)), np.zeros((4, 1))
    r = eval_detection_voc([hit] * 4, zeros, ones, gt, zeros)
    assert r['map'] == 1.0 and r['ap'][0] == 1.0
    r = eval_detection_voc([miss] * 4, zeros, ones, gt, zeros)
    assert r['map'] == 0.0
    # equal scores are ranked in reversed index order (argsort()[::-1]); hits last -> precision 1/3, 2/4
    r = eval_detection_voc([hit, hit, miss, miss], zeros, ones, gt, zeros)
    np.testing.assert_allclose(r['ap'][0], 0.25 * (0.5) + 0.25 * 0.5)
    # hits ranked first -> AP = recall reached at precision 1
    r = eval_detection_voc([miss, miss, hit, hit], zeros, ones, gt, zeros)
    np.testing.assert_allclose(r['ap'][0], 0.5)


def test_get_aabb_corners_clips_and_orders():
    import evaluate
    import loans_amd
    from oracle import chainer_ops as C
    theta = np.array([[[0.5, 0.2, 0.1], [-0.1, 0.6, 0.0]], [[1.5, 0, 0], [0, 1.5, 0]]], np.float32)
    grid, _ = C.st_grid_fwd(theta, (5, 7))
    top, left, bottom, right = evaluate.get_aabb_corners(grid, loans_amd.Size(100, 200))
    g = (grid + 1) / 2
    xs, ys = np.clip(g[:, 0] * 200, 0, 200), np.clip(g[:, 1] * 100, 0, 100)
    np.testing.assert_allclose(top, np.minimum(ys[:, 0, 0], ys[:, 0, -1]))
    np.testing.assert_allclose(right, np.maximum(xs[:, 0, -1], xs[:, -1, -1]))
    assert (top[1], left[1], bottom[1], right[1]) == (0.0, 0.0, 100.0, 200.0)      # scale 1.5 -> clipped to the frame
