#!/usr/bin/env python
"""Tile tables tuned INSIDE the step.  The autotuner times every candidate of a layer alone on the device; in the step two
streams share it, and what wins alone (one 512-thread block per CU, full rounds of block slots) is not always what the step
is shortest with.  This tool starts from a table (the autotuner's), builds the bench workload once, and walks the table's
entries: for every other candidate of an entry it runs the whole step a few times and keeps the candidate if the step got
shorter by more than the noise (confirmed by a second measurement of both).  Development tool: its output is a tile table like
any other (profiles/r*_<leg>_tune.json), checked by tests/test_gpu_tune_tables.py like any other.
usage: instep_tune.py <table in> <table out> [--steps 12] [--eps 0.04] [--max-entries N] [bench.py workload arguments]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument('table_in')
ap.add_argument('table_out')
ap.add_argument('--steps', type=int, default=12)
ap.add_argument('--eps', type=float, default=0.04, help='ms per step a candidate has to win by')
ap.add_argument('--max-entries', type=int, default=0)
ap.add_argument('--budget-s', type=float, default=900.0)
args, rest = ap.parse_known_args()
sys.argv = ['bench.py'] + rest
import bench                                                # noqa: E402
import loans_amd                                            # noqa: E402
from loans_amd import ops, parallel                         # noqa: E402
from loans_amd.datasets import synthetic                    # noqa: E402
from loans_amd.runtime import training                      # noqa: E402

bargs = bench.parse()
w = bench.workload_of(bargs)
comm = parallel.init_from_env()
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
ops.load_tune_table(args.table_in)

seen = {}
orig_tuned_tile = ops._tuned_tile


def recording_tuned_tile(geo, mode, run, candidates, cold=False):
    cands = [t for t in candidates if not (t & 16)] if ops.COMPUTE == 'bf16' else list(candidates)
    seen.setdefault((geo.key, mode), (geo, cands))
    return orig_tuned_tile(geo, mode, run, candidates, cold)


ops._tuned_tile = recording_tuned_tile

B, hw, crop = w.batch, w.image_size, w.target_size
pool = 32
frames = synthetic.make_frames(1000, pool, hw, hw)
real, labels = synthetic.make_assessor_batch(2000, pool, crop, crop)
reps = (B + pool - 1) // pool
frames_d = torch.from_numpy(np.tile(frames, (reps, 1, 1, 1))[:B]).to(dev)
real_d = torch.from_numpy(np.tile(real, (reps, 1, 1, 1))[:B]).to(dev)
labels_d = torch.from_numpy(np.tile(labels, (reps, 1))[:B]).to(dev)
np.random.seed(1234)
localizer = (loans_amd.Resnet50SheepLocalizer if w.resnet50 else loans_amd.SheepLocalizer)((crop, crop))
localizer.param_predictor.W.set_logical((1e-3 * np.random.standard_normal(localizer.param_predictor.W.logical_shape)).astype(np.float32))
discriminator = loans_amd.ResnetAssessor()
localizer.set_precision(w.dtype, w.storage)
discriminator.set_precision(w.dtype, w.storage)
with loans_amd.using_config('enable_backprop', False):
    discriminator(real_d[:2])
localizer.finalize(dev)
updater = loans_amd.SheepAssessor(
    models=[localizer, discriminator],
    iterator={'main': training.DeviceBatchIterator([frames_d]), 'real': training.DeviceBatchIterator([(real_d, labels_d)])},
    optimizer={'opt_gen': loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(localizer),
               'opt_dis': loans_amd.Adam(alpha=1e-3, amsgrad=True).setup(discriminator)},
    converter=training.identity_converter, device=0, comm=comm)
for _ in range(6):
    updater.update()
torch.cuda.synchronize()


def measure(n=None, blocks=3):
    n = n or args.steps
    ts = []
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            updater.update()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3 / n)
    return float(np.median(ts))


def set_tile(geo, mode, tile):
    geo.tuned[mode] = tile
    for k in [k for k in geo.tuned if k.startswith('~')]:
        del geo.tuned[k]


entries = [(k, v) for k, v in seen.items() if len(v[1]) > 1]
print('%d tuned entries with more than one candidate; %d launches of them per step are not counted here' % (len(entries), 0), flush=True)
best = measure()
print('start: %.3f ms per step' % best, flush=True)
t_start = time.time()
changed = []
n_eval = 0
for idx, ((key, mode), (geo, cands)) in enumerate(entries):
    if args.max_entries and idx >= args.max_entries:
        break
    if time.time() - t_start > args.budget_s:
        print('time budget used up after %d entries' % idx, flush=True)
        break
    cur = geo.tuned.get(mode)
    if cur is None:
        continue
    for cand in cands:
        if cand == cur:
            continue
        try:
            set_tile(geo, mode, cand)
            for _ in range(2):
                updater.update()
            t = measure(blocks=2)
        except RuntimeError as e:           # a candidate the launcher refuses for this call's flags
            set_tile(geo, mode, cur)
            print('  %s %s: tile %d refused (%s)' % (key, mode, cand, str(e)[:60]), flush=True)
            continue
        n_eval += 1
        if t < best - args.eps:
            # confirm: the incumbent again, then the candidate again
            set_tile(geo, mode, cur)
            for _ in range(2):
                updater.update()
            t_cur = measure(blocks=2)
            set_tile(geo, mode, cand)
            for _ in range(2):
                updater.update()
            t2 = measure(blocks=2)
            if t2 < t_cur - args.eps:
                print('  %s %s: tile %d -> %d: %.3f -> %.3f ms per step' % (key, mode, cur, cand, t_cur, t2), flush=True)
                changed.append((key, mode, cur, cand, t_cur, t2))
                cur, best = cand, min(t2, t)
                continue
            best = min(best, t_cur) if t_cur < best else t_cur      # the baseline drifted: follow it
        set_tile(geo, mode, cur)
    if idx % 10 == 9:
        for _ in range(2):
            updater.update()
        best = measure()
        print('after %d entries (%d evaluations, %d changes): %.3f ms per step' % (idx + 1, n_eval, len(changed), best), flush=True)
for _ in range(2):
    updater.update()
final = measure()
print('end: %.3f ms per step, %d entries changed in %d evaluations' % (final, len(changed), n_eval), flush=True)
n = ops.save_tune_table(args.table_out)
print('wrote %d shapes to %s' % (n, args.table_out), flush=True)
