#!/usr/bin/env python
"""What the step's streams cost each other (development probe, NOT a measurement of the product): configs[2] / configs[1]
timed (a) as shipped, (b) with the weight gradients skipped -- the main stream alone --, (c) with them on the main stream.
usage: step_probe.py [cfg3|b256|r50] [tune file]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402
import bench                # noqa: E402
from loans_amd import ops, parallel      # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
sys.argv = [sys.argv[0]]
args = bench.parse()
over = {'cfg3': dict(image_size=512, batch=128, dtype='bf16', storage='bf16'), 'b256': dict(),
        'r50': dict(image_size=512, batch=64, dtype='bf16', storage='bf16', resnet50=True)}[what]
comm = parallel.init_from_env()
torch.cuda.set_device(0)


def run(label):
    w = bench.workload_of(args, steps=10, warmup=3, **over)
    r = bench.run_workload(w, comm, 0, False)
    print('%-40s %.3f ms/step' % (label, r['ms_per_step']), flush=True)


run('as shipped')
real = ops._conv_wgrad
ops._conv_wgrad = lambda *a, **k: None
run('weight gradients skipped (main stream alone)')
ops._conv_wgrad = real
ops.ASYNC_WGRAD = False
run('weight gradients on the main stream')
ops.ASYNC_WGRAD = True
