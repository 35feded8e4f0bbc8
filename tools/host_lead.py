#!/usr/bin/env python
"""Where in a step the host runs ahead of the GPU (development probe): HIP events at a few points of the joint step, each with the
host's clock at enqueue time; both clocks are anchored at a synchronize() before the first probed step.
usage: host_lead.py [cfg3|b256|r50] [--fine]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                # noqa: E402
import bench                # noqa: E402
from loans_amd import ops, parallel      # noqa: E402
from loans_amd.runtime import training      # noqa: E402

fine = [a for a in sys.argv[1:] if a == '--fine']        # --fine: also the event behind every function's backward
sys.argv = [a for a in sys.argv if a != '--fine']
what = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
sys.argv = [sys.argv[0]]
args = bench.parse()
over = {'cfg3': dict(image_size=512, batch=128, dtype='bf16', storage='bf16'), 'b256': dict(),
        'r50': dict(image_size=512, batch=64, dtype='bf16', storage='bf16', resnet50=True)}[what]
comm = parallel.init_from_env()
torch.cuda.set_device(0)

real_update = training.StandardUpdater.update
state = {'n': 0}


def update(self):
    state['n'] += 1
    if state['n'] == 6:                 # warm-up done: anchor the clocks, then probe every following step
        torch.cuda.synchronize()
        ops.PROBE_LOG = []
        ops.probe('anchor')
    return real_update(self)


training.StandardUpdater.update = update
w = bench.workload_of(args, steps=8, warmup=3, **over)
r = bench.run_workload(w, comm, 0, False)
log, ops.PROBE_LOG = ops.PROBE_LOG, None
torch.cuda.synchronize()
a_name, a_ev, a_host = log[0]
print('%s: %.3f ms/step' % (what, r['ms_per_step']))
print('%-44s %12s %12s %10s' % ('probe', 'host ms', 'GPU ms', 'lead ms'))
step = -1
for name, ev, host in log[1:]:
    if name == 'step begin':
        step += 1
    if name.startswith('  ') and '--fine' not in fine:
        continue
    if step in ((4,) if fine else (0, 1, 4, 5)):
        h, g = (host - a_host) * 1e3, a_ev.elapsed_time(ev)
        print('step %d %-37s %12.3f %12.3f %10.3f' % (step, name, h, g, g - h))
