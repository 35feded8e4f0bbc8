#!/usr/bin/env python
"""Development: the same two steps with and without the step workspace (ops._StepArena), every Function's forward outputs and
backward results of the SECOND step fingerprinted: the first op whose result differs.  usage: arena_diff.py [f32|bf16]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loans_amd                                            # noqa: E402
from loans_amd import ops                                   # noqa: E402
from loans_amd.runtime import core, training                # noqa: E402
from tests.gpu_util import build_pair, dev, inputs          # noqa: E402

arm = sys.argv[1] if len(sys.argv) > 1 else 'f32'
B, H, W, crop = 4, 128, 128, (32, 32)
ops.SPLITK = False
ops.TUNE_POLICY = 'fixed'
frames, real, labels = inputs(81, B, H, W, crop)
log = None


def finger(t):
    if not torch.is_tensor(t):
        return None
    f = t.double()
    return (tuple(t.shape), float(f.sum()), float(f.abs().sum()))


fwd = core.Function.__call__


def checked_call(self, *inputs):
    out = fwd(self, *inputs)
    if log is not None:
        outs = out if isinstance(out, tuple) else (out,)
        log.append(('fwd', type(self).__name__, [finger(o.data) for o in outs]))
    return out


core.Function.__call__ = checked_call
for cls in list(core.Function.__subclasses__()) + [c for k in core.Function.__subclasses__() for c in k.__subclasses__()]:
    if 'backward' in cls.__dict__:
        def make(orig, name):
            def checked_backward(self, inputs, gys):
                gxs = orig(self, inputs, gys)
                if log is not None:
                    log.append(('bwd', name, [finger(g) for g in (gxs if isinstance(gxs, tuple) else (gxs,))]))
                return gxs
            return checked_backward
        cls.backward = make(cls.__dict__['backward'], cls.__name__)


def run(arena_on):
    global log
    ops.STEP_ARENA = arena_on
    ops._step_arenas.clear()
    loc, dis = build_pair(82, crop)
    loc.set_precision(arm)
    dis.set_precision(arm)
    up = loans_amd.SheepAssessor(
        models=[loc, dis], iterator={'main': training.DeviceBatchIterator([dev(frames)]),
                                     'real': training.DeviceBatchIterator([(dev(real), dev(labels))])},
        optimizer={'opt_gen': loans_amd.Adam(alpha=1e-4, amsgrad=True).setup(loc),
                   'opt_dis': loans_amd.Adam(alpha=1e-4, amsgrad=True).setup(dis)},
        converter=training.identity_converter, device=0)
    logs = []
    for it in range(3):
        log = []
        up.update()
        torch.cuda.synchronize()
        logs.append(log + [('grad', 'localizer', [finger(loc.arena.grad)]), ('grad', 'assessor', [finger(dis.arena.grad)]),
                           ('data', 'localizer', [finger(loc.arena.data)]), ('data', 'assessor', [finger(dis.arena.data)])])
    log = None
    return logs


a = run(False)
a2 = run(False)
b = run(True)
for name, x, y in (('plain vs plain', a, a2), ('plain vs workspace', a, b)):
    for it in range(3):
        n = 0
        assert len(x[it]) == len(y[it]), (len(x[it]), len(y[it]))
        for i, (p, q) in enumerate(zip(x[it], y[it])):
            if p != q:
                n += 1
                if n <= 6:
                    print('%s, step %d, op %d of %d: %s %s' % (name, it + 1, i, len(x[it]), p[0], p[1]))
                    for u, v in zip(p[2], q[2]):
                        if u != v:
                            print('     ', u, '\n     ', v)
        print('%s, step %d: %d of %d results differ' % (name, it + 1, n, len(x[it])), flush=True)
