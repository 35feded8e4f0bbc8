#!/usr/bin/env python
"""Development: which op reads memory it (or its producer) never wrote?  The step workspace (ops._StepArena) is filled with 0xFF
bytes (NaN in every float type) at the start of each step; every Function's forward / backward outputs are tested for NaN and the
first offender is reported.  usage: arena_poison.py [f32|bf16] [B H W]"""
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
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4, 128, 128)
crop = (32, 32)
ops.SPLITK = False
ops.TUNE_POLICY = 'fixed'
frames, real, labels = inputs(81, B, H, W, crop)
loc, dis = build_pair(82, crop)
loc.set_precision(arm)
dis.set_precision(arm)
up = loans_amd.SheepAssessor(
    models=[loc, dis], iterator={'main': training.DeviceBatchIterator([dev(frames)]),
                                 'real': training.DeviceBatchIterator([(dev(real), dev(labels))])},
    optimizer={'opt_gen': loans_amd.Adam(alpha=1e-4, amsgrad=True).setup(loc),
               'opt_dis': loans_amd.Adam(alpha=1e-4, amsgrad=True).setup(dis)},
    converter=training.identity_converter, device=0)
up.update()                                                 # the sizing step
found = []


def bad(t):
    return torch.is_tensor(t) and t.is_floating_point() and t.numel() and bool(torch.isnan(t.float()).any())


fwd, bwd = core.Function.__call__, None


def checked_call(self, *inputs):
    out = fwd(self, *inputs)
    outs = out if isinstance(out, tuple) else (out,)
    for i, o in enumerate(outs):
        if bad(o.data):
            found.append(('forward', type(self).__name__, i, tuple(o.data.shape)))
            print('NaN out of forward ', type(self).__name__, 'output', i, tuple(o.data.shape), flush=True)
    return out


core.Function.__call__ = checked_call
for cls in list(core.Function.__subclasses__()) + [c for k in core.Function.__subclasses__() for c in k.__subclasses__()]:
    if 'backward' in cls.__dict__:
        def make(orig, name):
            def checked_backward(self, inputs, gys):
                gxs = orig(self, inputs, gys)
                for i, g in enumerate(gxs if isinstance(gxs, tuple) else (gxs,)):
                    if bad(g):
                        found.append(('backward', name, i, tuple(g.shape)))
                        print('NaN out of backward', name, 'input', i, tuple(g.shape), flush=True)
                return gxs
            return checked_backward
        cls.backward = make(cls.__dict__['backward'], cls.__name__)
for a in ops._step_arenas.values():
    a.poison = True
for it in range(2):
    up.update()
    torch.cuda.synchronize()
    for name, m in (('localizer', loc), ('assessor', dis)):
        g = m.arena.grad
        if bad(g):
            keys = [k for k, p in m.namedparams() if bad(p.grad_view)]
            print('step', it + 2, name, 'gradient arena has NaN in', keys[:12], flush=True)
        if bad(m.arena.data):
            print('step', it + 2, name, 'parameters have NaN', flush=True)
print('%d offenders' % len(found))
