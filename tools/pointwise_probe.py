#!/usr/bin/env python
"""Development probe: every tile on offer for the 1x1 convolutions of the ResNet-50 localizer (configs[4] per GPU: B = 64 of
512 x 512 frames), forward with BN statistics and data gradient, against the layer's HBM bound."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                # noqa: E402
if os.environ.get('LOANS_EXP_LIB'):
    from loans_amd import _lib
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libloans_hip_exp.so')
from loans_amd import ops   # noqa: E402

ops.set_compute_dtype('bf16'); ops.set_storage_dtype('bf16')
ops.TUNE_VERBOSE = True
B = 64
for name, Cin, HW, Cout in [('res2 expand', 64, 128, 256), ('res2 reduce', 256, 128, 64), ('res3 expand', 128, 64, 512),
                            ('res3 reduce', 512, 64, 128), ('res4 expand', 256, 32, 1024), ('res4 reduce', 1024, 32, 256),
                            ('res5 expand', 512, 16, 2048), ('res5 reduce', 2048, 16, 512)]:
    geo = ops.ConvGeometry(B, HW, HW, Cin, Cout, 1, 1, 0)
    x = torch.randn((B, HW, HW, Cin), device='cuda').to(torch.bfloat16)
    w = torch.randn((Cout, 1, 1, Cin), device='cuda') * 0.05
    gy = torch.randn((B, HW, HW, Cout), device='cuda').to(torch.bfloat16)
    mb = B * HW * HW * (Cin + Cout) * 2 / 1e6
    print('%s: %d -> %d at %d^2: %.0f MB = %.3f ms at 5.5 TB/s' % (name, Cin, Cout, HW, mb, mb / 5.5e6), flush=True)
    ops.conv_fprop(x, w, geo, stats=ops.stats_buffer(Cout, x.device))
    ops.conv_dgrad(gy, w, geo)
