#!/usr/bin/env python
"""Development: bench.py with the tuned pixel-slice count of every bf16 weight gradient scaled (ops.WGRAD_SPLIT_SCALE): do the
main stream's kernels gain more from free compute units than the weight gradients lose?
usage: ab_wgrad_scale.py <scale> [bench.py arguments]"""
import os
import runpy
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import ops  # noqa: E402
ops.WGRAD_SPLIT_SCALE = float(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
