#!/usr/bin/env python
"""The reference's default batch (-b 16) as a hipGraph captured WITH the step's side streams (fork / join edges between graph nodes)
and on ONE stream (a dependent chain of nodes) -- development probe behind profiles/r5_b16_graph_nodes.txt.
usage: b16_graph_streams.py <0|1>   (1 = side streams inside the capture, the default of the product)"""
import os
import runpy
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import ops  # noqa: E402
ops.CAPTURE_STREAMS = sys.argv[1] != '0'
sys.argv = ['bench.py', '--batch', '16', '--graph', '--no-secondary', '--no-cpu-baseline', '--steps', '30', '--warmup', '3']
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
