#!/usr/bin/env python
"""A/B two builds of the kernel library on one box: runs bench.py with loans_amd._lib.LIB_PATH pointed at the given .so.
usage: ab_lib.py <path to .so> [bench.py arguments]   (development tool)"""
import os
import runpy
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
