#!/usr/bin/env python
"""Development: bench.py with one module-level constant of loans_amd.ops set to another value (the settled switches are plain
constants, not environment variables): ab_ops_attr.py NAME=VALUE [bench.py arguments]     e.g. POOL_ARGMAX_VALUES=False"""
import ast
import os
import runpy
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loans_amd import ops  # noqa: E402
name, _, value = sys.argv[1].partition('=')
assert hasattr(ops, name), name
setattr(ops, name, ast.literal_eval(value))
sys.argv = ['bench.py'] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name='__main__')
