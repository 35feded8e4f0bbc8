#!/usr/bin/env python
"""Development: a copy of a tile table with one igemm tile id replaced by another in every non-weight-gradient mode
(same-box in-step A/B of two tile forms: bench.py --tune-file A vs B).  usage: ab_tile_swap.py <in.json> <out.json> <from> <to>"""
import json
import sys
src, dst, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
d = json.load(open(src))
n = 0
for key, modes in d['entries'].items():
    for m, t in modes.items():
        if 'wgrad' not in m and t == a:
            modes[m] = b
            n += 1
json.dump(d, open(dst, 'w'))
print('%s: %d entries %d -> %d' % (dst, n, a, b))
