#!/usr/bin/env python
"""Every launch of the LAST step of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py, in start order:
offset from the step's first kernel, duration, queue, grid, kernel -- the per-launch view behind tools/trace_summary.py's
totals (which weight gradient sits beside which BN pass, what each launch of a class takes).   (development tool)
usage: step_timeline.py <rocprofv3 output dir> [> file]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
prep = [i for i, r in enumerate(rows) if 'prep_kernel' in r['Kernel_Name'] or 'prep_dense_kernel' in r['Kernel_Name']]
seg = rows[prep[-2]:prep[-1]]


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')
    if 'at::native' in n:
        return 'torch:' + (re.search(r'(\w+Functor|\w+_kernel)', n) or re.search(r'(.{0,30})', n)).group(1)
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', n)
    return m.group(1) if m else n[:60]


t0 = min(int(r['Start_Timestamp']) for r in seg)
queues = {}
for r in sorted(seg, key=lambda r: int(r['Start_Timestamp'])):
    q = queues.setdefault(r.get('Queue_Id', '?'), len(queues))
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f %8.1f q%d %9s %5s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?'),
                                          short(r['Kernel_Name'])))
