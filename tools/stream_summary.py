#!/usr/bin/env python
"""Per-queue view of one step of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: which HIP stream (queue) is
busy how long, and with what -- the main stream is the step's critical path, the weight-gradient and assessor-chain streams
run beside it.  usage: stream_summary.py <rocprof output dir>"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
names = [r['Kernel_Name'] for r in rows]
prep = [i for i, n in enumerate(names) if 'prep_kernel' in n or 'prep_dense_kernel' in n]
seg = rows[prep[-2]:prep[-1]]
short = lambda n: (re.search(r'(\w+_kernel(<[^>]*>)?)', n.replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')) or re.search(r'(.{0,40})', n)).group(1)   # noqa: E731
t0 = min(int(r['Start_Timestamp']) for r in seg)
t1 = max(int(r['End_Timestamp']) for r in seg)
print('step: %.2f ms' % ((t1 - t0) / 1e6))
qkey = 'Queue_Id' if 'Queue_Id' in seg[0] else 'Stream_Id'
by_q = collections.defaultdict(list)
for r in seg:
    by_q[r[qkey]].append(r)
for q, rs in sorted(by_q.items(), key=lambda kv: -sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in kv[1])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6
    first = (min(int(r['Start_Timestamp']) for r in rs) - t0) / 1e6
    last = (max(int(r['End_Timestamp']) for r in rs) - t0) / 1e6
    print('\nqueue %s: %d kernels, busy %.2f ms, active from +%.2f to +%.2f ms' % (q, len(rs), busy, first, last))
    tot, cnt = collections.Counter(), collections.Counter()
    for r in rs:
        k = 'torch fill/copy' if 'at::native' in r['Kernel_Name'] else short(r['Kernel_Name'])
        tot[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        cnt[k] += 1
    for k, v in tot.most_common(14):
        print('  %9.1f us %4d x  avg %7.1f  %s' % (v, cnt[k], v / cnt[k], k))
