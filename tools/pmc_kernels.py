#!/usr/bin/env python
"""Per-kernel-name summary of one rocprofv3 --pmc pass of SQ / GRBM counters (any program): where the waves' cycles
go, LDS bank conflicts, MFMA-pipe busy fraction.  usage: pmc_kernels.py <rocprof output dir> [name substring]"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
want = sys.argv[2] if len(sys.argv) > 2 else ''
per = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if want not in n or 'at::native' in n:
        continue
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', n.replace('(anonymous namespace)::', ''))
    k = m.group(1) if m else n[:40]
    c = per.setdefault(k, collections.Counter())
    c[r['Counter_Name']] += float(r['Counter_Value'])
    c['_rows'] += 1
for k, t in per.items():
    wc = t.get('SQ_WAVE_CYCLES', 0)
    line = '%-44s' % k
    if wc:
        line += ' parked %4.1f%% issue-stall %4.1f%% (LDS %4.1f%%) issuing %4.1f%%' % (
            100 * t['SQ_WAIT_ANY'] / wc, 100 * t['SQ_WAIT_INST_ANY'] / wc, 100 * t.get('SQ_WAIT_INST_LDS', 0) / wc,
            100 * t['SQ_ACTIVE_INST_ANY'] / wc)
    if t.get('SQ_LDS_IDX_ACTIVE'):
        line += ' | LDS conflict %4.1f%%' % (100 * t['SQ_LDS_BANK_CONFLICT'] / t['SQ_LDS_IDX_ACTIVE'])
    if t.get('GRBM_GUI_ACTIVE') and t.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        cycles = t['GRBM_GUI_ACTIVE'] / 8
        line += ' | MFMA busy %4.1f%% | LDS busy %4.1f%%' % (100 * t['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024),
                                                         100 * t.get('SQ_LDS_IDX_ACTIVE', 0) / (cycles * 256))
    print(line)
