#!/usr/bin/env python
"""One rocprofv3 --pmc pass (SQ / GRBM counters) over a bench.py leg: per kernel of the LAST step, where its waves' cycles go with the step's own operands and
cache state (counter collection serialises the launches, so the two streams do not overlap here and launches of a few microseconds
are inflated by the counter reads) -- parked on s_waitcnt / barriers, issue-stalled, issuing -- MFMA-pipe and LDS-array busy.
usage: pmc_step_table.py <rocprof output dir>"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'], 'c': collections.Counter()})
    d['c'][r['Counter_Name']] += float(r['Counter_Value'])
ids = sorted(disp)
prep = [i for i in ids if 'prep_' in disp[i]['name']]
seg = [i for i in ids if prep[-2] <= i < prep[-1]] if len(prep) >= 2 else ids


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:58]


tab = collections.OrderedDict()
for i in seg:
    t = tab.setdefault(short(disp[i]['name']), {'n': 0, 'c': collections.Counter()})
    t['n'] += 1
    t['c'].update(disp[i]['c'])
rows = sorted(tab.items(), key=lambda kv: -kv[1]['c'].get('GRBM_GUI_ACTIVE', 0))
print('one step: %d launches of %d kernels; GPU cycles = GRBM_GUI_ACTIVE / 8 XCDs, summed over the launches (serialised by the counter collection; microsecond launches are inflated)' % (len(seg), len(tab)))
print('%-58s %4s %10s %7s %7s %7s %6s %6s' % ('kernel', 'n', 'GPU cycles', 'parked', 'stalled', 'issuing', 'MFMA', 'LDS'))
for name, t in rows[:40]:
    c = t['c']
    wc = c.get('SQ_WAVE_CYCLES', 0) or 1
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8 or 1
    print('%-58s %4d %10.3g %6.1f%% %6.1f%% %6.1f%% %5.1f%% %5.1f%%' % (
        name, t['n'], cyc, 100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_WAIT_INST_ANY'] / wc, 100 * c['SQ_ACTIVE_INST_ANY'] / wc,
        100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), 100 * c['SQ_LDS_IDX_ACTIVE'] / (cyc * 256)))
