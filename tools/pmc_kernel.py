#!/usr/bin/env python
"""One rocprofv3 --pmc pass (SQ / GRBM counters), summed over the dispatches whose kernel name contains a substring: where the
waves' cycles go, LDS-array busy and conflict cycles, MFMA-pipe busy.  usage: pmc_kernel.py <rocprof output dir> <substring>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
want = sys.argv[2]
tot, n, name = collections.Counter(), set(), None
for r in csv.DictReader(open(f)):
    if want in r['Kernel_Name']:
        tot[r['Counter_Name']] += float(r['Counter_Value'])
        n.add(r['Dispatch_Id'])
        name = r['Kernel_Name']
print('%d dispatches of %s' % (len(n), (name or want)[:100]))
for k in sorted(tot):
    print('  %-28s %.4g' % (k, tot[k]))
wc = tot.get('SQ_WAVE_CYCLES', 0)
if wc:
    print('wave cycles: parked on s_waitcnt / barrier %.1f %%, issue-stalled %.1f %% (LDS issue %.1f %%), issuing %.1f %%'
          % (100 * tot['SQ_WAIT_ANY'] / wc, 100 * tot['SQ_WAIT_INST_ANY'] / wc, 100 * tot.get('SQ_WAIT_INST_LDS', 0) / wc,
             100 * tot['SQ_ACTIVE_INST_ANY'] / wc))
if tot.get('GRBM_GUI_ACTIVE'):
    cycles = tot['GRBM_GUI_ACTIVE'] / 8          # the counter is summed over the 8 XCDs
    print('GPU-active cycles: %.4g per dispatch' % (cycles / max(len(n), 1)))
    if tot.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        print('MFMA pipe busy: %.1f %% of SIMD-cycles' % (100 * tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024)))
    if tot.get('SQ_LDS_IDX_ACTIVE'):
        # (in cycles: x 1 reproduces the byte counts of the kernels' fragment reads at ds_read_b128's 4 cycles per KiB)
        print('LDS array busy: %.1f %% of CU-cycles (SQ_LDS_IDX_ACTIVE / (cycles x 256)); bank-conflict cycles %.1f %% of those'
              % (100 * tot['SQ_LDS_IDX_ACTIVE'] / (cycles * 256), 100 * tot.get('SQ_LDS_BANK_CONFLICT', 0) / tot['SQ_LDS_IDX_ACTIVE']))
