#!/usr/bin/env python
"""Summarise one rocprofv3 --pmc pass of SQ / GRBM counters over the ResNet-18 conv-forward launches of the last
bench.py step (same selection as tools/trace_summary.py): where the waves' cycles go, LDS bank conflicts, MFMA-pipe
busy fraction.  usage: pmc_sq.py <rocprof output dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'], 'grid': r['Grid_Size'], 'c': {}})
    d['c'][r['Counter_Name']] = d['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
ids = sorted(disp)
prep = [i for i in ids if 'prep_' in disp[i]['name']]
seg = [i for i in ids if prep[-2] <= i < prep[-1]]
gap = max(k for k, i in enumerate(seg) if 'gap_fwd_kernel' in disp[i]['name'])
fwd = [disp[i] for i in seg[:gap] if "igemm" in disp[i]["name"] or "stem7" in disp[i]["name"] or "halo16" in disp[i]["name"] or "ws8_kernel" in disp[i]["name"] or "wsw_kernel" in disp[i]["name"] or ("pw16_" in disp[i]["name"] and "pack_batch" not in disp[i]["name"])]
tot = collections.Counter()
for d in fwd:
    tot.update(d['c'])
print('%d conv-forward igemm launches of one step' % len(fwd))
for k in sorted(tot):
    print('  %-28s %.4g' % (k, tot[k]))
wc = tot.get('SQ_WAVE_CYCLES', 0)
if wc:
    print('wave cycles: parked on s_waitcnt/barrier %.1f %%, issue-stalled %.1f %% (of which LDS issue %.1f %%), issuing %.1f %%'
          % (100 * tot['SQ_WAIT_ANY'] / wc, 100 * tot['SQ_WAIT_INST_ANY'] / wc, 100 * tot.get('SQ_WAIT_INST_LDS', 0) / wc,
             100 * tot['SQ_ACTIVE_INST_ANY'] / wc))
if tot.get('SQ_LDS_IDX_ACTIVE'):
    print('LDS: bank-conflict cycles / LDS-array cycles = %.2f %%' % (100 * tot['SQ_LDS_BANK_CONFLICT'] / tot['SQ_LDS_IDX_ACTIVE']))
if tot.get('GRBM_GUI_ACTIVE') and tot.get('SQ_VALU_MFMA_BUSY_CYCLES'):
    simds, xcds = 256 * 4, 8
    cycles = tot['GRBM_GUI_ACTIVE'] / xcds          # the counter is summed over the 8 XCDs
    print('GPU-active cycles of these launches: %.4g (GRBM_GUI_ACTIVE / %d XCDs)' % (cycles, xcds))
    print('MFMA pipe busy: %.1f %% of SIMD-cycles (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x %d SIMDs); 64 busy cycles per '
          'v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16)' % (100 * tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * simds), simds))
