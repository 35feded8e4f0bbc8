#!/usr/bin/env python
"""Summarise a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: per-kernel totals of the LAST
step and the 21 ResNet-18 conv-forward launches (the ones bench.py's `roofline` times with HIP events)."""
import collections
import csv
import glob
import re
import sys

path = sys.argv[1]
f = glob.glob(path + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
names = [r['Kernel_Name'] for r in rows]
prep = [i for i, n in enumerate(names) if 'prep_kernel' in n or 'prep_dense_kernel' in n]
seg = rows[prep[-2]:prep[-1]]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3     # noqa: E731
short = lambda n: (re.search(r'(\w+_kernel(<[^>]*>)?)', n.replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')) or re.search(r'(.{0,40})', n)).group(1)   # noqa: E731
tot, cnt = collections.Counter(), collections.Counter()
for r in seg:
    k = 'torch fill/copy' if 'at::native' in r['Kernel_Name'] else short(r['Kernel_Name'])
    tot[k] += dur(r)
    cnt[k] += 1
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
print('one step (prep_kernel to prep_kernel): span %.1f us, sum of kernel durations %.1f us (two streams overlap)' % (span, sum(tot.values())))
for k, v in tot.most_common(30):
    print('%10.1f us %5d launches  avg %9.1f us  %s' % (v, cnt[k], v / cnt[k], k))
gap = max(i for i, r in enumerate(seg) if 'gap_fwd_kernel' in r['Kernel_Name'])       # end of the localizer's conv forward (ResNet-50: pool5, then the one behind res6 / res7)
fwd = [r for r in seg[:gap] if 'igemm' in r['Kernel_Name'] or 'stem7' in r['Kernel_Name'] or 'halo16' in r['Kernel_Name'] or 'ws8_kernel' in r['Kernel_Name'] or 'wsw_kernel' in r['Kernel_Name'] or ('pw16_' in r['Kernel_Name'] and 'pack_batch' not in r['Kernel_Name'])]     # the backbone's convs; a LOANS_TILE_SPLIT conv is two launches
t = sum(dur(r) for r in fwd)
# optional: batch, MFMA peak (TFLOP/s) and algorithmic conv-forward FLOP per image of the run (defaults: configs[1], fp32)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
peak = float(sys.argv[3]) if len(sys.argv) > 3 else 157.3
per_image = int(sys.argv[4]) if len(sys.argv) > 4 else 4166615040
flop = B * per_image
print('\nlocalizer conv forward: %d conv launches (implicit GEMM, direct stem, split-K finalize), %.1f us total, avg %.1f us per launch' % (len(fwd), t, t / len(fwd)))
print('algorithmic FLOP per step %d -> %.2f TFLOP/s = %.1f %% of the %.1f TFLOP/s MFMA peak' % (flop, flop / t / 1e6, flop / t / 1e6 / peak * 100, peak))
for r in fwd:
    print('   %-40s grid %9s  %8.1f us' % (short(r['Kernel_Name']), r['Grid_Size_X'], dur(r)))


def union(iv):
    iv = sorted(iv)
    out, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                out += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return out + (ce - cs if cs is not None else 0)


iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in seg]
mf = [(s, e) for s, e, n in iv if 'igemm' in n or 'wgrad' in n or 'stem7' in n or 'halo16' in n or 'ws8_kernel' in n or 'wsw_kernel' in n or ('pw16_k' in n) or 'crop_dgrad' in n]
al = [(s, e) for s, e, n in iv]
t0, t1 = min(s for s, _ in al), max(e for _, e in al)
print('\ntimeline of the step: %.2f ms; some kernel running %.2f ms (%.1f %%); an MFMA GEMM running %.2f ms (%.1f %%); '
      'only non-GEMM kernels running %.2f ms; idle %.2f ms'
      % ((t1 - t0) / 1e6, union(al) / 1e6, union(al) / (t1 - t0) * 100, union(mf) / 1e6, union(mf) / (t1 - t0) * 100,
         (union(al) - union(mf)) / 1e6, (t1 - t0 - union(al)) / 1e6))

# where no GEMM runs: the stretches of the step (>= 30 us) between MFMA kernels, with what runs there instead
mf.sort()
merged = []
for s, e in mf:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
holes = [(merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1)]
holes = [(t0, merged[0][0])] + holes + [(merged[-1][1], t1)]
print('\nGEMM-free stretches of the step (>= 30 us): offset, length, kernels running inside')
for s, e in holes:
    if e - s < 30000:
        continue
    inside = collections.Counter()
    for ks, ke, n in iv:
        ov = min(e, ke) - max(s, ks)
        if ov > 0 and not ('igemm' in n or 'wgrad' in n or 'stem7' in n or 'halo16' in n or 'ws8_kernel' in n or 'wsw_kernel' in n or ('pw16_k' in n) or 'crop_dgrad' in n):
            inside['torch fill/copy' if 'at::native' in n else short(n)] += ov
    print('  +%8.1f us  %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3,
                                          ', '.join('%s %.0f' % (k, v / 1e3) for k, v in inside.most_common(4))))
