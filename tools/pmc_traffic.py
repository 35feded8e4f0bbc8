#!/usr/bin/env python
"""HBM bytes of the 21 ResNet-18 conv-forward launches of one bench.py step, from two separate rocprofv3 PMC passes
(`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`) of the same bench.py command -> profiles/r1_conv_fwd_hbm_traffic.json.

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>
Counter units are KB; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads,
MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import sys


def conv_fwd_sum(path, counter):
    f = glob.glob(path + '/*/*counter_collection.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    prep = [i for i, r in enumerate(rows) if 'prep_' in r['Kernel_Name']]
    seg = rows[prep[-2]:prep[-1]]
    gap = next(i for i, r in enumerate(seg) if 'gap_fwd_kernel' in r['Kernel_Name'])      # end of the backbone forward
    fwd = [r for r in seg[:gap] if 'igemm' in r['Kernel_Name'] or 'stem7' in r['Kernel_Name']]      # 21 convs; a LOANS_TILE_SPLIT conv is two launches
    assert len(fwd) >= 17          # 21 convs; BasicA pairs are one launch, a LOANS_TILE_SPLIT conv is two
    return sum(float(r['Counter_Value']) for r in fwd) * 1024.0, len(fwd)


fetch, n = conv_fwd_sum(sys.argv[1], 'FETCH_SIZE')
fetch *= 2.0
write, n2 = conv_fwd_sum(sys.argv[2], 'WRITE_SIZE')
assert n == n2
out = {
    "what": "HBM bytes of the ResNet-18 conv-forward igemm launches (21 convs; a BasicA pair is one launch) of one bench.py step (B=256, 224^2, fp32)",
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `python3 bench.py --steps 2 "
              "--warmup 2 --no-cpu-baseline`; KB -> bytes x1024; FETCH_SIZE doubled (gfx950 reports half the bytes of "
              "wide coalesced reads, MI355X_MICROARCH.md HBM section); tools/pmc_traffic.py",
    "fetch_bytes_corrected": fetch, "write_bytes": write, "total_bytes_per_step": fetch + write,
    "bytes_per_launch": (fetch + write) / n, "launches": n, "convs": 21,
}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out))
