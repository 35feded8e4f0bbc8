#!/usr/bin/env python
"""HBM bytes of the localizer's conv-forward launches of one bench.py step, from two separate rocprofv3 PMC passes
(`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`) of the same bench.py command run on the SAME tile table as the timed run
(bench.py --tune-file) -> profiles/<tag>_conv_fwd_hbm_traffic.json, which bench.py reads for `roofline.traffic`.

usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [<bench line .json>] [<what>]
Counter units are KB; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads,
MI355X_MICROARCH.md, HBM section).  With a bench line, the number of launches found in each PMC pass must equal the
`roofline.launches_per_step` of the timed run: a pass that ran other kernels does not describe it."""
import csv
import glob
import json
import sys


def conv_fwd_rows(path, counter):
    f = glob.glob(path + '/*/*counter_collection.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    prep = [i for i, r in enumerate(rows) if 'prep_' in r['Kernel_Name']]
    seg = rows[prep[-2]:prep[-1]]                                                           # the last whole step
    gap = max(i for i, r in enumerate(seg) if 'gap_fwd_kernel' in r['Kernel_Name'])       # end of the localizer's conv forward (ResNet-50: pool5, then the one behind res6 / res7)
    # what bench.py brackets: implicit-GEMM launches (a BasicA pair is one, a two-shape split two), the direct stem kernel,
    # the finalize pass of a fine-tail / split-K conv
    return [r for r in seg[:gap] if 'igemm' in r['Kernel_Name'] or 'stem7' in r['Kernel_Name'] or 'halo16' in r['Kernel_Name'] or 'ws8_kernel' in r['Kernel_Name'] or 'wsw_kernel' in r['Kernel_Name'] or ('pw16_' in r['Kernel_Name'] and 'pack_batch' not in r['Kernel_Name'])]


def main():
    fdir, wdir, out_path = sys.argv[1:4]
    bench = json.load(open(sys.argv[4])) if len(sys.argv) > 4 and sys.argv[4] else None
    what = sys.argv[5] if len(sys.argv) > 5 else "localizer conv-forward launches of one bench.py step"
    fr, wr = conv_fwd_rows(fdir, 'FETCH_SIZE'), conv_fwd_rows(wdir, 'WRITE_SIZE')
    assert len(fr) == len(wr), (len(fr), len(wr))
    assert [r['Kernel_Name'] for r in fr] == [r['Kernel_Name'] for r in wr], 'the two PMC passes ran different kernels'
    n = len(fr)
    if bench is not None:
        want = bench['roofline']['launches_per_step']
        assert n == want, 'PMC passes saw %d conv-forward launches per step, the timed run had %d: not the same kernels' % (n, want)
    fetch = 2.0 * 1024.0 * sum(float(r['Counter_Value']) for r in fr)
    write = 1024.0 * sum(float(r['Counter_Value']) for r in wr)
    per_kernel = {}
    for a, b in zip(fr, wr):
        k = a['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:80]
        e = per_kernel.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += 2.0 * 1024.0 * float(a['Counter_Value']) + 1024.0 * float(b['Counter_Value'])
    out = {
        "what": "HBM bytes of the " + what,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of the bench.py command, both reading "
                  "the tile table the timed run wrote (--tune-file); KB -> bytes x1024; FETCH_SIZE doubled (gfx950 reports half "
                  "the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); tools/pmc_traffic.py",
        "fetch_bytes_corrected": fetch, "write_bytes": write, "total_bytes_per_step": fetch + write,
        "bytes_per_launch": (fetch + write) / n, "launches": n,
        "launches_of_timed_run": None if bench is None else bench['roofline']['launches_per_step'],
        "by_kernel": {k: {"launches": v[0], "bytes": v[1]} for k, v in sorted(per_kernel.items())},
    }
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
