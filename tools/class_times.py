#!/usr/bin/env python
"""Kernel time per class of the LAST step of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py -- the measured side of
bench.py's `roofline.whole_step.binding` (the algorithmic side is counted live by loans_amd/ops.py: CLASS_COUNT, same class names).
usage: class_times.py <rocprofv3 output dir> <out.json> [note]"""
import collections
import csv
import glob
import json
import sys

CLASSES = (          # first match wins; matched against the demangled-ish kernel name
    ('wgrad', ('wgrad', 'fold_slabs')),
    ('stem', ('bn_relu_maxpool', 'pool_bn_bwd', 'maxpool_relu_bwd', 'stem_bwd16', 'fold_replicas')),
    ('conv', ('igemm', 'halo16', 'ws8_kernel', 'wsw_kernel', 'pw16_kernel', 'pw16_k256', 'stem7')),
    ('bn_bwd', ('bn_bwd',)),
    ('bn_fwd', ('bn_apply', 'bn_finalize', 'bn_eval')),
    ('crop', ('crop_dgrad', 'crop_pack', 'st_sampler', 'st_grid', 'dgrad_c4')),
    ('optimizer', ('adam', 'cast_bf16', 'repack', 'pw16_pack', 'at::native', 'rocclr', 'mul_kernel')),
    ('heads', ('',)),
)


def classify(name):
    for cls, keys in CLASSES:
        if any(k in name for k in keys):
            return cls
    return 'heads'


def main():
    f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
    rows = list(csv.DictReader(open(f)))
    prep = [i for i, r in enumerate(rows) if 'prep_kernel' in r['Kernel_Name'] or 'prep_dense_kernel' in r['Kernel_Name']]
    seg = rows[prep[-2]:prep[-1]]
    ms, n = collections.Counter(), collections.Counter()
    for r in seg:
        c = classify(r['Kernel_Name'])
        ms[c] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        n[c] += 1
    span = (max(int(r['End_Timestamp']) for r in seg) - min(int(r['Start_Timestamp']) for r in seg)) / 1e6
    out = {"what": "kernel time per class over ONE step (the last of the trace), ms; classes overlap in time (streams)",
           "note": sys.argv[3] if len(sys.argv) > 3 else "", "step_span_ms": round(span, 3),
           "kernel_ms": {k: round(v, 4) for k, v in sorted(ms.items())}, "launches": dict(sorted(n.items())),
           "kernel_ms_total": round(sum(ms.values()), 3)}
    json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
