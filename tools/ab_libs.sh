#!/bin/bash
# Same-box, interleaved A/B of two builds of the kernel library on bench.py legs (development tool):
#   tools/ab_libs.sh <out file> <reps> "<bench args>" <lib A .so> <lib B .so> ...
out=$1; reps=$2; args=$3; shift 3
for r in $(seq 1 "$reps"); do
  for lib in "$@"; do
    ms=$(python3 tools/ab_lib.py $lib --no-secondary --no-cpu-baseline $args 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['roofline'].get('conv_fwd_ms_per_step', ''))")
    echo "round $r | $args | $(basename $lib) | $ms" | tee -a "$out"
  done
done
python3 - "$out" <<'PY'
import sys, collections, statistics
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = [x.strip() for x in line.split('|')]
    if len(p) == 4 and p[3] and p[0].startswith('round'):
        d[(p[1], p[2])].append([float(v) for v in p[3].split()])
with open(sys.argv[1], 'a') as f:
    for k, v in d.items():
        ms = [x[0] for x in v]
        cf = [x[1] for x in v if len(x) > 1]
        s = 'median %-44s %-24s %.3f ms/step, conv forward %.3f ms over %d runs (%s)' % (k[0], k[1], statistics.median(ms), statistics.median(cf) if cf else 0, len(v), ' '.join('%.2f' % x for x in ms))
        print(s); f.write(s + '\n')
PY
