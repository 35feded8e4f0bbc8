#!/bin/bash
# Same-box, interleaved A/B of environment switches on bench.py legs (development tool):
#   tools/ab_env.sh <out file> <reps> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
# Every round runs each setting once, in order; prints ms_per_step per run and the per-setting medians at the end.
out=$1; reps=$2; args=$3; shift 3
: > "$out"
settings=("$@")
n=${#settings[@]}
for r in $(seq 1 "$reps"); do
  i=0
  # the arms start with another one every round: the first run of a round reads ~0.07 ms slow on this pool (two identical arms in
  # profiles/r5_issue_orders_ab.txt), which a fixed order would book on one arm
  for k in $(seq 0 $((n - 1))); do
    setting=${settings[$(( (k + r - 1) % n ))]}
    i=$((i + 1))
    ms=$(env $setting python3 bench.py --no-secondary --no-cpu-baseline $args 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['roofline'].get('conv_fwd_ms_per_step', ''))")
    echo "round $r | $setting | $ms" | tee -a "$out"
  done
done
python3 - "$out" <<'PY'
import sys, collections, statistics
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = [x.strip() for x in line.split('|')]
    if len(p) == 3 and p[2]:
        d[p[1]].append(float(p[2].split()[0]))
with open(sys.argv[1], 'a') as f:
    for k, v in d.items():
        s = 'median %-60s %.3f ms over %d runs (%s)' % (k, statistics.median(v), len(v), ' '.join('%.2f' % x for x in v))
        print(s); f.write(s + '\n')
PY
