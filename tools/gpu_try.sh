#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/try_f32_split_scale.txt
: > $out
for round in 1 2 3; do
  for v in 1.0 0.75 0.5; do
    ms=$(python3 tools/ab_ops_attr.py WGRAD_SPLIT_SCALE_F32=$v --no-secondary --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['roofline'].get('conv_fwd_ms_per_step', ''))")
    echo "round $round | fp32 default | WGRAD_SPLIT_SCALE_F32=$v | $ms" | tee -a $out
  done
done
