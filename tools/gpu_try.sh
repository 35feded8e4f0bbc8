#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/try_stem_wgrad_ab.txt
: > $out
for r in 1 2 3; do
  for leg in "--dtype bf16 --batch 128 --image-size 512" "--dtype bf16 --batch 64 --image-size 512 --resnet50"; do
    for v in True False; do
      ms=$(python3 tools/ab_ops_attr.py STEM_WGRAD16_DIRECT=$v --no-secondary --no-cpu-baseline $leg 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'])")
      echo "round $r | $leg | STEM_WGRAD16_DIRECT=$v | $ms" | tee -a $out
    done
  done
done
