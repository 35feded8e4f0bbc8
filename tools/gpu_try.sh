#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/try_early_chain.txt
: > $out
for leg in "--image-size 512 --batch 128 --dtype bf16" "--image-size 512 --batch 64 --dtype bf16 --resnet50"; do
  for round in 1 2 3; do
    for v in 0 1; do
      ms=$(LOANS_EARLY_CHAIN=$v python3 bench.py --no-secondary --no-cpu-baseline $leg 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['roofline'].get('conv_fwd_ms_per_step', ''))")
      echo "round $round | $leg | LOANS_EARLY_CHAIN=$v | $ms" | tee -a $out
    done
  done
done
