#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bf16_storage.py tests/test_gpu_model.py -x -q -m gpu -k "stem or pool or update_core or trajectory or region" > gpurun_out/try_tests.log 2>&1; rc=$?; echo "tests exit $rc" | tee -a gpurun_out/try_tests.log; tail -3 gpurun_out/try_tests.log
[ $rc -eq 0 ] || exit 1
out=gpurun_out/r5_pool_argmax_values_ab.txt
: > $out
for leg in "--image-size 512 --batch 128 --dtype bf16" "--image-size 512 --batch 64 --dtype bf16 --resnet50" ""; do
  for round in 1 2 3; do
    for v in False True; do
      ms=$(python3 tools/ab_ops_attr.py POOL_ARGMAX_VALUES=$v --no-secondary --no-cpu-baseline $leg 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['roofline'].get('conv_fwd_ms_per_step', ''))")
      echo "round $round | ${leg:-fp32 default} | POOL_ARGMAX_VALUES=$v | $ms" | tee -a $out
    done
  done
done
