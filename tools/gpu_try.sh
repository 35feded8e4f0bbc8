#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_storage.py -x -q -m gpu -k "weight_gradient or fold" > gpurun_out/try_tests.log 2>&1; rc=$?; echo "tests exit $rc" | tee -a gpurun_out/try_tests.log; tail -3 gpurun_out/try_tests.log
[ $rc -eq 0 ] || exit 1
out=gpurun_out/r5_wgrad_pipelined_ab.txt
: > $out
D=$PWD/loans_amd/csrc
for round in 1 2; do
  for lib in $D/libloans_hip_base.so $D/libloans_hip.so; do
    echo "== round $round $(basename $lib)" >> $out
    LOANS_BENCH_LIB=$lib python tools/wgrad_bench.py --layers res2,res3,res4,res5,r50_res2 --reps 5 2>/dev/null | grep "halo best" | cut -c1-200 >> $out
  done
done
cat $out
bash tools/ab_libs.sh $out 3 "--image-size 512 --batch 128 --dtype bf16" $D/libloans_hip_base.so $D/libloans_hip.so
bash tools/ab_libs.sh $out 3 "--image-size 512 --batch 64 --dtype bf16 --resnet50" $D/libloans_hip_base.so $D/libloans_hip.so
