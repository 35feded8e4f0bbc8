#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_storage.py -x -q -m gpu > gpurun_out/try_tests.log 2>&1; rc=$?; echo "tests exit $rc" | tee -a gpurun_out/try_tests.log; tail -3 gpurun_out/try_tests.log
[ $rc -eq 0 ] || exit 1
out=gpurun_out/r5_wsw_regstage_ab.txt
: > $out
D=loans_amd/csrc
for round in 1 2 3; do
  for lib in $D/libloans_hip_swz.so $D/libloans_hip.so; do
    echo "== round $round $(basename $lib)" >> $out
    HALO_BENCH_LIB=$lib python tools/halo_bench.py 128 128 128 64 64 37,15,12 2>/dev/null >> $out
    HALO_BENCH_LIB=$lib python tools/halo_bench.py 64 128 128 64 64 37 2>/dev/null >> $out
  done
done
cat $out | tail -30
bash tools/ab_libs.sh $out 3 "--image-size 512 --batch 128 --dtype bf16" $D/libloans_hip_swz.so $D/libloans_hip_rs0.so $D/libloans_hip.so
bash tools/ab_libs.sh $out 2 "--image-size 512 --batch 64 --dtype bf16 --resnet50" $D/libloans_hip_swz.so $D/libloans_hip_rs0.so $D/libloans_hip.so
