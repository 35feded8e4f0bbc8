#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r5_bn_reverse_ab.txt
: > $out
D=$PWD/loans_amd/csrc
for shape in "128 128 128 64" "128 64 64 128" "128 32 32 256" "64 128 128 256" "64 64 64 512"; do
  for nt in "" 0; do
    for lib in $D/libloans_hip.so $D/libloans_hip_rev.so; do
      echo "== $shape | LOANS_BN_NT=${nt:-default} | $(basename $lib)" >> $out
      LOANS_BN_NT=$nt BN_BENCH_LIB=$lib python tools/bn_bench2.py $shape 2>/dev/null | grep "bn_backward" >> $out
    done
  done
done
cat $out
