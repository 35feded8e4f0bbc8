#!/bin/bash
# development: what bounds the res2 convolution of configs[2] on the wave-autonomous weights-stationary kernel (tile 37)?
# ablation bits of the LOANS_EXPERIMENT build: 1 no image DMA, 2 no MFMAs, 4 no fragment reads, 8 no stores, 16 four waves per CU
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/r5_wsw_ablation.txt
: > $out
for dbg in 0 1 2 4 8 16 6 9 15 17 24 25 31; do
  echo "== LOANS_HALO_DBG=$dbg" >> $out
  LOANS_HALO_DBG=$dbg timeout -k 10 120 python tools/halo_bench.py 128 128 128 64 64 37,15,12,14 >> $out 2>&1 || exit 1
done
cat $out
