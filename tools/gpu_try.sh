#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
out=gpurun_out/try_halo_nowait.txt
: > $out
for dbg in 0 32 0 32; do
  echo "== LOANS_HALO_DBG=$dbg" >> $out
  LOANS_HALO_DBG=$dbg python tools/halo_bench.py 128 64 64 128 128 36,11 2>/dev/null >> $out
  LOANS_HALO_DBG=$dbg python tools/halo_bench.py 64 64 64 128 128 11,36 2>/dev/null >> $out
done
cat $out
