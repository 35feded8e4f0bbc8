#!/bin/bash
# Ablation builds of csrc/wgrad_halo_bf16.hip (compile-time LOANS_WGH_DBG bits: 1 no atomics, 2 no MFMAs, 4 no global loads
# after the first tile, 8 no fragment reads / MFMAs), each linked with the library's other objects and timed on the res2 / res4
# shapes of configs[2].  Run on the GPU box from the repo root: bash tools/wgrad_ablate.sh "0 1 2 4 8"
set -e
cd loans_amd/csrc
others=$(ls *.o | grep -v wgrad_halo_bf16)
for d in ${1:-0 1 2 4 8 9}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -DLOANS_WGH_DBG=$d $EXTRA -c wgrad_halo_bf16.hip -o /tmp/wgh_$d.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libloans_wgh_$d.so $others /tmp/wgh_$d.o
  echo "== LOANS_WGH_DBG=$d"
  (cd ../.. && LOANS_BENCH_LIB=/tmp/libloans_wgh_$d.so python tools/wgrad_bench.py --layers ${2:-res2,res4} 2>&1 | grep "^      " | sed -e 's/.*\(38\/0\)/\1/')
done
