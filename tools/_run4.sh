set -e
mkdir -p gpurun_out
out=gpurun_out/r6_pp_ablation.txt
: > $out
for dbg in 0 8 16 24 32 40 48 64 72; do
  echo "== LOANS_DBG=$dbg (8 no DMA in loop, 16 no fragment reads, 32 no MFMAs, 64 cache-hot A rows)" >> $out
  LOANS_DBG=$dbg HALO_BENCH_LIB=loans_amd/csrc/libloans_hip_exp.so timeout -k 10 120 python tools/halo_bench.py 128 32 32 256 256 43 2>&1 | grep "ms" >> $out
  LOANS_DBG=$dbg HALO_BENCH_LIB=loans_amd/csrc/libloans_hip_exp.so timeout -k 10 120 python tools/halo_bench.py 128 16 16 512 512 43 2>&1 | grep "ms" >> $out
done
echo "== tile 9, LOANS_DBG=0 / 4 (cache-hot gathers)" >> $out
for dbg in 0 4; do
  LOANS_DBG=$dbg HALO_BENCH_LIB=loans_amd/csrc/libloans_hip_exp.so timeout -k 10 120 python tools/halo_bench.py 128 32 32 256 256 9 2>&1 | grep "ms" >> $out
done
cat $out
