set -e
mkdir -p gpurun_out
out=gpurun_out/r6_fixed_cost.txt
: > $out
for cin in 512 256 128 64; do
  echo "== M = 32768, N = 512, K = 9 x $cin" >> $out
  timeout -k 10 120 python tools/halo_bench.py 128 16 16 $cin 512 9,43,9,43 2>&1 | grep "ms" >> $out
done
for cin in 256 128 64; do
  echo "== M = 131072, N = 256, K = 9 x $cin" >> $out
  timeout -k 10 120 python tools/halo_bench.py 128 32 32 $cin 256 9,43,9,43 2>&1 | grep "ms" >> $out
done
cat $out
