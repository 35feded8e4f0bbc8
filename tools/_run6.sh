set -e
mkdir -p gpurun_out
for n in cfg3 r50; do
  cp profiles/r5_${n}_tune.json gpurun_out/ab_${n}_9.json
  python tools/ab_tile_swap.py profiles/r5_${n}_tune.json gpurun_out/ab_${n}_43.json 9 43
  python tools/ab_tile_swap.py profiles/r5_${n}_tune.json gpurun_out/ab_${n}_44.json 9 44
done
out=gpurun_out/r6_pp_instep_ab.txt
: > $out
run () {  # run <label> <tune> <args...>
  local label=$1 tune=$2; shift 2
  python3 bench.py --no-secondary --no-cpu-baseline --detail-file /dev/null --tune-file $tune "$@" 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print('$label', d['ms_per_step'], d['roofline']['conv_fwd_ms_per_step'], d['roofline']['frac'])" >> $out
}
for rep in 1 2 3; do
  for t in 9 43 44; do
    run "cfg3 tile $t" gpurun_out/ab_cfg3_$t.json --dtype bf16 --batch 128 --image-size 512 --steps 20 --warmup 5
  done
done
for rep in 1 2; do
  for t in 9 43 44; do
    run "r50 tile $t" gpurun_out/ab_r50_$t.json --dtype bf16 --batch 64 --image-size 512 --resnet50 --steps 20 --warmup 5
  done
done
cat $out
