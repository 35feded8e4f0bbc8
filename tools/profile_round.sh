#!/bin/bash
# The measurement set of a round, on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [fp32|b128|cfg3|r50|all]   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# Per workload: 1 the bench line (default command; it WRITES the kernel-tile table) | 2 rocprofv3 kernel trace + stats of the
# same command READING that table | 3 two PMC passes (FETCH_SIZE, WRITE_SIZE) reading it too -> HBM traffic of the
# conv-forward launches, asserted to be the launches of the timed run | 4 one SQ pass (MFMA busy) on the same kernels
set -eo pipefail
tag=${1:-r2x}
what=${2:-all}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LOANS_BENCH_RETUNE=1      # the first run of a workload tunes afresh and writes the table the other passes read
mkdir -p $out

one () {   # one <name> <B> <peak TFLOP/s> <flop per image> <bench args...>
  local name=$1 B=$2 peak=$3 fpi=$4; shift 4
  local tune=$out/${tag}_${name}_tune.json
  rm -f $tune
  # (the stdout line is the compact form; the full object -- per-layer / per-class tables -- is the detail file)
  python3 bench.py --no-secondary --tune-file $tune --detail-file $out/${tag}_${name}_bench.json "$@" > $out/${tag}_${name}_bench.log 2>&1
  echo "[$name] bench done"; cut -c1-600 $out/${tag}_${name}_bench.json
  if [ "$fpi" = auto ]; then      # algorithmic conv-forward FLOP per image from the bench line itself
    fpi=$(python3 -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['algorithmic_flop_per_step'] // $B)" $out/${tag}_${name}_bench.json)
  fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${name}_tr -- python3 bench.py --no-secondary --tune-file $tune --steps 3 --warmup 2 --no-cpu-baseline --detail-file /dev/null "$@" > $out/${tag}_${name}_trace.log 2>&1
  python3 tools/trace_summary.py $out/${tag}_${name}_tr $B $peak $fpi > $out/${tag}_${name}_trace_summary.txt
  cp $out/${tag}_${name}_tr/*/*kernel_stats.csv $out/${tag}_${name}_kernel_stats.csv
  python3 tools/class_times.py $out/${tag}_${name}_tr $out/${tag}_${name}_class_times.json "bench.py $*"
  python3 tools/step_timeline.py $out/${tag}_${name}_tr > $out/${tag}_${name}_timeline.txt
  rm -rf $out/${tag}_${name}_tr
  echo "[$name] trace done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_${name}_f -- python3 bench.py --no-secondary --tune-file $tune --steps 2 --warmup 2 --no-cpu-baseline --detail-file /dev/null "$@" > $out/${tag}_${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_${name}_w -- python3 bench.py --no-secondary --tune-file $tune --steps 2 --warmup 2 --no-cpu-baseline --detail-file /dev/null "$@" > $out/${tag}_${name}_write.log 2>&1
  python3 tools/pmc_traffic.py $out/${tag}_${name}_f $out/${tag}_${name}_w $out/${tag}_${name}_conv_fwd_hbm_traffic.json $out/${tag}_${name}_bench.json "localizer conv-forward launches of one step of: bench.py $*"
  rm -rf $out/${tag}_${name}_f $out/${tag}_${name}_w
  echo "[$name] traffic done"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_${name}_sq -- python3 bench.py --no-secondary --tune-file $tune --steps 2 --warmup 2 --no-cpu-baseline --detail-file /dev/null "$@" > $out/${tag}_${name}_sq.log 2>&1
  python3 tools/pmc_sq.py $out/${tag}_${name}_sq > $out/${tag}_${name}_conv_fwd_sq_pmc_summary.txt || echo "[$name] SQ summary failed"
  rm -rf $out/${tag}_${name}_sq
  grep "^localizer conv\|^algorithmic\|^timeline" $out/${tag}_${name}_trace_summary.txt
}

if [ "$what" = fp32 ] || [ "$what" = all ]; then
  one b256 256 157.3 4166615040
fi
if [ "$what" = b128 ] || [ "$what" = all ]; then
  one b128 128 157.3 4166615040 --batch 128
fi
if [ "$what" = cfg3 ] || [ "$what" = all ]; then
  one cfg3 128 2500 23655874560 --dtype bf16 --batch 128 --image-size 512
fi
if [ "$what" = r50 ] || [ "$what" = all ]; then
  one r50 64 2500 auto --dtype bf16 --batch 64 --image-size 512 --resnet50
fi
