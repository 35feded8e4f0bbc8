#!/bin/bash
# The measurement set of a round, on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>       -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# 1 bench line (default command) | 2 rocprofv3 kernel trace + stats of the same command | 3 PMC passes (HBM traffic of the
# conv-forward launches: FETCH_SIZE and WRITE_SIZE in separate runs) | 4 config 3 (bf16 storage) bench line + trace
set -eo pipefail
tag=${1:-r1x}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $out/${tag}_bench.log 2>&1
grep '^{' $out/${tag}_bench.log | tail -1 > $out/${tag}_bench_b256.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_tr -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/${tag}_trace.log 2>&1
python3 tools/trace_summary.py $out/${tag}_tr > $out/${tag}_bench_b256_trace_summary.txt
cp $out/${tag}_tr/*/*kernel_stats.csv $out/${tag}_bench_b256_kernel_stats.csv
rm -rf $out/${tag}_tr
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_f -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline > $out/${tag}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_w -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline > $out/${tag}_write.log 2>&1
python3 tools/pmc_traffic.py $out/${tag}_f $out/${tag}_w $out/${tag}_conv_fwd_hbm_traffic.json
rm -rf $out/${tag}_f $out/${tag}_w
python3 bench.py --dtype bf16 --batch 128 --image-size 512 --steps 20 > $out/${tag}_cfg3_bench.log 2>&1
grep '^{' $out/${tag}_cfg3_bench.log | tail -1 > $out/${tag}_cfg3_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_tr3 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --dtype bf16 --batch 128 --image-size 512 > $out/${tag}_cfg3_trace.log 2>&1
python3 tools/trace_summary.py $out/${tag}_tr3 128 2500 23655874560 > $out/${tag}_cfg3_trace_summary.txt
cp $out/${tag}_tr3/*/*kernel_stats.csv $out/${tag}_cfg3_kernel_stats.csv
rm -rf $out/${tag}_tr3
cat $out/${tag}_bench_b256.json
cat $out/${tag}_cfg3_bench.json
grep "^localizer conv\|^algorithmic\|^timeline" $out/${tag}_bench_b256_trace_summary.txt $out/${tag}_cfg3_trace_summary.txt
cat $out/${tag}_conv_fwd_hbm_traffic.json
