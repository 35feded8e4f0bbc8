#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
(time python bench.py) > gpurun_out/r5_default_bench.log 2>&1; grep "^{" gpurun_out/r5_default_bench.log | tail -1 > gpurun_out/r5_default_bench.json; tail -4 gpurun_out/r5_default_bench.log | cut -c1-200
bash tools/gputest_runs.sh 1 3
