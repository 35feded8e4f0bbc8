#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r5_gputest_full.log 2>&1; rc=$?
echo "exit $rc"; tail -5 gpurun_out/r5_gputest_full.log
