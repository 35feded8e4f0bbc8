#!/bin/bash
# The -m gpu suite N times over, each run with another assignment of tiles to layer shapes (LOANS_TUNE_SALT: the parity session
# picks tiles by crc32(shape, mode, salt), tests/conftest.py) and another PYTHONHASHSEED -- what the driver's box may differ in.
#   tools/gputest_runs.sh <first salt> <last salt>   -> gpurun_out/r5_gputest_salt<k>.txt (the tail of each run)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for k in $(seq ${1:-0} ${2:-4}); do
  log=gpurun_out/r5_gputest_salt$k.log
  PYTHONHASHSEED=$k LOANS_TUNE_SALT=$k python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $log 2>&1
  rc=$?
  { echo "== LOANS_TUNE_SALT=$k PYTHONHASHSEED=$k: python -m pytest tests -m gpu -x -q  (exit code $rc)"; tail -n 4 $log | cut -c1-200; } > gpurun_out/r5_gputest_salt$k.txt
  cat gpurun_out/r5_gputest_salt$k.txt
done
