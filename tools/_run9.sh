set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_gputest_salt0.log 2>&1
rc=$?
{ echo "== default session (LOANS_TUNE_SALT=0): python -m pytest tests -m gpu -x -q (exit code $rc)"; tail -n 6 gpurun_out/r6_gputest_salt0.log | cut -c1-300; } > gpurun_out/r6_gputest_salt0.txt
cat gpurun_out/r6_gputest_salt0.txt
[ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" gpurun_out/r6_gputest_salt0.log | head -30; exit $rc; }
for k in 9 3; do
  PYTHONHASHSEED=$k LOANS_TUNE_SALT=$k python3 -m pytest tests/test_gpu_tall_frames.py -m gpu -x -q -s -p no:cacheprovider > gpurun_out/r6_tall_frames_salt$k.log 2>&1
  rc=$?
  { echo "== LOANS_TUNE_SALT=$k: tests/test_gpu_tall_frames.py (exit code $rc)"; grep "per image\|took the other branch\|d loss / d rois\|passed\|failed" gpurun_out/r6_tall_frames_salt$k.log | cut -c1-300; } > gpurun_out/r6_tall_frames_salt$k.txt
  cat gpurun_out/r6_tall_frames_salt$k.txt
  [ $rc -eq 0 ] || { tail -40 gpurun_out/r6_tall_frames_salt$k.log; exit $rc; }
done
