set -e
for w in fp32 cfg3 r50; do
  bash tools/profile_round.sh r6 $w > gpurun_out/r6_profile_$w.log 2>&1 || (tail -30 gpurun_out/r6_profile_$w.log; exit 1)
  tail -4 gpurun_out/r6_profile_$w.log
done
