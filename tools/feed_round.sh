#!/bin/bash
# files -> device batches of the training loop (tools/feed_bench.py) with ONE RANK'S SHARE of the host -- an 8-GPU node with 64
# cores gives each rank 8 (train_sheep_localizer.loader_threads: affinity // world) -- and with the whole 16-core 1-GPU box:
#   tools/feed_round.sh <tag>  -> gpurun_out/<tag>_feed_bench.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r4}_feed_bench.txt
: > $out
for args in "--cores 8 --processes 8 --threads 8" "--cores 8 --processes 8 --threads 8 --no-imgaug" \
            "--processes 16 --threads 16" "--processes 16 --threads 16 --no-imgaug"; do
  python3 tools/feed_bench.py --jpeg $args 2>/dev/null | grep -v amdgpu.ids >> $out
done
cat $out
