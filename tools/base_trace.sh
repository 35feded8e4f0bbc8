cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
one() { name=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${name}_tr -- python3 bench.py --no-secondary --steps 3 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/${name}.log 2>&1 &&
  python3 tools/step_timeline.py gpurun_out/${name}_tr > gpurun_out/${name}_timeline.txt && rm -rf gpurun_out/${name}_tr && echo "$name done"; }
one base_cfg3 --tune-file profiles/r4_cfg3_tune.json --dtype bf16 --batch 128 --image-size 512 &&
one base_r50 --tune-file profiles/r4_r50_tune.json --dtype bf16 --batch 64 --image-size 512 --resnet50
