mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_input_path.py tests/test_gpu_resample.py -x -q > gpurun_out/t_input.log 2>&1; echo "input rc=$?"; tail -3 gpurun_out/t_input.log
timeout -k 10 600 python -m pytest tests/test_gpu_parallel.py -x -q > gpurun_out/t_par.log 2>&1; echo "parallel rc=$?"; tail -3 gpurun_out/t_par.log
timeout -k 10 300 python tools/feed_bench.py --jpeg --cores 8 --processes 8 --threads 8 --cache device > gpurun_out/feed_cache_device.txt 2>&1; tail -6 gpurun_out/feed_cache_device.txt
timeout -k 10 300 python tools/feed_bench.py --jpeg --cores 8 --processes 8 --threads 8 --cache host > gpurun_out/feed_cache_host.txt 2>&1; tail -2 gpurun_out/feed_cache_host.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cfg3_tr -- python3 bench.py --no-secondary --steps 3 --warmup 2 --no-cpu-baseline --tune-file profiles/r4_cfg3_tune.json --dtype bf16 --batch 128 --image-size 512 > gpurun_out/cfg3_trace.log 2>&1
python3 tools/step_timeline.py gpurun_out/cfg3_tr > gpurun_out/cfg3_timeline.txt; python3 tools/class_times.py gpurun_out/cfg3_tr gpurun_out/cfg3_class_times.json; rm -rf gpurun_out/cfg3_tr
grep '^{' gpurun_out/cfg3_trace.log | python3 -c "import json,sys; d=json.load(sys.stdin); print(json.dumps(d['roofline']['whole_step'].get('binding'), indent=0)[:3000])"
