set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "crop_dgrad" > gpurun_out/r6_crop_tests.log 2>&1 || (tail -40 gpurun_out/r6_crop_tests.log; exit 1)
tail -2 gpurun_out/r6_crop_tests.log
{
timeout -k 10 120 python tools/crop_bench.py 128 bf16
timeout -k 10 120 python tools/crop_bench.py 64 bf16
timeout -k 10 120 python tools/crop_bench.py 256
} > gpurun_out/r6_crop_bench.txt 2>&1 || (tail -20 gpurun_out/r6_crop_bench.txt; exit 1)
grep -v amdgpu gpurun_out/r6_crop_bench.txt
timeout -k 10 900 python -m pytest tests/test_gpu_bf16_storage.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r6_bf16_tests.log 2>&1 || (tail -60 gpurun_out/r6_bf16_tests.log; exit 1)
tail -3 gpurun_out/r6_bf16_tests.log
