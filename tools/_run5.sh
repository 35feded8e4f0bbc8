set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_bf16_storage.py -m gpu -x -q -k "(test_conv_bf16_storage or test_conv_pair_bf16_storage or takes_the_bn_backward_sums) and (44 or 43 or bf16)" > gpurun_out/r6_pp16_tests.log 2>&1 || (tail -40 gpurun_out/r6_pp16_tests.log; exit 1)
tail -3 gpurun_out/r6_pp16_tests.log
{
timeout -k 10 120 python tools/halo_bench.py 128 32 32 256 256 9,43,44,9,43,44
timeout -k 10 120 python tools/halo_bench.py 128 16 16 512 512 9,43,44,9,43,44
timeout -k 10 120 python tools/halo_bench.py 64 32 32 256 256 9,43,44
timeout -k 10 120 python tools/halo_bench.py 64 16 16 512 512 9,43,44,7
timeout -k 10 120 python tools/halo_bench.py 128 64 64 256 256 9,43,44
timeout -k 10 120 python tools/halo_bench.py 32 64 64 512 512 9,43,44
} > gpurun_out/r6_pp16_bench.txt 2>&1 || (tail -20 gpurun_out/r6_pp16_bench.txt; exit 1)
grep -v amdgpu.ids gpurun_out/r6_pp16_bench.txt
