set -e
mkdir -p gpurun_out
{
python tools/halo_bench.py 128 64 64 128 128 7,9,41,36
python tools/halo_bench.py 128 32 32 256 256 9,41,7
python tools/halo_bench.py 128 16 16 512 512 9,41,7
python tools/halo_bench.py 64 64 64 128 128 7,9,41
python tools/halo_bench.py 64 32 32 256 256 9,41,7
python tools/halo_bench.py 64 16 16 512 512 9,41,7
} > gpurun_out/r6_w4_bench.txt 2>&1
python bench.py > gpurun_out/r6_first_bench.log 2>&1
cp bench_detail.json gpurun_out/r6_first_bench_detail.json
tail -c 3000 gpurun_out/r6_first_bench.log
python -m pytest tests/test_gpu_launch.py -m gpu -x -q > gpurun_out/r6_launch_tests.log 2>&1 || (tail -30 gpurun_out/r6_launch_tests.log; exit 1)
tail -3 gpurun_out/r6_launch_tests.log
