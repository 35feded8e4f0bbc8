#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
out=$PWD/gpurun_out/r5_conv_kernels_sq.txt
: > $out
cd /tmp && export TMPDIR=/tmp
run() {  # run <label> <substring> B H W Cin Cout tile
  rm -rf /tmp/pmc1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/halo_bench.py $3 $4 $5 $6 $7 $8 > /tmp/pmc1.log 2>&1 || { tail -5 /tmp/pmc1.log; return 1; }
  echo "== $1: B $3, $4 x $5, $6 -> $7, tile $8" >> $out
  grep "^tile" /tmp/pmc1.log >> $out
  python3 $GRAFT_REPO_ROOT/tools/pmc_kernel.py /tmp/pmc1 "$2" >> $out
}
run "res2 3x3 (wsw_kernel)" wsw_kernel 128 128 128 64 64 37 && \
run "res3 3x3 (halo 16x16x128)" halo16_kernel 128 64 64 128 128 36 && \
run "res4 3x3 (igemm16 256x256)" igemm16_kernel 128 32 32 256 256 9 && \
run "res5 3x3 (igemm16 256x256)" igemm16_kernel 128 16 16 512 512 9
cat $out
cd $GRAFT_REPO_ROOT && timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k one_process > gpurun_out/try_tests.log 2>&1; echo "tests exit $?" | tee -a gpurun_out/try_tests.log; tail -5 gpurun_out/try_tests.log
