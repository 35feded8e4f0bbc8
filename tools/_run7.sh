set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
out=gpurun_out/r6_pp_clock.txt
: > $out
export HALO_BENCH_LIB=loans_amd/csrc/libloans_hip_exp.so
for cfg in "0 43 igemm16pp" "24 43 igemm16pp" "32 43 igemm16pp" "0 9 igemm16_kernel"; do
  set -- $cfg
  export LOANS_DBG=$1
  for shape in "128 16 16 512 512" "128 32 32 256 256"; do
    echo "== LOANS_DBG=$1 tile $2 shape $shape" >> $out
    rm -rf gpurun_out/_pmc gpurun_out/_tr
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/_pmc -- python3 tools/halo_bench.py $shape $2 > gpurun_out/_pmc.log 2>&1
    python3 tools/pmc_kernel.py gpurun_out/_pmc $3 >> $out
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_tr -- python3 tools/halo_bench.py $shape $2 > gpurun_out/_tr.log 2>&1
    grep "$3" gpurun_out/_tr/*/*kernel_stats.csv | cut -c1-200 >> $out
  done
done
rm -rf gpurun_out/_pmc gpurun_out/_tr
cat $out
