set -e
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-secondary --no-cpu-baseline $1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$2', d['ms_per_step'], 'conv fwd', r['conv_fwd_ms_per_step'], d['config']['baseline_config'])"; }
export LOANS_BENCH_RETUNE=1
for cfg in "--dtype bf16 --batch 128 --image-size 512" "--dtype bf16 --batch 64 --image-size 512 --resnet50"; do
LOANS_CONV_NT_MB=-1 LOANS_BN_NT_MB=96 run "$cfg" "conv off bn>=96 "
LOANS_CONV_NT_MB=-1 LOANS_BN_NT_MB=200 run "$cfg" "conv off bn>=200"
LOANS_CONV_NT_MB=-1 LOANS_BN_NT_MB=300 run "$cfg" "conv off bn>=300"
LOANS_CONV_NT_MB=200 LOANS_BN_NT_MB=200 run "$cfg" "conv>=200 bn>=200"
LOANS_CONV_NT_MB=16 LOANS_BN_NT_MB=96 run "$cfg" "conv>=16 bn>=96 "
LOANS_CONV_NT_MB=16 LOANS_BN_NT_MB=16 run "$cfg" "conv>=16 bn>=16 "
LOANS_CONV_NT_MB=-1 LOANS_BN_NT_MB=96 run "$cfg" "conv off bn>=96 "
done
