set -e
cd $GRAFT_REPO_ROOT
for cfg in "--dtype bf16 --batch 128 --image-size 512 --tune-file profiles/r3_cfg3_tune.json" "--tune-file profiles/r3_b256_tune.json" "--dtype bf16 --batch 64 --image-size 512 --resnet50 --tune-file profiles/r3_r50_tune.json"; do
for e in 0 1 x; do
if [ $e = x ]; then unset LOANS_BN_NT; else export LOANS_BN_NT=$e; fi
python3 bench.py --no-secondary --no-cpu-baseline $cfg --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nt=$e', d['ms_per_step'], d['config']['baseline_config'])"
done; done
