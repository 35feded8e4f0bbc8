set -e
cd $GRAFT_REPO_ROOT
for e in 0 1; do
LOANS_EARLY_CHAIN=$e python3 bench.py --no-secondary --no-cpu-baseline --tune-file profiles/r3_b256_tune.json --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('early=$e', d['ms_per_step'], r['frac'], r['conv_fwd_ms_per_step'], r['whole_step']['frac'], r['binding']['frac'])"
done
