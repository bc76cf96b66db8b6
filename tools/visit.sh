cd $GRAFT_REPO_ROOT
for on in 1 0 1 0 1 0; do
  SPT_SEG_STREAM=$on timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('stream=$on ms/step', d['ms_per_step'], 'segmax ms', r['ms_per_launch'], 'frac', r['frac'])"
done
