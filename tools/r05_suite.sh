#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05p}
timeout 2400 python -m pytest tests/ -q -m gpu --no-header -p no:cacheprovider > gpurun_out/${T}_pytest.log 2>&1
echo "gpu suite rc=$?"
grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest.log | grep -v '^E    *+' | cut -c1-220 | head -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for A in "--dtype bf16" "--model spt128 --scene T" "--scene T" "--mode iteration --scene T" "--graph local --order grouped"; do
python bench.py $A --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$A', d['ms_per_step'], d['value'], d['roofline'].get('ms_transform_chain'))"
done
