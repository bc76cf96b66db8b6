#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05n}
timeout 1500 python -m pytest tests/test_fused_pool_gpu.py tests/test_sub_views_gpu.py tests/test_skinny_linear_gpu.py tests/test_batch_pipeline_gpu.py tests/test_modes_gpu.py tests/test_norms_gpu.py tests/test_pool_gpu.py -q --no-header -p no:cacheprovider > gpurun_out/${T}_pytest_sel.log 2>&1
echo "selected tests rc=$?"
grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest_sel.log | grep -v '^E    *+' | cut -c1-220 | head -40
python tools/fpool_bench.py --reps 5 2>&1 | tail -2
python tools/fpool_bench.py --reps 5 --mode 3 2>&1 | tail -2
for SC in T S; do
python bench.py --scene $SC --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$SC', d['ms_per_step'], d['value'])"
done
python bench.py --model spt128 --scene T --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spt128 T', d['ms_per_step'])"
