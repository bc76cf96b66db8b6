#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05e}
timeout 1200 python -m pytest tests/test_fused_pool_gpu.py -q --no-header -p no:cacheprovider > gpurun_out/${T}_pytest_pool.log 2>&1
echo "pool tests rc=$?"
grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest_pool.log | grep -v '^E    *+' | cut -c1-200 | head -40
python tools/fpool_bench.py --reps 5 2>&1 | tail -2
python tools/fpool_bench.py --reps 5 --mode 3 2>&1 | tail -2
