#!/bin/bash
# last visit of round 5: the full GPU suite on the last commit + the SPT-128 line and its kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05zz}
timeout 2400 python -m pytest tests/ -q -m gpu --no-header -p no:cacheprovider > gpurun_out/${T}_pytest.log 2>&1
echo "gpu suite rc=$?"
grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest.log | grep -v '^E    *+' | cut -c1-220 | head -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --model spt128 --scene T --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local 2>/dev/null | grep '^{"metric' > gpurun_out/${T}_bench_spt128_sceneT.json
python -c "
import json
d = json.loads(open('gpurun_out/${T}_bench_spt128_sceneT.json').read()); print('spt128 T', d['ms_per_step'])"
rm -rf /tmp/kt_128
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_128 -- python $GRAFT_REPO_ROOT/bench.py --model spt128 --scene T --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /dev/null 2>&1)
python tools/rocpd_summary.py /tmp/kt_128 > gpurun_out/${T}_spt128_trainstep_sceneT_kernel_stats.csv
grep -c Cijk gpurun_out/${T}_spt128_trainstep_sceneT_kernel_stats.csv
grep Cijk gpurun_out/${T}_spt128_trainstep_sceneT_kernel_stats.csv | cut -c1-60,200-260
