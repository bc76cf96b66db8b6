#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05zzz}
timeout 900 python -m pytest tests/test_skinny_linear_gpu.py tests/test_spt_reference.py tests/test_model_gpu.py -q --no-header -p no:cacheprovider > gpurun_out/${T}_pytest_sel.log 2>&1
echo "selected rc=$?"; grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest_sel.log | head
for i in 1 2; do
python bench.py --model spt128 --scene T --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local 2>/dev/null | grep '^{"metric' > gpurun_out/${T}_bench_spt128_sceneT.json
python -c "
import json
d = json.loads(open('gpurun_out/${T}_bench_spt128_sceneT.json').read()); print('spt128 T', d['ms_per_step'])"
done
rm -rf /tmp/kt_128
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_128 -- python $GRAFT_REPO_ROOT/bench.py --model spt128 --scene T --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /dev/null 2>&1)
python tools/rocpd_summary.py /tmp/kt_128 > gpurun_out/${T}_spt128_trainstep_sceneT_kernel_stats.csv
echo "Cijk rows: $(grep -c Cijk gpurun_out/${T}_spt128_trainstep_sceneT_kernel_stats.csv)"
python bench.py --scene T --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spt64 T', d['ms_per_step'])"
