#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05q}
timeout 1500 python -m pytest tests/test_fused_pool_gpu.py tests/test_batch_pipeline_gpu.py "tests/test_fused_mlp_gpu.py::test_fused_mlp_matches_oracle_and_unfused_path" tests/test_modes_gpu.py -q --no-header -p no:cacheprovider > gpurun_out/${T}_pytest_sel.log 2>&1
echo "selected tests rc=$?"
grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest_sel.log | grep -v '^E    *+' | cut -c1-220 | head -40
grep -E 'elements above' gpurun_out/${T}_pytest_sel.log | head -5
rm -rf /tmp/kt_bf
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_bf -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /tmp/kt_bf.log 2>&1)
python tools/rocpd_summary.py /tmp/kt_bf > gpurun_out/${T}_bf16_kernel_stats.csv
head -12 gpurun_out/${T}_bf16_kernel_stats.csv | cut -c1-130
grep '^{"metric' /tmp/kt_bf.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 (profiled)', d['ms_per_step'])"
