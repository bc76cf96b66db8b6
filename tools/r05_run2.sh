#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05b}
timeout 1200 python -m pytest tests/test_fused_pool_gpu.py -q --no-header -p no:cacheprovider > gpurun_out/${T}_pytest_pool.log 2>&1
echo "pool tests rc=$?"
grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest_pool.log | head -60
timeout 600 python -m pytest "tests/test_modes_gpu.py::test_two_attention_backward_formulations_interleaved_on_two_streams" "tests/test_attention_gpu.py::test_edge_lane_backward_stays_inside_an_exact_workspace_when_edges_are_fewer_than_nodes" -q --no-header -p no:cacheprovider > gpurun_out/${T}_pytest_misc.log 2>&1
echo "misc tests rc=$?"
grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest_misc.log | head -20
rm -rf /tmp/kt
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --settle 1 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /tmp/kt.log 2>&1)
python tools/rocpd_summary.py /tmp/kt > gpurun_out/${T}_kernel_stats.csv
head -25 gpurun_out/${T}_kernel_stats.csv | cut -c1-150
grep -E 'fpool|reduce_tables|gn_bwd_stats_sparse' gpurun_out/${T}_kernel_stats.csv | cut -c1-170
