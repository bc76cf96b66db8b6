#!/bin/bash
# round 5, preprocessing visit: kNN + eigenfeatures out of one kernel - parity, timing, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05p}
timeout 600 python -m pytest tests/test_neighbors_gpu.py -q -x --no-header -p no:cacheprovider -s -k "knn_1_features or probe or geometric" > gpurun_out/${T}_pytest_fused.log 2>&1
echo "fused tests rc=$?"
grep -E "passed|failed|^FAILED|^E  " gpurun_out/${T}_pytest_fused.log | cut -c1-200 | head -12
grep "bit for bit" gpurun_out/${T}_pytest_fused.log | sort | uniq -c | sort -n | head -8
grep "demo room" gpurun_out/${T}_pytest_fused.log | head -2
timeout 300 python tools/pre_fused_bench.py S > gpurun_out/${T}_preprocess_legs.txt 2>&1
timeout 300 python tools/pre_fused_bench.py D >> gpurun_out/${T}_preprocess_legs.txt 2>&1
grep scene gpurun_out/${T}_preprocess_legs.txt
if [ "$2" = "trace" ]; then
  rm -rf /tmp/kt_pre
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_pre -- python $GRAFT_REPO_ROOT/tools/pre_fused_bench.py S 15000000 3 > /dev/null 2>&1)
  python tools/rocpd_summary.py /tmp/kt_pre > gpurun_out/${T}_preprocess_S_kernel_stats.csv
  head -12 gpurun_out/${T}_preprocess_S_kernel_stats.csv | cut -c1-90,200-290
fi
if [ "$3" = "full" ]; then
  timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x --no-header -p no:cacheprovider -s -k "cpu_twin" > gpurun_out/${T}_pytest_fullsize.log 2>&1
  echo "fullsize rc=$?"
  grep -E "passed|failed|^FAILED|^E  |bit for bit" gpurun_out/${T}_pytest_fullsize.log | cut -c1-200 | head
fi
