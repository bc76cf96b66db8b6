#!/bin/bash
# after the last visit: the bench line once more (no CPU legs), kernel stats of the preprocessing leg
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05yy}
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/${T}_bench_sceneS_nocpu.json
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_sceneS_nocpu.json').read())
print('S', d['ms_per_step'], 'pre', d['preprocess']['value'], d['preprocess']['roofline']['knn_geof'])
PY
rm -rf /tmp/kt_pre
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_pre -- python $GRAFT_REPO_ROOT/tools/pre_fused_bench.py S 15000000 3 > $GRAFT_REPO_ROOT/gpurun_out/${T}_preprocess_legs.txt 2>&1)
python tools/rocpd_summary.py /tmp/kt_pre > gpurun_out/${T}_preprocess_S_kernel_stats.csv
grep scene gpurun_out/${T}_preprocess_legs.txt
timeout 200 python tools/pre_fused_bench.py D >> gpurun_out/${T}_preprocess_legs.txt 2>&1
grep scene gpurun_out/${T}_preprocess_legs.txt | tail -1
head -5 gpurun_out/${T}_preprocess_S_kernel_stats.csv | cut -c1-50,190-260
