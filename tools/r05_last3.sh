#!/bin/bash
# the other BASELINE configurations and the train-batch regime on the round's last commit
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05y}
python bench.py --scene T --no-cpu-baseline --no-preprocess 2>/dev/null | grep '^{"metric' > gpurun_out/${T}_bench_sceneT.json
: > gpurun_out/${T}_bench_configs.jsonl
for ARGS in "--mode infer --scene D" "--mode panoptic" "--model spt128 --scene T" "--dtype bf16" "--mode iteration --scene T" "--graph local --order grouped"; do
  timeout 300 python bench.py $ARGS --no-cpu-baseline --no-preprocess --no-f32-exact --no-local --steps 8 2>/dev/null | grep '^{"metric' >> gpurun_out/${T}_bench_configs.jsonl
done
rm -rf /tmp/kt_T
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_T -- python $GRAFT_REPO_ROOT/bench.py --scene T --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /dev/null 2>&1)
python tools/rocpd_summary.py /tmp/kt_T > gpurun_out/${T}_spt64_trainstep_sceneT_kernel_stats.csv
python - <<PY
import json
for l in open('gpurun_out/${T}_bench_configs.jsonl'):
    d = json.loads(l); print(d['config'].get('mode'), d['config'].get('net'), d['config'].get('scene'), d['config'].get('graph'), d['dtype'][:12], d['ms_per_step'])
d = json.loads(open('gpurun_out/${T}_bench_sceneT.json').read()); print('T', d['ms_per_step'], d['value'])
PY
