#!/bin/bash
# One visit: the default bench line, the other BASELINE configs, kernel stats of the train step at
# scenes S and T (rocprofv3 --kernel-trace of the same command), PMC traffic of the roofline kernels.
#   tools/round_profiles.sh <tag>      -> gpurun_out/<tag>_*
cd $GRAFT_REPO_ROOT
TAG=${1:-r05x}
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py 2> gpurun_out/${TAG}_bench_sceneS.err | grep '^{"metric' > gpurun_out/${TAG}_bench_sceneS.json
python bench.py --scene T --no-cpu-baseline --no-preprocess 2>/dev/null | grep '^{"metric' > gpurun_out/${TAG}_bench_sceneT.json
: > gpurun_out/${TAG}_bench_configs.jsonl
for ARGS in "--mode infer --scene D" "--mode panoptic" "--model spt128 --scene T" "--dtype bf16" "--dtype f32-exact --no-f32-exact" "--mode iteration --scene T" "--graph local --order grouped"; do
  python bench.py $ARGS --no-cpu-baseline --no-preprocess --no-f32-exact --no-local --steps 8 2>/dev/null | grep '^{"metric' >> gpurun_out/${TAG}_bench_configs.jsonl
done
for SC in S T; do
  rm -rf /tmp/kt_$SC
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_$SC -- python $GRAFT_REPO_ROOT/bench.py --scene $SC --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /dev/null 2>&1)
  python tools/rocpd_summary.py /tmp/kt_$SC > gpurun_out/${TAG}_spt64_trainstep_scene${SC}_kernel_stats.csv
done
rm -rf /tmp/kt_L
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_L -- python $GRAFT_REPO_ROOT/bench.py --graph local --order grouped --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /dev/null 2>&1)
python tools/rocpd_summary.py /tmp/kt_L > gpurun_out/${TAG}_spt64_trainstep_sceneS_local_grouped_kernel_stats.csv
rm -rf /tmp/kt_128
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_128 -- python $GRAFT_REPO_ROOT/bench.py --model spt128 --scene T --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /dev/null 2>&1)
python tools/rocpd_summary.py /tmp/kt_128 > gpurun_out/${TAG}_spt128_trainstep_sceneT_kernel_stats.csv
rm -rf /tmp/kt_bf
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_bf -- python $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /dev/null 2>&1)
python tools/rocpd_summary.py /tmp/kt_bf > gpurun_out/${TAG}_spt64_trainstep_sceneS_bf16_kernel_stats.csv
bash tools/pmc_step.sh > gpurun_out/${TAG}_pmc_step_traffic.txt 2>&1
bash tools/pmc_step.sh --dtype bf16 > gpurun_out/${TAG}_pmc_step_traffic_bf16.txt 2>&1
bash tools/pmc_step.sh --graph local --order grouped > gpurun_out/${TAG}_pmc_step_traffic_local_grouped.txt 2>&1
bash tools/pmc_segmax.sh > gpurun_out/${TAG}_pmc_segmax_standalone.txt 2>&1
for SC in S D; do python tools/knn_bench.py $SC 0 3 2>/dev/null | tail -1; done > gpurun_out/${TAG}_preprocess_legs.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
head -c 400 gpurun_out/${TAG}_bench_sceneS.json; echo
python - <<PY
import json
for l in open('gpurun_out/${TAG}_bench_configs.jsonl'):
    d = json.loads(l); print(d['config'].get('mode'), d['config'].get('net'), d['config'].get('scene'), d['config'].get('graph'), d['dtype'][:12], d['ms_per_step'])
d = json.loads(open('gpurun_out/${TAG}_bench_sceneT.json').read()); print('T', d['ms_per_step'])
PY
