#!/bin/bash
# the round's last visit: the whole GPU suite, smoke(), the driver's bench command, kernel stats of the
# train step at scene S and of the preprocessing leg on the last commit, VALU share of the kNN kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05y}
timeout 1500 python -m pytest tests/ -q -m gpu --no-header -p no:cacheprovider > gpurun_out/${T}_pytest.log 2>&1
echo "gpu suite rc=$?"
grep -E 'passed|failed|^FAILED|^E  ' gpurun_out/${T}_pytest.log | grep -v '^E    *+' | cut -c1-220 | head -30
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/${T}_bench_sceneS.err | grep '^{"metric' > gpurun_out/${T}_bench_sceneS.json
python - <<PY
import json
d = json.loads(open('gpurun_out/${T}_bench_sceneS.json').read())
print('S', d['ms_per_step'], d['value'], 'bf16', d.get('ms_per_step_bf16'), 'exact', d.get('ms_per_step_f32_exact'), 'local', d.get('ms_per_step_local'))
print('roofline', d['roofline']['frac'], d['roofline'].get('traffic'))
p = d.get('preprocess') or {}
print('pre', p.get('value'), p.get('ms_total'), (p.get('two_calls') or {}).get('value'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
tail -4 gpurun_out/${T}_bench_sceneS.err
rm -rf /tmp/kt_S
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_S -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local > /dev/null 2>&1)
python tools/rocpd_summary.py /tmp/kt_S > gpurun_out/${T}_spt64_trainstep_sceneS_kernel_stats.csv
head -4 gpurun_out/${T}_spt64_trainstep_sceneS_kernel_stats.csv | cut -c1-60,200-270
rm -rf /tmp/kt_pre
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_pre -- python $GRAFT_REPO_ROOT/tools/pre_fused_bench.py S 15000000 3 > $GRAFT_REPO_ROOT/gpurun_out/${T}_preprocess_legs.txt 2>&1)
python tools/rocpd_summary.py /tmp/kt_pre > gpurun_out/${T}_preprocess_S_kernel_stats.csv
grep scene gpurun_out/${T}_preprocess_legs.txt
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pmck$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $G -d /tmp/pmck$i -o p -- python $GRAFT_REPO_ROOT/tools/pre_fused_bench.py S 15000000 1 > /tmp/pmck$i.log 2>&1)
  python tools/pmc_query.py /tmp/pmck$i "%knn_cell_kernel%" 2>&1 | grep -v "^no .db" | cut -c1-130
done > gpurun_out/${T}_pmc_preprocess_S.txt
head -20 gpurun_out/${T}_pmc_preprocess_S.txt
