set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_modes_gpu.py -x -q > gpurun_out/r2e_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e_pytest.log
tail -15 gpurun_out/r2e_pytest.log
for cfg in "--mode infer --scene D" "--mode panoptic --scene S" "--mode train --model spt128 --scene T" "--mode train --scene T"; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-preprocess $cfg >> gpurun_out/r2e_bench_modes.jsonl 2>> gpurun_out/r2e_bench_modes.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2e_bench_modes.jsonl'):
    d = json.loads(l); print(d['config']['mode'], d['config']['net'], d['config']['scene'], d['ms_per_step'], 'ms', d['value'], 'Mpts/s')
PY
tail -3 gpurun_out/r2e_bench_modes.err
for sc in D V S; do timeout 300 python tools/knn_bench.py $sc 0 2 >> gpurun_out/r2e_knn.log 2>&1; done
cat gpurun_out/r2e_knn.log | grep scene
