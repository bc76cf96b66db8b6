set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2g_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g_pytest.log
tail -12 gpurun_out/r2g_pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r2g_bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
