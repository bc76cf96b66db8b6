cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2o_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2o_pytest.log
tail -6 gpurun_out/r2o_pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess > gpurun_out/r2o_bench.json 2> gpurun_out/r2o_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r2o_bench.json')); print('S', d['ms_per_step'], d['value'], d['roofline']['frac'])"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess --scene T 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('T', d['ms_per_step'], d['value'])"
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_step -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess > /tmp/prof_step.log 2>&1)
db=$(find /tmp/prof_step -name "*.db" | head -1)
python tools/rocpd_summary.py $db > gpurun_out/r2o_spt64_trainstep_sceneS_kernel_stats.csv
head -14 gpurun_out/r2o_spt64_trainstep_sceneS_kernel_stats.csv | cut -c1-90,170-290
