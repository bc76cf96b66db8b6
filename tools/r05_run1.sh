#!/bin/bash
# round 5, GPU visit 1: the pool-fused top layer + the boundary fixes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_pool_gpu.py -q -x --no-header -p no:cacheprovider > gpurun_out/r05a_pytest_pool.log 2>&1
echo "pool tests rc=$?" 
tail -30 gpurun_out/r05a_pytest_pool.log
timeout 900 python -m pytest tests/test_sub_views_gpu.py tests/test_modes_gpu.py "tests/test_attention_gpu.py" -q --no-header -p no:cacheprovider > gpurun_out/r05a_pytest_misc.log 2>&1
echo "misc tests rc=$?"
tail -15 gpurun_out/r05a_pytest_misc.log
for on in 1 0; do
  SPT_POOL_IN_FORWARD=$on timeout 600 python bench.py --steps 8 --warmup 2 --settle 2 --no-cpu-baseline --no-preprocess --no-f32-exact > gpurun_out/r05a_bench_pool$on.json 2> gpurun_out/r05a_bench_pool$on.err
  echo "bench pool=$on rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05a_bench_pool$on.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"])
    for k in d["roofline"]["kernels"]:
        print("  %.3f ms x%.1f frac %.3f  %s" % (k["ms_per_launch"], k["launches_per_step"], k["frac"], k["kernel"][:90]))
except Exception as e:
    print("bench parse failed", e)
    print(open("gpurun_out/r05a_bench_pool$on.err").read()[-2000:])
PY
done
