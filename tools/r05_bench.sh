#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05i}
timeout ${2:-420} python bench.py > gpurun_out/${T}_bench_sceneS.json 2> gpurun_out/${T}_bench_sceneS.err
echo "bench rc=$?"
grep '^\[bench' gpurun_out/${T}_bench_sceneS.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench_sceneS.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "exact", d["ms_per_step_f32_exact"], "bf16", d["ms_per_step_bf16"], "local", d.get("ms_per_step_local"))
    r = d["roofline"]
    print("northstar", r.get("ms_per_launch"), r.get("frac"), r["kernel"][:60])
    for k in r["kernels"]:
        print("  %.3f ms x%.1f frac %.3f  %s" % (k["ms_per_launch"], k["launches_per_step"], k["frac"], k["kernel"][:90]))
    print("local", json.dumps(d.get("local_graph"))[:900])
    print("pre", d["preprocess"]["value"] if d.get("preprocess") else None, "cpu", d["cpu_baseline"])
except Exception as e:
    print("bench parse failed", e)
    print(open("gpurun_out/${T}_bench_sceneS.err").read()[-3000:])
PY
