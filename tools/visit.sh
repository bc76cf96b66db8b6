#!/bin/bash
# One GPU visit, parameterised (replaces the per-visit r05_*.sh scripts):
#   tools/visit.sh <tag> <leg> [<leg> ...]        -> gpurun_out/<tag>_*
# legs:
#   tests[:<pytest args>]  the -m gpu suite (or the given selection), summary + log
#   smoke                  __graft_entry__.smoke()
#   bench[:<bench args>]   the default bench line (or with the given arguments) -> <tag>_bench<suffix>.json
#   configs                the other BASELINE configurations, one line each -> <tag>_bench_configs.jsonl
#   stats:<S|T|L|128|bf16> rocprofv3 --kernel-trace of the train step -> <tag>_*_kernel_stats.csv
#   pmc[:<bench args>]     FETCH_SIZE / WRITE_SIZE passes of the step (tools/pmc_step.sh)
#   pre                    preprocessing legs (tools/knn_bench.py at S and D settings)
#   py:<script and args>   any tools/*.py, output -> <tag>_<script>.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=$1; shift
QUIET="--no-cpu-baseline --no-preprocess --no-f32-exact --no-local --no-train-batch"
for LEG in "$@"; do
  KIND=${LEG%%:*}; ARG=""; [[ "$LEG" == *:* ]] && ARG=${LEG#*:}
  case $KIND in
    tests)
      timeout 2700 python -m pytest ${ARG:-tests/} -q -m gpu --no-header -p no:cacheprovider -s \
        > gpurun_out/${TAG}_pytest.log 2>&1
      echo "gpu tests rc=$?"
      grep -E ' passed| failed|^FAILED|^ERROR|^E  ' gpurun_out/${TAG}_pytest.log | grep -v '^E    *+' | cut -c1-240 | head -40
      grep -E ' passed| failed' gpurun_out/${TAG}_pytest.log | tail -1 > gpurun_out/${TAG}_pytest_summary.txt
      grep -E 'arg-mismatch fraction|single-kink rows|near-kink element' gpurun_out/${TAG}_pytest.log \
        > gpurun_out/${TAG}_pytest_parity_notes.txt ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)
      SUF=$(echo "$ARG" | tr -c 'A-Za-z0-9\n' '_' | sed 's/__*/_/g; s/^_//; s/_$//')
      OUT=gpurun_out/${TAG}_bench${SUF:+_$SUF}.json
      python bench.py $ARG 2> ${OUT%.json}.err | grep '^{"metric' > $OUT
      python - "$OUT" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
tb = d.get("train_batch") or {}
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "value", d["value"], "local", d.get("ms_per_step_local"),
      "bf16", d.get("ms_per_step_bf16"), "T", tb.get("ms_per_step_T"), "T eager", tb.get("ms_per_step_T_eager"),
      "iter T", tb.get("ms_per_iteration_T"), "pre", (d.get("preprocess") or {}).get("value"))
for k in (d.get("roofline") or {}).get("kernels", []):
    print("   ", k["kernel"][:70], k["ms_per_launch"], k.get("frac"))
PY
      ;;
    configs)
      : > gpurun_out/${TAG}_bench_configs.jsonl
      for A in "--mode infer --scene D" "--mode panoptic" "--model spt128 --scene T" "--dtype bf16" \
               "--dtype f32-exact" "--mode iteration --scene T" "--graph local --order grouped" "--scene T"; do
        python bench.py $A $QUIET --steps 8 2>/dev/null | grep '^{"metric' >> gpurun_out/${TAG}_bench_configs.jsonl
      done
      python - gpurun_out/${TAG}_bench_configs.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); c = d["config"]
    print(c.get("mode"), c.get("net"), c.get("scene"), c.get("graph"), d["dtype"][:10], d["ms_per_step"], d["value"])
PY
      ;;
    stats)
      case $ARG in
        S) A="--scene S"; N=spt64_trainstep_sceneS ;;
        T) A="--scene T"; N=spt64_trainstep_sceneT ;;
        Tgraph) A="--scene T --capture"; N=spt64_trainstep_sceneT_captured ;;
        L) A="--graph local --order grouped"; N=spt64_trainstep_sceneS_local_grouped ;;
        128) A="--model spt128 --scene T"; N=spt128_trainstep_sceneT ;;
        128graph) A="--model spt128 --scene T --capture"; N=spt128_trainstep_sceneT_captured ;;
        bf16) A="--dtype bf16"; N=spt64_trainstep_sceneS_bf16 ;;
      esac
      rm -rf /tmp/kt_$ARG
      (cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_$ARG -- python $GRAFT_REPO_ROOT/bench.py $A --steps 5 --warmup 2 --settle 0.5 $QUIET > /dev/null 2>&1)
      python tools/rocpd_summary.py /tmp/kt_$ARG > gpurun_out/${TAG}_${N}_kernel_stats.csv
      python tools/launch_count.py gpurun_out/${TAG}_${N}_kernel_stats.csv ;;
    pmc)
      SUF=$(echo "$ARG" | tr -c 'A-Za-z0-9\n' '_' | sed 's/__*/_/g; s/^_//; s/_$//')
      bash tools/pmc_step.sh $ARG > gpurun_out/${TAG}_pmc_step_traffic${SUF:+_$SUF}.txt 2>&1
      tail -25 gpurun_out/${TAG}_pmc_step_traffic${SUF:+_$SUF}.txt ;;
    pre)
      for SC in S D; do python tools/knn_bench.py $SC 0 3 2>/dev/null | tail -1; done | tee gpurun_out/${TAG}_preprocess_legs.txt ;;
    py)
      NAME=$(basename ${ARG%% *} .py)
      python tools/$ARG > gpurun_out/${TAG}_${NAME}.txt 2>&1; tail -30 gpurun_out/${TAG}_${NAME}.txt ;;
    *) echo "unknown leg $LEG" ;;
  esac
done
