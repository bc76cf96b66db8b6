# A/B of whole-library variants on bench.py legs:  tools/lib_ab.sh "<bench args>" ["<bench args>" ...]
# runs every leg with the in-tree library and with each gpurun_variants/lib_*.so, prints ms_per_step
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-preprocess --no-f32-exact --no-local --no-train-batch --steps 20"
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
for round in 1 2; do
for f in /tmp/lib_base.so gpurun_variants/lib_*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  for A in "$@"; do
    python bench.py $A $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$f', '[$A]', 'ms/step', d['ms_per_step'])"
  done
done
done
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
