#!/bin/bash
# A/B the prebuilt library variants gpurun_variants/lib_knn_*.so on the preprocessing legs
# (tools/pre_fused_bench.py: knn_1, geometric_features, knn_1_features)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_main.so
for f in /tmp/lib_main.so gpurun_variants/lib_knn_*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  echo "== $f"
  for SC in ${SCENES:-S D}; do
    timeout 120 python tools/pre_fused_bench.py $SC 2>&1 | grep scene
  done
done
cp /tmp/lib_main.so superpoint_transformer_amd/lib/libspt_hip.so
