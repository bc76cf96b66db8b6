#!/bin/bash
# A/B the library variants of gpurun_variants/ on the fused-MLP backward microbench
cd $GRAFT_REPO_ROOT
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
for f in /tmp/lib_base.so gpurun_variants/*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  echo "== $f"
  python tools/fmlp_bwd_bench.py "$@" 2>&1 | grep -v amdgpu.ids | head -2
done
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
