# A/B of the build variants of the target-order attention backward (gpurun_variants/lib_to_*.so,
# built by tools/build_variant.sh to_<name> edge_attn_to.hip -mllvm -amdgpu-mfma-vgpr-form=1 -D...):
# level-1 shape of scene S (N = 428 571, E = 7.03 M), kernel times from rocprofv3 --kernel-trace.
#   tools/to_variants.sh [--local W]      -> stdout (and gpurun_out/<TAG>_to_variants.txt if TAG is set)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
for f in /tmp/lib_base.so gpurun_variants/lib_to_*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  rm -rf /tmp/pv; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/pv -o v -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py --packed 2 --reps 4 "$@" > /dev/null 2>&1)
  echo "== $f"; python tools/rocpd_top.py /tmp/pv 14 | grep "attn_bwd_to\|q_reduce\|attn_fwd"
done
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
