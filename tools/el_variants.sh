# A/B of the measurement variants of the edge-lane attention backward (gpurun_variants/lib_el_*.so)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
for f in /tmp/lib_base.so gpurun_variants/lib_el_*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  rm -rf /tmp/pv; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pv -o v -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py --packed 2 --reps 3 > /dev/null 2>&1)
  echo "== $f"; python tools/rocpd_top.py /tmp/pv 12 | grep "attn_bwd_el\|kv_reduce"
done
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
