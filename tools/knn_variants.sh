#!/bin/bash
# A/B the prebuilt library variants of gpurun_variants/lib_knn_*.so: knn_cell_kernel's own time
# (rocprofv3 kernel trace) and the preprocess legs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_main.so
for f in /tmp/lib_main.so gpurun_variants/lib_knn_*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  echo "== $f"
  for SC in ${SCENES:-S}; do
    rm -rf /tmp/kt_v
    (cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_v -- python $GRAFT_REPO_ROOT/tools/knn_bench.py $SC 0 2 2>/dev/null | tail -1)
    python tools/rocpd_summary.py /tmp/kt_v | grep "knn_cell_kernel\|knn_search_kernel" | sed 's/.*)",/   /'
  done
done
cp /tmp/lib_main.so superpoint_transformer_amd/lib/libspt_hip.so
