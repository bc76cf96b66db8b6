#!/bin/bash
# HBM traffic of the preprocessing kernels (two PMC passes, as the microarchitecture guide prescribes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcp_$C
  (cd /tmp && timeout 300 rocprofv3 --pmc $C -d /tmp/pmcp_$C -o p -- python $GRAFT_REPO_ROOT/tools/pre_fused_bench.py S 15000000 1 > /tmp/pmcp_$C.log 2>&1)
  python tools/pmc_query.py /tmp/pmcp_$C "%knn_cell_kernel%" 2>&1 | grep -v "^no .db" | cut -c1-130
  python tools/pmc_query.py /tmp/pmcp_$C "%point_geof_dense_kernel%" 2>&1 | grep -v "^no .db" | cut -c1-130
done
