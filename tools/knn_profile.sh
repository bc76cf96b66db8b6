#!/bin/bash
# kernel trace + two PMC passes of the preprocess leg (tools/knn_bench.py <scene>)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
SC=${1:-S}
rm -rf /tmp/kt_knn
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/kt_knn -- python $GRAFT_REPO_ROOT/tools/knn_bench.py $SC 0 2 > /tmp/kt_knn.log 2>&1)
python tools/rocpd_summary.py /tmp/kt_knn | cut -c1-140 | head -24
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pmck$i
  (cd /tmp && rocprofv3 --pmc $G -d /tmp/pmck$i -o p -- python $GRAFT_REPO_ROOT/tools/knn_bench.py $SC 0 1 > /tmp/pmck$i.log 2>&1)
  python tools/pmc_query.py /tmp/pmck$i "%knn_cell_kernel%" 2>&1 | grep -v "^no .db" | cut -c1-130
done
