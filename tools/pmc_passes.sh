# PMC passes over tools/attn_microbench.py (one rocprofv3 run per counter group); prints per-kernel averages
# usage: [PMC_BENCH=tools/fmlp_bwd_bench.py] bash tools/pmc_passes.sh <kernel-name-pattern> [microbench args...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PAT=$1; shift
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU" \
         "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_TAG_STALL_sum TCC_ATOMIC_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  (cd /tmp && rocprofv3 --pmc $G -d /tmp/pmc$i -o p -- python $GRAFT_REPO_ROOT/${PMC_BENCH:-tools/attn_microbench.py} --reps 2 "$@" > /tmp/pmc$i.log 2>&1)
  python tools/pmc_query.py /tmp/pmc$i "%$PAT%" 2>&1 | grep -v "^no .db"
done
