cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/${PMC_ATTN_OUT:-r06k_pmc_attention_sq.txt}
: > $OUT
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU" \
         "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
         "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  (cd /tmp && rocprofv3 --pmc $G -d /tmp/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py --reps 2 > /tmp/pmc$i.log 2>&1)
  python tools/pmc_query.py /tmp/pmc$i "%attn_%" 2>&1 | grep -v "^no .db" >> $OUT
done
python tools/attn_microbench.py --reps 4 | tail -3 >> $OUT
python tools/attn_microbench.py --reps 4 --local 64 | tail -3 >> $OUT
cat $OUT
