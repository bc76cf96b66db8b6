cd $GRAFT_REPO_ROOT
# SQ counters of the pool-fused top layer (fwd_pool_kernel / bwd_pool_kernel) at scene-S size, through
# tools/fpool_bench.py: what the wave cycles are spent on (VALU / MFMA / LDS / waiting), LDS bank
# conflicts, instruction counts.  One counter group per pass (the pool's rule: --pmc alone).
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/${PMC_FPOOL_OUT:-r06_pmc_fpool_sq.txt}
MODE=${PMC_FPOOL_MODE:-1}
: > $OUT
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU" \
         "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcf$i
  (cd /tmp && rocprofv3 --pmc $G -d /tmp/pmcf$i -o p -- python $GRAFT_REPO_ROOT/tools/fpool_bench.py --mode $MODE --reps 2 > /tmp/pmcf$i.log 2>&1)
  tail -2 /tmp/pmcf$i.log | grep -i -E "error|invalid|unknown" >> $OUT
  python tools/pmc_query.py /tmp/pmcf$i "%_pool_kernel%" 2>&1 | grep -v "^no .db" >> $OUT
done
python tools/fpool_bench.py --mode $MODE --reps 5 | tail -4 >> $OUT
cat $OUT
