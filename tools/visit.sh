cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d /tmp/pmc_step -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-preprocess > /tmp/pmc_step.log 2>&1)
python tools/pmc_query.py /tmp/pmc_step "%fmlp%" > gpurun_out/r2q_pmc_fmlp.txt 2>&1
grep -E "bwd_kernel_bf<16, 8|bwd_kernel_bf<8, 4|fwd_kernel<16, 8" gpurun_out/r2q_pmc_fmlp.txt | cut -c1-30,40-140
tail -3 /tmp/pmc_step.log
