# PMC FETCH_SIZE / WRITE_SIZE (separate passes) of the train step's leading kernels at scene S
# usage: bash tools/pmc_step.sh [extra bench.py args, e.g. --dtype bf16]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for G in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmcs_$G
  (cd /tmp && rocprofv3 --pmc $G -d /tmp/pmcs_$G -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --settle 0 --no-cpu-baseline --no-preprocess --no-f32-exact --no-local "$@" > /tmp/pmcs_$G.log 2>&1)
  for K in segmax_stream attn_bwd_to attn_q_reduce attn_bwd_el attn_kv_reduce attn_fwd_mfma "bwd_dma_kernel<64, 128" "fwd_kernel<16, 8" "fwd_kernel_x3<16, 8" segcsr_bwd fwd_pool_kernel bwd_pool_kernel pool_apply_kernel pool_bwd_gm_kernel gn_bwd_stats_sparse; do
    python tools/pmc_query.py /tmp/pmcs_$G "%$K%" 2>&1 | grep -v "^no .db" | cut -c1-220
  done
done
