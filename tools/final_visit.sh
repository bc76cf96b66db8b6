#!/bin/bash
# The evidence visit of a round's last commit:  tools/final_visit.sh <tag>  -> gpurun_out/<tag>_*
# (tests, smoke, the driver's bench command, the other configurations, kernel stats of every step
# variant, PMC traffic of the step on both graphs and in bf16, SQ counters of the attention kernels
# and of the pool-fused layer, preprocessing legs)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1
bash tools/visit.sh $TAG tests smoke bench configs stats:S stats:L stats:bf16 stats:Tgraph stats:T stats:128graph pre
bash tools/visit.sh $TAG pmc "pmc:--graph local --order grouped" "pmc:--dtype bf16"
PMC_ATTN_OUT=${TAG}_pmc_attention_sq.txt bash tools/pmc_attn.sh > /dev/null 2>&1
PMC_BENCH=tools/fpool_bench.py bash tools/pmc_passes.sh pool_kernel > gpurun_out/${TAG}_pmc_fpool.txt 2>&1
tail -5 gpurun_out/${TAG}_pmc_fpool.txt
