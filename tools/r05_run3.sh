#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-r05d}
python tools/fpool_bench.py --reps 5 2>&1 | tail -3
export PMC_BENCH=tools/fpool_bench.py
bash tools/pmc_passes.sh "pool_kernel" > gpurun_out/${T}_pmc_fpool.txt 2>&1
cut -c1-200 gpurun_out/${T}_pmc_fpool.txt | grep -v 'n=0' | awk '{print $1, $2, $3, $4, $5}' | cut -c1-150
