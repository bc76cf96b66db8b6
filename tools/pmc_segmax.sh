# PMC FETCH_SIZE / WRITE_SIZE of the stand-alone north-star kernel (tools/segmax_bench.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for G in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmcm_$G
  (cd /tmp && rocprofv3 --pmc $G -d /tmp/pmcm_$G -o p -- python $GRAFT_REPO_ROOT/tools/segmax_bench.py --reps 2 > /tmp/pmcm_$G.log 2>&1)
  python tools/pmc_query.py /tmp/pmcm_$G "%segmax_stream%" 2>&1 | grep -v "^no .db" | cut -c1-220
done
