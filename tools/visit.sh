cd $GRAFT_REPO_ROOT
for v in "" "SPT_DBG_NO_NARROW=1" "SPT_DBG_NO_B1=1" "SPT_DBG_NO_DW=1" ""; do
  echo "== $v"; env $v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | cut -c100-160
done
