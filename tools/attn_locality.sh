cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for P in 2 1; do for LOC in 0 2000 64; do
  rm -rf /tmp/pv; (cd /tmp && rocprofv3 --kernel-trace -d /tmp/pv -o v -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py --packed $P --reps 3 --local $LOC > /dev/null 2>&1)
  echo "== packed=$P local=$LOC"; python tools/rocpd_top.py /tmp/pv 12 | grep "attn_bwd\|kv_reduce\|attn_fwd"
done; done
