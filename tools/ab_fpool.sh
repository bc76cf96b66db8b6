cd $GRAFT_REPO_ROOT
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
for round in 1 2; do
for f in /tmp/lib_base.so gpurun_variants/lib_*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  echo "== $f"; python tools/fpool_bench.py --mode 1 --reps 8 | tail -2
  python tools/fpool_bench.py --mode 1 --reps 8 --order sorted | tail -2
done
done
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
