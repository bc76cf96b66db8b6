#!/bin/bash
# A/B the prebuilt library variants of gpurun_variants/ on the scatter-chain bench
cd $GRAFT_REPO_ROOT
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
for f in /tmp/lib_base.so gpurun_variants/*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  echo "== $f"
  python bench.py --stages scatter --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('ms/step', d['ms_per_step'], 'segmax ms', r['ms_per_launch'], 'GB/s', r['achieved'], 'frac', r['frac'])"
done
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
