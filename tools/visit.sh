cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
for f in /tmp/lib_base.so gpurun_variants/*.so /tmp/lib_base.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  echo "== $f" >> gpurun_out/r2i_segcsr_variants.txt
  python bench.py --stages scatter --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('scatter chain: ms/step', d['ms_per_step'], 'segmax ms', r['ms_per_launch'], 'GB/s', r['achieved'], 'frac', r['frac'])" >> gpurun_out/r2i_segcsr_variants.txt
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-preprocess 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('train step   : ms/step', d['ms_per_step'], 'segmax ms', r['ms_per_launch'], 'GB/s', r['achieved'], 'frac', r['frac'])" >> gpurun_out/r2i_segcsr_variants.txt
done
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
cat gpurun_out/r2i_segcsr_variants.txt
