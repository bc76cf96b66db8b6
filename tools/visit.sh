set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_neighbors_gpu.py -x -q -k "dense_to_csr or scatter_pca" > gpurun_out/r2d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d_pytest.log
timeout 600 python -m pytest "tests/test_fullsize_gpu.py::test_attention_two_formulations_agree_at_scene_scale" -x -q >> gpurun_out/r2d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d_pytest.log
tail -12 gpurun_out/r2d_pytest.log
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
echo "== base mode 1 (f32 mfma)" > gpurun_out/r2d_attn_variants.log
timeout 200 python tools/attn_microbench.py --reps 3 --mode 1 2>&1 | tail -2 >> gpurun_out/r2d_attn_variants.log
for f in /tmp/lib_base.so gpurun_variants/*.so; do
  cp $f superpoint_transformer_amd/lib/libspt_hip.so
  echo "== $f (mode 2)" >> gpurun_out/r2d_attn_variants.log
  timeout 200 python tools/attn_microbench.py --reps 3 2>&1 | tail -2 >> gpurun_out/r2d_attn_variants.log
done
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
cat gpurun_out/r2d_attn_variants.log
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/pmc_attn -- python $GRAFT_REPO_ROOT/tools/attn_microbench.py --reps 2 > /tmp/pmc_attn.log 2>&1)
python tools/pmc_query.py /tmp/pmc_attn "%attn%" > gpurun_out/r2d_pmc_attn.txt 2>&1
cat gpurun_out/r2d_pmc_attn.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_step -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess > /tmp/prof_step.log 2>&1)
db=$(find /tmp/prof_step -name "*.db" | head -1)
python tools/rocpd_summary.py $db > gpurun_out/r2d_spt64_trainstep_sceneS_kernel_stats.csv
head -30 gpurun_out/r2d_spt64_trainstep_sceneS_kernel_stats.csv | cut -c1-120,200-280
