cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -6
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | cut -c100-200
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
cp gpurun_variants/lib_fmlpprof.so superpoint_transformer_amd/lib/libspt_hip.so
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-preprocess 2>&1 | grep "fmlp bwd" | sort | uniq -c | head
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
