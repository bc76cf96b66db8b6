cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_attention_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/attn_microbench.py --mode 2 --packed 1 --reps 4 2>&1 | tail -3
cp superpoint_transformer_amd/lib/libspt_hip.so /tmp/lib_base.so
cp gpurun_variants/lib_attnprof.so superpoint_transformer_amd/lib/libspt_hip.so
timeout 200 python tools/attn_microbench.py --mode 2 --packed 1 --reps 1 2>&1 | tail -2
cp /tmp/lib_base.so superpoint_transformer_amd/lib/libspt_hip.so
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | cut -c100-200
