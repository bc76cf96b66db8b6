cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | cut -c100-200
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | cut -c100-200
