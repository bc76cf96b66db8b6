cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -8
timeout 200 python tools/attn_microbench.py --mode 2 --packed 1 2>&1 | tail -4
timeout 200 python tools/attn_microbench.py --mode 2 --packed 0 2>&1 | tail -2
for sh in 1 0; do
  echo "share=$sh"
  SPT_EA_GRAD_SHARE=$sh timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess 2>/dev/null | cut -c100-200
done
