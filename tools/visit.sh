cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_neighbors_gpu.py -m gpu -x -q -k "geometric or geof or pipeline or visiting" 2>&1 | tail -3
timeout 300 python tools/knn_bench.py S 0 3 2>&1 | grep "^scene"
timeout 300 python tools/knn_bench.py D 0 2 2>&1 | grep "^scene"
