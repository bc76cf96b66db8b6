cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -3
