cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_subedges.py tests/test_h5io.py -x -q > gpurun_out/r2j_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j_pytest.log
tail -15 gpurun_out/r2j_pytest.log
