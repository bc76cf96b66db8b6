set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_neighbors_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/r2b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b_pytest.log
tail -8 gpurun_out/r2b_pytest.log
for occ in 0.5 0.75 1.0 1.5; do
  echo "== occ $occ" >> gpurun_out/r2b_knn.log
  SPT_KNN_OCC=$occ timeout 300 python tools/knn_bench.py S 0 2 >> gpurun_out/r2b_knn.log 2>&1
  SPT_KNN_OCC=$occ timeout 300 python tools/knn_bench.py D 0 2 >> gpurun_out/r2b_knn.log 2>&1
done
cat gpurun_out/r2b_knn.log
