cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r02z_bench_sceneS.json 2> gpurun_out/r02z_bench_sceneS.err; cut -c1-300 gpurun_out/r02z_bench_sceneS.json
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_step -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess > /tmp/prof_step.log 2>&1)
tail -1 /tmp/prof_step.log | cut -c1-200
python tools/rocpd_summary.py /tmp/prof_step > gpurun_out/r02z_spt64_trainstep_sceneS_kernel_stats.csv
timeout 200 python bench.py --scene T --steps 30 --warmup 5 --no-cpu-baseline --no-preprocess 2>/dev/null > gpurun_out/r02z_bench_sceneT.json; cut -c100-200 gpurun_out/r02z_bench_sceneT.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_T -- python $GRAFT_REPO_ROOT/bench.py --scene T --steps 10 --warmup 3 --no-cpu-baseline --no-preprocess > /tmp/prof_T.log 2>&1)
python tools/rocpd_summary.py /tmp/prof_T > gpurun_out/r02z_spt64_trainstep_sceneT_kernel_stats.csv
SPT_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --scene T --steps 5 --warmup 2 --no-cpu-baseline --no-preprocess 2>/dev/null | cut -c1-200
