#!/bin/bash
# round 3, second call: re-run the tests that failed / are new, host-side profiles (setup + upload), 2-rank rehearsal of the N > 1 bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dist.py -m gpu -x -q > gpurun_out/r03_t3.log 2>&1
echo "dist tests rc=$?" | tee -a gpurun_out/r03_t3.log
tail -4 gpurun_out/r03_t3.log
timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_kernels.py tests/test_gpu_setup.py -m gpu -x -q -k "long_dependency or pinv or unit_outer" > gpurun_out/r03_t4.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r03_t4.log
tail -4 gpurun_out/r03_t4.log
PAMG_TIMING=1 timeout 600 python tools/host_profile.py --grid 256 256 256 --smoother gs > gpurun_out/r03_hostprof_256.log 2>&1
echo "host profile rc=$?"
PAMG_BENCH_BACKEND=gloo PAMG_BENCH_ONE_GPU=1 PAMG_SHARD_WORKLOAD=c4s timeout 900 python bench.py --gpus 2 --no-extras > gpurun_out/r03_bench_2rank_gloo.json 2> gpurun_out/r03_bench_2rank_gloo.err
echo "2-rank rehearsal rc=$?"
tail -5 gpurun_out/r03_bench_2rank_gloo.err
head -c 1500 gpurun_out/r03_bench_2rank_gloo.json
