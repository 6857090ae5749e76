#!/bin/bash
# round 3, first contact: the new sharded driver + setup tests, then the full default bench line (with the 512^3 leg)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dist.py tests/test_gpu_setup.py -m gpu -x -q > gpurun_out/r03_t1.log 2>&1
echo "dist+setup tests rc=$?" >> gpurun_out/r03_t1.log
tail -5 gpurun_out/r03_t1.log
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/r03_t2.log 2>&1
echo "solver+kernel tests rc=$?" >> gpurun_out/r03_t2.log
tail -5 gpurun_out/r03_t2.log
timeout 1700 python bench.py --no-setup-compare > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err
echo "bench rc=$?"
tail -30 gpurun_out/r03_bench_a.err
