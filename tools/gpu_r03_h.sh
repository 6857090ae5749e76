#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py -m gpu -x -q > gpurun_out/r03_t10.log 2>&1
echo "kernel+solver suite rc=$?" | tee -a gpurun_out/r03_t10.log
tail -3 gpurun_out/r03_t10.log
timeout 700 python tools/microbench_gs2.py --exp c --levels 1 2 3 --tag gs2c_256 2>&1 | grep variant
