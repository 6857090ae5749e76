#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "value_codes" > gpurun_out/r03_t17.log 2>&1; tail -15 gpurun_out/r03_t17.log
timeout 600 python tools/microbench_spmv_val8.py 2>&1 | tail -16
