#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "value_codes" > gpurun_out/r03_t12.log 2>&1
echo "val8 tests rc=$?" | tee -a gpurun_out/r03_t12.log
tail -5 gpurun_out/r03_t12.log
timeout 600 python tools/microbench_spmv_val8.py 2>&1 | tail -40
