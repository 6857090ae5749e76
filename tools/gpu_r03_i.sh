#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_setup.py -m gpu -q > gpurun_out/r03_t11.log 2>&1
echo "setup suite rc=$?" | tee -a gpurun_out/r03_t11.log
tail -3 gpurun_out/r03_t11.log
