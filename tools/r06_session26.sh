#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py -x -q -m gpu -k "schwarz or timeout" > gpurun_out/s26_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s26_tests.log
timeout 900 python tools/microbench_schwarz.py --tag schwarz_persistent 2>&1 | tail -4
PAMG_SCHWARZ_LEVELS=1 timeout 900 python tools/microbench_schwarz.py --no-kernel --tag schwarz_level_launches 2>&1 | tail -3
