#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_kernels.py -m gpu -q -k "cg or gmres or cgne or cgnr or block" > gpurun_out/r03_t13.log 2>&1; tail -40 gpurun_out/r03_t13.log
