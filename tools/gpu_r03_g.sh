#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_t9.log 2>&1
echo "full gpu suite rc=$?" | tee -a gpurun_out/r03_t9.log
tail -4 gpurun_out/r03_t9.log
PAMG_TIMING=0 timeout 600 python tools/host_profile.py --grid 256 256 256 --smoother gs --top 25 > gpurun_out/r03_hostprof_256_f.log 2>&1
echo "host profile rc=$?"; grep -n "^setup\|^upload" gpurun_out/r03_hostprof_256_f.log
