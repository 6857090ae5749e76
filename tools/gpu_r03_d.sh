#!/bin/bash
# round 3, fourth call: the WHOLE gpu suite, host-side timings after f4 + planner threads, GS range-geometry sweep on the SA levels
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_t6.log 2>&1
echo "full gpu suite rc=$?" | tee -a gpurun_out/r03_t6.log
tail -4 gpurun_out/r03_t6.log
PAMG_TIMING=1 timeout 600 python tools/host_profile.py --grid 256 256 256 --smoother gs --top 22 > gpurun_out/r03_hostprof_256_c.log 2>&1
echo "host profile rc=$?"; grep -n "^setup\|^upload" gpurun_out/r03_hostprof_256_c.log
timeout 900 python tools/microbench_gs2.py --tag gs2_256 > gpurun_out/r03_microbench_gs2.log 2>&1
echo "gs2 rc=$?"; grep -c variant gpurun_out/r03_microbench_gs2.log
