#!/bin/bash
# round 3, third call: new setup tests, RCCL binding (subprocess), upload after the host-side parallelisation, C5 under device prolongation smoothing, 512^3 setup profile
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dist.py tests/test_gpu_setup.py -m gpu -x -q -k "rccl or block_prolongation or block_difference or unit_outer or device_setup or prolongation" > gpurun_out/r03_t5.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03_t5.log
tail -4 gpurun_out/r03_t5.log
PAMG_TIMING=1 timeout 600 python tools/host_profile.py --grid 256 256 256 --smoother gs --top 12 > gpurun_out/r03_hostprof_256_b.log 2>&1
echo "host profile rc=$?"; grep -n "^setup\|^upload" gpurun_out/r03_hostprof_256_b.log
timeout 600 python bench.py --workload c5 --no-extras --no-setup-compare --steps 30 > gpurun_out/r03_bench_c5.json 2> gpurun_out/r03_bench_c5.err
echo "c5 rc=$?"; tail -3 gpurun_out/r03_bench_c5.err; head -c 600 gpurun_out/r03_bench_c5.json
PAMG_TIMING=1 timeout 900 python tools/host_profile.py --grid 512 512 512 --smoother cheby --top 25 > gpurun_out/r03_hostprof_512.log 2>&1
echo "512 profile rc=$?"; grep -n "^setup\|^upload" gpurun_out/r03_hostprof_512.log
