#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PAMG_BENCH_BACKEND=gloo PAMG_BENCH_ONE_GPU=1 timeout 1100 python bench.py --gpus 2 --no-extras --steps 5 --warmup 1 > gpurun_out/r03_bench_2rank_gloo_c4x.json 2> gpurun_out/r03_bench_2rank_gloo_c4x.err
echo "2-rank 512^3 rehearsal rc=$?"; tail -c 1800 gpurun_out/r03_bench_2rank_gloo_c4x.json; grep -E "bench\]|Error|error|Traceback" gpurun_out/r03_bench_2rank_gloo_c4x.err | tail -12
