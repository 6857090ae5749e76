#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dist.py -m gpu -q > gpurun_out/r03_t14.log 2>&1; tail -30 gpurun_out/r03_t14.log
PAMG_BENCH_BACKEND=gloo PAMG_BENCH_ONE_GPU=1 PAMG_SHARD_WORKLOAD=c4s timeout 900 python bench.py --gpus 2 --no-extras > gpurun_out/r03_bench_2rank_gloo.json 2> gpurun_out/r03_bench_2rank_gloo.err
echo "2-rank rehearsal rc=$?"; tail -c 1500 gpurun_out/r03_bench_2rank_gloo.json; tail -5 gpurun_out/r03_bench_2rank_gloo.err
