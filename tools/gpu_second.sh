#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --workload c2 --steps 20 --warmup 3 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.log; echo "bench c2 exit $?" >> gpurun_out/bench_c2.log
timeout 1500 python bench.py --workload c3 --steps 10 --warmup 2 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.log; echo "bench c3 exit $?" >> gpurun_out/bench_c3.log
tail -8 gpurun_out/pytest_gpu.log; tail -5 gpurun_out/smoke.log; tail -4 gpurun_out/bench_c2.log; cat gpurun_out/bench_c2.json; tail -4 gpurun_out/bench_c3.log; cat gpurun_out/bench_c3.json
