#!/bin/bash
# does the GPU clock down during latency-bound sweeps?  can we pin it?
mkdir -p gpurun_out
{
echo "== clocks idle"; rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk" | head -6
echo "== perf level"; rocm-smi --showperflevel 2>&1 | tail -4
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 1; rocm-smi --showclocks 2>&1 | grep -E "sclk" | head -1; done ) > gpurun_out/clock_trace.log 2>&1 &
timeout 300 python tools/microbench_gs.py --grid 96 96 96 --tag clk0 --check 0 2>&1 | grep -E "launch|gran128|flow1 " 
wait
echo "== clock samples during run"; cat gpurun_out/clock_trace.log
echo "== try perflevel high"; rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showclocks 2>&1 | grep -E "sclk" | head -2
timeout 300 python tools/microbench_gs.py --grid 96 96 96 --tag clk1 --check 0 2>&1 | grep -E "launch|gran128|flow1 "
echo "== try determinism"; rocm-smi --setperfdeterminism 2100 2>&1 | tail -3
timeout 300 python tools/microbench_gs.py --grid 96 96 96 --tag clk2 --check 0 2>&1 | grep -E "launch|gran128|flow1 "
rocm-smi --resetperfdeterminism 2>&1 | tail -1; rocm-smi --setperflevel auto 2>&1 | tail -1
} 2>&1 | tee gpurun_out/clock_exp.log
