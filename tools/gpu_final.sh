#!/bin/bash
# round-end evidence run: GPU tests, smoke, default bench, rocprofv3 kernel trace of the same bench command
# (PMC passes: tools/gpu_pmc.sh on the Jacobi workload -- rocprofv3 --pmc segfaults on the 40k-launch GS run)
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/${TAG}_smoke.log
timeout 1200 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.log; echo "bench exit $?" >> gpurun_out/${TAG}_bench_n1.log
OUT=$PWD/gpurun_out/prof_${TAG}_c3
mkdir -p $OUT
CMD="python bench.py --steps 5 --warmup 1 --cpu-cycles 0 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.log; echo "trace exit $?" >> $OUT/trace.log
python tools/summarize_prof.py $OUT c3 $TAG > $OUT/summarize.log 2>&1
find $OUT -name "*.csv" -size +4M -delete
tail -3 gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_smoke.log; cat gpurun_out/${TAG}_bench_n1.json | head -c 1500; echo; head -16 $OUT/kernel_stats_summary.txt
