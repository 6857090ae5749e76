#!/bin/bash
# PMC passes (HBM bytes) on the bandwidth-bound legs: c3j = 256^3 hierarchy with the Jacobi smoother
# (few launches -- rocprofv3 --pmc segfaulted on the 40k-launch Gauss-Seidel workload)
TAG=${1:-r01}; WL=${2:-c3j}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
CMD="python bench.py --workload $WL --steps 5 --warmup 1 --cpu-cycles 0 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.log; echo "trace exit $?" >> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.log; echo "exit $?" >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.log; echo "exit $?" >> $OUT/pmc_write.log
python tools/summarize_prof.py $OUT $WL $TAG > $OUT/summarize.log 2>&1
find $OUT -name "*.csv" -size +4M -delete
head -14 $OUT/kernel_stats_summary.txt; tail -40 $OUT/summarize.log
