#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command + separate PMC passes (HBM bytes).
# usage: bash tools/gpu_profile.sh <workload> <tag>
WL=${1:-c2}; TAG=${2:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
CMD="python bench.py --workload $WL --steps 10 --warmup 2 --cpu-cycles 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.log
echo "trace exit $?" >> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.log
echo "pmc fetch exit $?" >> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.log
echo "pmc write exit $?" >> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -20
python tools/summarize_prof.py $OUT $WL $TAG
# keep only summaries small enough to merge back
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*counter_collection.csv" -size +20M -delete
du -sh $OUT
