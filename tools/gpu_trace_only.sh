#!/bin/bash
# kernel-trace stats only: bash tools/gpu_trace_only.sh <workload> <tag> [steps]
WL=${1:-c3}; TAG=${2:-r01}; STEPS=${3:-3}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
CMD="python bench.py --workload $WL --steps $STEPS --warmup 1 --cpu-cycles 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace_bench.json 2> $OUT/trace.log
echo "trace exit $?" >> $OUT/trace.log
python tools/summarize_prof.py $OUT $WL $TAG
find $OUT -name "*.csv" -size +8M -delete
cat $OUT/trace_bench.json
