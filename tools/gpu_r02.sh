#!/bin/bash
# round-2 evidence: rocprofv3 kernel-trace stats of the C3 and C2 bench commands, the full default bench line, the c4 leg
export TMPDIR=/tmp
mkdir -p gpurun_out
for WL in c3 c2; do
  OUT=$PWD/gpurun_out/prof_r02_${WL}
  mkdir -p $OUT
  CMD="python bench.py --workload $WL --steps 10 --warmup 2 --cpu-cycles 0 --no-extras --no-pmc"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --workload $WL --steps 10 --warmup 2 --cpu-cycles 0 --no-extras --no-pmc > $OUT/trace_bench.json 2> $OUT/trace.log)
  echo "trace exit $?" >> $OUT/trace.log
  python tools/summarize_prof.py $OUT $WL r02 > $OUT/summarize.log 2>&1
  find $OUT -name "*.csv" -size +4M -delete
  head -16 $OUT/kernel_stats_summary.txt
done
timeout 900 python bench.py > gpurun_out/bench_r02_full.log 2>&1; tail -1 gpurun_out/bench_r02_full.log | cut -c1-400
timeout 900 python bench.py --workload c4 --no-extras --cpu-cycles 1 --steps 10 > gpurun_out/bench_r02_c4.log 2>&1; tail -1 gpurun_out/bench_r02_c4.log | cut -c1-1500
