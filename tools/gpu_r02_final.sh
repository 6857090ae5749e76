#!/bin/bash
# round-2 final evidence: full GPU test suite, the default bench line, rocprofv3 kernel stats of the C3 bench command and
# of the device setup (tools/setup_bench.py 128^3)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_final.log 2>&1; tail -3 gpurun_out/gpu_tests_final.log
timeout 330 python bench.py > gpurun_out/bench_r02_final.log 2>gpurun_out/bench_r02_final.err; tail -1 gpurun_out/bench_r02_final.log | cut -c1-600
OUT=$PWD/gpurun_out/prof_r02b_c3; mkdir -p $OUT
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --workload c3 --steps 10 --warmup 2 --cpu-cycles 0 --no-extras --no-pmc --no-setup-compare > $OUT/trace_bench.json 2> $OUT/trace.log)
python tools/summarize_prof.py $OUT c3 r02b > $OUT/summarize.log 2>&1
find $OUT -name "*.csv" -size +4M -delete
head -14 $OUT/kernel_stats_summary.txt
OUT=$PWD/gpurun_out/prof_r02b_setup; mkdir -p $OUT
(cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/tools/setup_bench.py --grid 128 128 128 > $OUT/setup_bench.json 2> $OUT/trace.log)
python tools/summarize_prof.py $OUT setup r02b > $OUT/summarize.log 2>&1
find $OUT -name "*.csv" -size +4M -delete
head -14 $OUT/kernel_stats_summary.txt
