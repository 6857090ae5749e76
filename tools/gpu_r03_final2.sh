#!/bin/bash
# round-3 evidence, final: the whole GPU suite, the default bench line (with the 512^3 leg), rocprofv3 kernel stats of the C3 / C2 /
# C5 / C4 (512^3) bench commands, and the 2-rank rehearsal of the N > 1 line (two ranks on the one GPU, gloo)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests.log 2>&1; tail -3 gpurun_out/r03_gpu_tests.log
timeout 1500 python bench.py > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r03_bench_n1.json
prof() {   # workload tag extra-args
    OUT=$PWD/gpurun_out/prof_r03_$2; mkdir -p $OUT
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --workload $1 --steps 10 --warmup 2 --cpu-cycles 0 --no-extras --no-pmc --no-setup-compare $3 > $OUT/trace_bench.json 2> $OUT/trace.log)
    python tools/summarize_prof.py $OUT $1 r03 > $OUT/summarize.log 2>&1
    find $OUT -name "*.csv" -size +4M -delete
    head -8 $OUT/kernel_stats_summary.txt
}
prof c3 c3
prof c2 c2
prof c4x c4x "--steps 5"
PAMG_BENCH_BACKEND=gloo PAMG_BENCH_ONE_GPU=1 PAMG_SHARD_WORKLOAD=c4s timeout 900 python bench.py --gpus 2 --no-extras > gpurun_out/r03_bench_2rank_gloo.json 2> gpurun_out/r03_bench_2rank_gloo.err
echo "2-rank rehearsal rc=$?"
