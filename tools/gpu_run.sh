#!/bin/bash
# The one GPU-box launcher (replaces the per-experiment tools/gpu_r0N_x.sh of earlier rounds):
#   gpurun --timeout T -- 'bash tools/gpu_run.sh <step> [<step> ...]'
# Steps run in order; each writes under gpurun_out/ with the TAG prefix (env TAG, default r04).
#   tests[:expr]      GPU suite (optionally -k expr)            -> ${TAG}_gpu_tests.log
#   py:<script+args>  python tools/<script> args (',' = space)  -> ${TAG}_<script>.log
#   bench[:args]      python bench.py args (',' = space)        -> ${TAG}_bench<suffix>.json/.err
#   prof:<workload>[:args]  rocprofv3 kernel stats of bench.py --workload W -> prof_${TAG}_<W>/  (+ per-kernel roofline table)
#   dist2[:args]            bench.py --gpus 2 as two ranks sharing the one GPU (gloo transport): rehearsal of the N > 1 line
#   pmc:<workload>[:args]   two more passes with --pmc FETCH_SIZE / --pmc WRITE_SIZE (kernel trace only) into the same directory
#   smoke             __graft_entry__.smoke()
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
TAG=${TAG:-r04}
for step in "$@"; do
    kind=${step%%:*}; rest=""; [[ "$step" == *:* ]] && rest=${step#*:}
    case $kind in
        tests)
            if [ -n "$rest" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "${rest//,/ }" > gpurun_out/${TAG}_gpu_tests_k.log 2>&1; tail -5 gpurun_out/${TAG}_gpu_tests_k.log
            else timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -5 gpurun_out/${TAG}_gpu_tests.log; fi ;;
        py)
            args=${rest//,/ }; name=$(echo "$args" | awk '{print $1}'); name=${name%.py}
            suffix=${PYTAG:-}
            timeout ${PYTIMEOUT:-900} python tools/$args > gpurun_out/${TAG}_${name}${suffix}.log 2>&1; echo "$name rc=$?"; tail -${PYTAIL:-40} gpurun_out/${TAG}_${name}${suffix}.log ;;
        bench)
            args=${rest//,/ }; suffix=${BENCHTAG:-_n1}
            timeout 1500 python bench.py $args > gpurun_out/${TAG}_bench${suffix}.json 2> gpurun_out/${TAG}_bench${suffix}.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/${TAG}_bench${suffix}.json ;;
        dist2)
            # two ranks on the ONE GPU, gloo host-callback transport: rehearsal of the N > 1 line (the rate means nothing)
            args=${rest//,/ }
            PAMG_BENCH_BACKEND=gloo PAMG_BENCH_ONE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
                bench.py --gpus 2 --steps 5 --warmup 2 $args > gpurun_out/${TAG}_bench_2ranks_one_gpu.json 2> gpurun_out/${TAG}_bench_2ranks_one_gpu.err; echo "dist2 rc=$?"; tail -c 2500 gpurun_out/${TAG}_bench_2ranks_one_gpu.json ;;
        prof)
            wl=${rest%%:*}; extra=""; [[ "$rest" == *:* ]] && extra=${rest#*:}; extra=${extra//,/ }
            OUT=$PWD/gpurun_out/prof_${TAG}_$wl; mkdir -p $OUT
            (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /root/repo/bench.py --workload $wl --steps 10 --warmup 2 --cpu-cycles 0 --no-extras --no-pmc --no-setup-compare --kernel-map $OUT/kernel_map.json $extra > $OUT/trace_bench.json 2> $OUT/trace.log)
            python tools/summarize_prof.py $OUT $wl $TAG > $OUT/summarize.log 2>&1
            find $OUT -name "*.csv" -size +4M -delete
            head -12 $OUT/kernel_stats_summary.txt; head -30 $OUT/kernel_roofline.txt ;;
        pmc)
            wl=${rest%%:*}; extra=""; [[ "$rest" == *:* ]] && extra=${rest#*:}; extra=${extra//,/ }
            OUT=$PWD/gpurun_out/prof_${TAG}_$wl; mkdir -p $OUT
            for c in FETCH_SIZE WRITE_SIZE; do
                sub=pmc_fetch; [ $c = WRITE_SIZE ] && sub=pmc_write
                (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$sub -- python /root/repo/bench.py --workload $wl --steps 5 --warmup 1 --cpu-cycles 0 --no-extras --no-pmc --no-setup-compare $extra > $OUT/${sub}_bench.json 2> $OUT/$sub.log)
            done
            python tools/summarize_prof.py $OUT $wl $TAG > $OUT/summarize.log 2>&1
            find $OUT -name "*.csv" -size +4M -delete
            tail -30 $OUT/summarize.log ;;
        smoke)
            timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -4 ;;
        *) echo "unknown step $step" ;;
    esac
done
