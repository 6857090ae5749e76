cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for rnd in 1 2; do
for lib in pyamg_amd/libpyamg_amd.so pyamg_amd/build/ab/libpyamg_amd_forcevc.so; do
PAMG_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python tools/microbench_sa_ops.py --idx16 1 --tag r06_sa_ops_$(basename $lib .so)_$rnd > gpurun_out/r06_sa_ops_$(basename $lib .so)_$rnd.log 2>&1; echo "== $lib round $rnd"; grep -v "^setup" gpurun_out/r06_sa_ops_$(basename $lib .so)_$rnd.log | tail -14
done; done
PAMG_TIMING=1 python bench.py --no-extras --no-pmc --no-setup-compare --cpu-cycles 0 > gpurun_out/r06_bench_c3_timing.json 2> gpurun_out/r06_bench_c3_timing.err; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_c3_timing.json')); print(d['ms_per_step'], d['host'])"; grep -i "lanem\|line\|lane" gpurun_out/r06_bench_c3_timing.err | head -30
