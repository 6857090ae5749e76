cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "merged or fast_order or timeout" > gpurun_out/r06_tests_merged.log 2>&1; tail -3 gpurun_out/r06_tests_merged.log
timeout 900 python tools/microbench_lanem.py --levels 2 3 --s 1 3 4 6 --grids 0 --xcd 1 2 --tag r06_lanem_xcd > gpurun_out/r06_microbench_lanem_xcd.log 2>&1; grep -o '^[0-9] \|"s": [0-9].*"ms_forward": [0-9.]*\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_xcd.log | paste - - - | head -40
python bench.py --no-extras --no-pmc --no-setup-compare --cpu-cycles 3 > gpurun_out/r06_bench_c3_merged.json 2> gpurun_out/r06_bench_c3_merged.err; echo rc=$?; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_c3_merged.json')); print({k:v for k,v in d.items() if k in ('value','ms_per_step') or k.startswith('gs_sweep') or k.startswith('accel')}); print(d['parity'], d['host'])"
