cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "merged or fast_order or midsize" 2>&1 | tail -2
for sw in forward backward; do echo "== $sw"
timeout 600 python tools/microbench_lanem.py --levels 1 2 3 --s 0 --ahead 40 --grids 0 --sweep $sw --tag r06_lanem_final_${sw} > gpurun_out/r06_microbench_lanem_final_${sw}.log 2>&1; grep -o '^[0-9] \|"s": [0-9].*"ms_forward": [0-9.]*\|"hand_offs": [0-9]*\|"units_per_row": [0-9.]*\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_final_${sw}.log | paste - - - - - 
done
python bench.py --no-extras --no-pmc --no-setup-compare --cpu-cycles 3 > gpurun_out/r06_bench_c3_v3.json 2> gpurun_out/r06_bench_c3_v3.err; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_c3_v3.json')); print({k:v for k,v in d.items() if k in ('value','ms_per_step') or k.startswith('gs_sweep_ms')}); print(d['parity'], d['host'])"
