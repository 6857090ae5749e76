cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "merged" > gpurun_out/r06_tests_merged.log 2>&1; tail -3 gpurun_out/r06_tests_merged.log
for d in 0 16; do echo dbg=$d
PAMG_LANEM_DBG=$d timeout 900 python tools/microbench_lanem.py --levels 1 2 --s 2 3 --grids 512 768 1024 1536 --tag r06_lanem_plain$d > gpurun_out/r06_microbench_lanem_plain$d.log 2>&1; grep -o '^[0-9] \|"s": [0-9].*"ms_forward": [0-9.]*\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_plain$d.log | paste - - - | head -60
done
