cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "merged" > gpurun_out/r06_tests_merged.log 2>&1; tail -3 gpurun_out/r06_tests_merged.log
timeout 1500 python tools/microbench_lanem.py --levels 1 --s 2 3 4 --rpw 1 2 --grids 0 512 768 1024 --tag r06_lanem_rpw > gpurun_out/r06_microbench_lanem_rpw.log 2>&1; grep -o '^[0-9] \|"s": [0-9].*"ms_forward": [0-9.]*\|"max_rel[^,]*,\|"units_per_row": [0-9.]*\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_rpw.log | paste - - - - - | head -60
