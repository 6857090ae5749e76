cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "merged" > gpurun_out/r06_tests_merged.log 2>&1; tail -5 gpurun_out/r06_tests_merged.log
timeout 1500 python tools/microbench_lanem.py --levels 1 --s 2 3 --grids 256 512 768 1024 1536 2048 4096 --tag r06_lanem_grid > gpurun_out/r06_microbench_lanem_grid.log 2>&1; grep -o '"s": [0-9].*"ms_forward": [0-9.]*\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_grid.log | paste - - | head -40
