cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
export PAMG_LANEM_SIMPLE=2
timeout 900 python -m pytest tests -m gpu -x -q -k "merged" > gpurun_out/r06_tests_merged.log 2>&1; tail -3 gpurun_out/r06_tests_merged.log
timeout 1500 python tools/microbench_lanem.py --levels 1 2 3 --s 2 3 4 --grids 0 512 768 1024 1536 --ahead 23 --tag r06_lanem_pub > gpurun_out/r06_microbench_lanem_pub.log 2>&1; grep -o '^[0-9] \|"s": [0-9].*"ms_forward": [0-9.]*\|"max_rel[^,]*,\|"grid": [0-9]*' gpurun_out/r06_microbench_lanem_pub.log | paste - - - - | head -60
