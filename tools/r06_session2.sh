cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "merged or fast_order" > gpurun_out/r06_tests_merged.log 2>&1; tail -15 gpurun_out/r06_tests_merged.log
timeout 1500 python tools/microbench_lanem.py --levels 1 2 3 --s 1 2 3 4 5 6 --ahead 23 46 --tag r06_lanem_first > gpurun_out/r06_microbench_lanem_first.log 2>&1; tail -70 gpurun_out/r06_microbench_lanem_first.log
