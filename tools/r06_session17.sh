cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.log 2>&1; tail -12 gpurun_out/r06_gpu_tests.log
