set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_setup.py -x -q 2>&1 | tail -30 > gpurun_out/setup_tests.log
cat gpurun_out/setup_tests.log
G=${SETUP_GRID:-128}
timeout 900 python tools/setup_bench.py --grid $G $G $G > gpurun_out/setup_bench_$G.json 2> gpurun_out/setup_bench_$G.err
cat gpurun_out/setup_bench_$G.json; tail -5 gpurun_out/setup_bench_$G.err
