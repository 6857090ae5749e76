cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/ab_csr_stream.py head pyamg_amd/build/ab/libpyamg_amd_novc.so pyamg_amd/build/ab/libpyamg_amd_r05head.so > gpurun_out/r06_ab_csr_stream_novc.json 2> gpurun_out/r06_ab_csr_stream_novc.err; grep "^round" gpurun_out/r06_ab_csr_stream_novc.json
timeout 1500 python -m pytest tests -m gpu -x -q -k "solver or cheb or poly or dist" > gpurun_out/r06_tests_solver.log 2>&1; tail -5 gpurun_out/r06_tests_solver.log
