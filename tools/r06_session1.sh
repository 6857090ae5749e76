cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python tools/ab_csr_stream.py > gpurun_out/r06_ab_csr_stream.json 2> gpurun_out/r06_ab_csr_stream.err; tail -30 gpurun_out/r06_ab_csr_stream.json
TAG=r06 bash tools/gpu_run.sh tests:timeout,or,kaczmarz
python bench.py --no-pmc > gpurun_out/r06_bench_first.json 2> gpurun_out/r06_bench_first.err; echo bench rc=$?; wc -c gpurun_out/r06_bench_first.json; cat gpurun_out/r06_bench_first.json; cp gpurun_out/bench_detail.json gpurun_out/r06_bench_first_detail.json
