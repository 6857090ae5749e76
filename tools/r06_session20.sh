cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r06 bash tools/gpu_run.sh tests dist2 smoke
python bench.py > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo "bench rc=$?"; wc -c gpurun_out/r06_bench_n1.json; cp gpurun_out/bench_detail.json gpurun_out/r06_bench_detail_n1.json; python -c "
import json; d=json.load(open('gpurun_out/r06_bench_n1.json')); print({k:v for k,v in d.items() if not isinstance(v,dict)}); print(d['roofline']); print(d['host'])"
