# the round's final pass on the GPU box: suite, two-rank rehearsal, smoke, rocprof tables of four workloads, the default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r06 bash tools/gpu_run.sh tests dist2 smoke prof:c3 prof:c2 prof:c4x prof:c5 2>&1 | grep -v "^csr_\|^gs_\|^bsr_\|^  \|^kernel \[" | tail -30
python bench.py > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; echo "bench rc=$?"; wc -c gpurun_out/r06_bench_n1.json; cp gpurun_out/bench_detail.json gpurun_out/r06_bench_detail_n1.json
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_n1.json')); print({k:v for k,v in d.items() if k in ('value','ms_per_step','extra_c4_ms','extra_c2_ms','extra_c5_block_gauss_seidel_ms','extra_c1_ms','extra_c6n3_ms','extra_c8s_ms','extra_c8s_cpu_cycles_per_s','extra_c8s_cycles_per_s','modelled_ms_n2','modelled_ms_n4','modelled_ms_n8','accel_cg_s_to_tol','seconds_to_tol_1e-8') or k.startswith('gs_sweep_ms')}); print(d['roofline']['general_csr_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['host'], d['parity'])"
