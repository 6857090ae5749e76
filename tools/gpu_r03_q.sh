#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py tests/test_dist.py -m gpu -x -q > gpurun_out/r03_t15.log 2>&1; tail -5 gpurun_out/r03_t15.log
timeout 600 python tools/microbench_spmv_val8.py 2>&1 | tail -16
for wl in c4s c2; do
timeout 600 python bench.py --workload $wl --no-extras --cpu-cycles 1 --no-setup-compare --steps 30 > gpurun_out/r03_rowg_$wl.json 2> gpurun_out/r03_rowg_$wl.err; echo "$wl rc=$?"; python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r03_rowg_$wl.json') if l.startswith('{')][-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['ms_per_launch'], r['frac'], r.get('traffic_over_algorithmic'), r.get('frac_on_streamed_bytes'), r.get('values_streamed_as_stored'), d['parity'].get('max_rel_diff'))
PY
done
