#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py tests/test_dist.py -m gpu -x -q > gpurun_out/r03_t18.log 2>&1; tail -6 gpurun_out/r03_t18.log
for wl in c4s c2 c1; do
timeout 600 python bench.py --workload $wl --no-extras --cpu-cycles 1 --no-setup-compare --steps 30 --no-pmc > gpurun_out/r03_rowpat_$wl.json 2> gpurun_out/r03_rowpat_$wl.err; echo "$wl rc=$?"; python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r03_rowpat_$wl.json') if l.startswith('{')][-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['ms_per_launch'], r['frac'], d['parity'].get('max_rel_diff'))
PY
done
