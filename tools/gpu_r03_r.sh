#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "value_codes or spmv or jacobi" > gpurun_out/r03_t16.log 2>&1; tail -3 gpurun_out/r03_t16.log
for wl in c1 c2; do
timeout 600 python bench.py --workload $wl --no-extras --cpu-cycles 1 --no-setup-compare --steps 50 --no-pmc > gpurun_out/r03_rows_$wl.json 2> gpurun_out/r03_rows_$wl.err; echo "$wl rc=$?"; python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r03_rows_$wl.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['parity'].get('max_rel_diff'))
PY
done
