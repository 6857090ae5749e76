#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_val8.log 2>&1; tail -3 gpurun_out/r03_gpu_tests_val8.log
timeout 600 python bench.py --workload c4s --no-extras --cpu-cycles 1 --no-setup-compare > gpurun_out/r03_val8_c4s.json 2> gpurun_out/r03_val8_c4s.err; echo "c4s rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_val8_c4s.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:900], d['parity'].get('max_rel_diff'))
PY
timeout 600 python bench.py --workload c2 --no-extras --cpu-cycles 1 --no-setup-compare --steps 50 > gpurun_out/r03_val8_c2.json 2> gpurun_out/r03_val8_c2.err; echo "c2 rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_val8_c2.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:600], d['parity'].get('max_rel_diff'))
PY
