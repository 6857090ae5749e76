#!/bin/bash
# renumbered interior levels: GPU tests of the new path, then the 512^3 Chebyshev cycle with and without (one session, one box)
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_solver.py -x -q -m gpu -k "renumbered or live_reference" > gpurun_out/s23_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/s23_tests.log
for r in 0 1 0 1; do
  PAMG_RENUMBER=$r python bench.py --workload c4x --no-extras --no-model --no-pmc --cpu-cycles 0 --no-setup-compare --steps 20 --warmup 3 > gpurun_out/s23_c4x_renumber_$r.json 2> gpurun_out/s23_c4x_renumber_$r.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/s23_c4x_renumber_$r.json').read().strip().splitlines()[-1])
print('renumber=$r', d['ms_per_step'], d['value'], {k:v for k,v in d.items() if 'parity' in k}, d.get('host'))
PY
done
