#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for o in exact fast; do
  PAMG_GS_ORDER=$o python bench.py --workload c5p --no-extras --no-model --no-pmc --no-setup-compare --steps 20 --warmup 3 > gpurun_out/s28_c5p_$o.json 2> gpurun_out/s28_c5p_$o.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/s28_c5p_$o.json').read().strip().splitlines()[-1])
print('$o', d['ms_per_step'], d['value'], d.get('cpu_baseline_value'), {k:v for k,v in d.items() if k.startswith('parity')})
PY
done
timeout 900 python tools/microbench_schwarz.py --tag schwarz_persistent 2>&1 | tail -2
PAMG_SCHWARZ_LEVELS=1 timeout 900 python tools/microbench_schwarz.py --no-kernel --tag schwarz_level_launches 2>&1 | tail -1
timeout 600 python tools/microbench_schwarz.py --no-cycle --grid 64 64 64 --tag schwarz_3d 2>&1 | tail -1
