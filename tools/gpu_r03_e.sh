#!/bin/bash
# round 3, fifth call: token-passing aggregation, per-schedule range size, relaxation coarse solvers; headline only
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_t7.log 2>&1
echo "full gpu suite rc=$?" | tee -a gpurun_out/r03_t7.log
tail -4 gpurun_out/r03_t7.log
PAMG_TIMING=1 timeout 600 python tools/host_profile.py --grid 256 256 256 --smoother gs --top 14 > gpurun_out/r03_hostprof_256_d.log 2>&1
echo "host profile rc=$?"; grep -n "^setup\|^upload\|standard_aggregation\|fit_candidates" gpurun_out/r03_hostprof_256_d.log | cut -c1-160
timeout 900 python bench.py --no-extras --no-setup-compare --no-pmc > gpurun_out/r03_bench_c3_b.json 2> gpurun_out/r03_bench_c3_b.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_bench_c3_b.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['host'], d['parity']['reference_protocol']['max_rel_diff'])
print([(s['level'], s['ms_per_forward_sweep'], s['us_per_dependency_level']) for s in d['gs_sweeps']['per_level']])
PY
