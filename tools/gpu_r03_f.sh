#!/bin/bash
# round 3, sixth call: full suite (tail kernel, timeout fallback, aggregation fix), setup/upload profile, C2 + C1 with the tail kernel
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_t8.log 2>&1
echo "full gpu suite rc=$?" | tee -a gpurun_out/r03_t8.log
tail -4 gpurun_out/r03_t8.log
PAMG_TIMING=1 timeout 600 python tools/host_profile.py --grid 256 256 256 --smoother gs --top 16 > gpurun_out/r03_hostprof_256_e.log 2>&1
echo "host profile rc=$?"; grep -n "^setup\|^upload\|fit_candidates" gpurun_out/r03_hostprof_256_e.log | cut -c1-160
for wl in c2 c1; do
timeout 600 python bench.py --workload $wl --no-extras --no-setup-compare --no-pmc --steps 50 --warmup 5 > gpurun_out/r03_bench_${wl}_tail.json 2> gpurun_out/r03_bench_${wl}_tail.err
PAMG_TAIL=0 timeout 600 python bench.py --workload $wl --no-extras --no-setup-compare --no-pmc --steps 50 --warmup 5 --cpu-cycles 0 > gpurun_out/r03_bench_${wl}_notail.json 2> gpurun_out/r03_bench_${wl}_notail.err
python - <<PY
import json
for t in ("tail","notail"):
    d=json.loads([l for l in open('gpurun_out/r03_bench_${wl}_%s.json'%t) if l.startswith('{')][-1])
    print("${wl}", t, d['value'], d['ms_per_step'], (d.get('parity') or {}).get('reference_protocol',{}).get('max_rel_diff'))
PY
done
