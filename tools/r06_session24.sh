#!/bin/bash
# per-kernel times of the 512^3 Chebyshev cycle with and without renumbered interior levels
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 0 1; do
  rm -rf gpurun_out/s24_prof_$r
  PAMG_RENUMBER=$r rocprofv3 --kernel-trace --stats -d gpurun_out/s24_prof_$r -o c4x -- python bench.py --workload c4x --no-extras --no-model --no-pmc --cpu-cycles 0 --no-setup-compare --protocol-cycles 0 --steps 10 --warmup 2 > gpurun_out/s24_c4x_$r.json 2> gpurun_out/s24_c4x_$r.err
  f=$(find gpurun_out/s24_prof_$r -name '*kernel_stats.csv' | head -1)
  echo "== renumber=$r  $f"
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if int(r['Calls'])>=10]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:22]:
    print(f"{r['Name'][:110]:110s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.2f}")
PY
  find gpurun_out/s24_prof_$r -name '*kernel_trace.csv' -size +60M -delete
done
