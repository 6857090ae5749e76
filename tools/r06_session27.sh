#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for w in 0; do
  echo "== waves $w"; PAMG_SCHWARZ_WAVES=$w timeout 600 python tools/microbench_schwarz.py --no-cycle --tag schwarz_w$w 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('persistent_ms_per_sweep','level_launches_ms_per_sweep','us_per_dependency_level_persistent','bit_identical_between_schedulers','persistent_error_word')})"
done
PAMG_SCHWARZ_WAVES=0 timeout 600 python tools/microbench_schwarz.py --no-cycle --grid 64 64 64 --tag schwarz_3d 2>&1 | tail -1
