#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for cfg in "0 1 0" "0 1 1" "0 0 0" "0 1 0"; do
  set -- $cfg
  PAMG_PLAN_HUGEPAGES=$2 PAMG_PLAN_PREFAULT=$3 PAMG_TIMING=1 python tools/profile_host_cost.py 256 > gpurun_out/host_cost_x.log 2>&1
  echo "== huge=$2 prefault=$3 $(grep "== setup\|== upload" gpurun_out/host_cost_x.log | tr "\n" " ")"
  grep "threads,.*windows" gpurun_out/host_cost_x.log | tail -2 | cut -c20-220
  grep "build_lanem_part" gpurun_out/host_cost_x.log | tail -2 | tr '\n' ';'; echo
done
