#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag
free -g | head -2
for h in 0 1 0 1; do
  PAMG_PLAN_HUGEPAGES=$h PAMG_TIMING=1 python tools/profile_host_cost.py 256 > gpurun_out/host_cost_h$h.log 2>&1
  echo "== hugepages=$h"; grep "== setup\|== upload" gpurun_out/host_cost_h$h.log
  grep "build_lanem_part\|build_line_part\|merged rows\|slots filled" gpurun_out/host_cost_h$h.log | awk '{t=$(NF-2); if (t+0 > 0.3 || $0 ~ /merged rows|slots/) print}' | sort | uniq -c | sort -k1,1nr | awk '$NF!="" {print}' | tail -12
done
