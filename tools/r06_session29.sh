#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
lscpu | grep -i "model name\|socket\|numa\|thread\|core" | head -12
for t in 8 16 32 64 96 128; do
  PAMG_PLAN_THREADS=$t PAMG_PLAN_HUGEPAGES=1 PAMG_TIMING=1 python tools/profile_host_cost.py 256 > gpurun_out/host_cost_t$t.log 2>&1
  echo "== threads=$t $(grep '== upload' gpurun_out/host_cost_t$t.log)"
  grep "merged rows\|slots filled\|build_lanem_part" gpurun_out/host_cost_t$t.log | awk '{t=$(NF-1); if ($0 ~ /build_lanem/) t=$(NF-2); if (t+0 > 0.25) print}' | tr -s ' ' | tr '\n' ';'; echo
done
