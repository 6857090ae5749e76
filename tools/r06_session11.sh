cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
export PAMG_LANEM_DBG=16
timeout 1500 python tools/pmc_lane_probe.py 1 '{"lane_merge":1}' r06_unmerged > gpurun_out/r06_pmc_unmerged.log 2>&1
timeout 1500 python tools/pmc_lane_probe.py 1 '{"lane_merge":2,"lane_G":768}' r06_merged_s2 > gpurun_out/r06_pmc_merged_s2.log 2>&1
timeout 1500 python tools/pmc_lane_probe.py 1 '{"lane_merge":3,"lane_G":768}' r06_merged_s3 > gpurun_out/r06_pmc_merged_s3.log 2>&1
python - <<'P'
import json
for t in ("r06_unmerged","r06_merged_s2","r06_merged_s3"):
    d=json.load(open(f"gpurun_out/pmc_lane_probe_{t}.json"))
    for k,v in d.items():
        if isinstance(v,dict) and "gs_lane" in k and "TCP_TCC_READ_REQ_sum" in v:
            print(t,k[:60], {q:v.get(q) for q in ("GRBM_GUI_ACTIVE","TCP_TCC_READ_REQ_sum","TCP_TCC_WRITE_REQ_sum","TCP_TOTAL_CACHE_ACCESSES_sum","TCP_TCC_READ_REQ_LATENCY_sum","TCC_HIT_sum","TCC_MISS_sum","TCC_EA0_RDREQ_sum","SQ_INSTS_VMEM_RD","SQ_INSTS_VALU","TCP_PENDING_STALL_CYCLES_sum")})
P
