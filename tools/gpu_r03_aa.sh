#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PAMG_BENCH_BACKEND=gloo PAMG_BENCH_ONE_GPU=1 PAMG_SHARD_WORKLOAD=c4s timeout 600 python bench.py --gpus 2 --no-extras > gpurun_out/r03_bench_2rank_gloo.json 2> gpurun_out/r03_bench_2rank_gloo.err
echo "2-rank rehearsal rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_bench_2rank_gloo.json') if l.strip().startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['parity'])
PY
timeout 200 python -m pytest tests/test_dist.py tests/test_gpu_kernels.py -m gpu -x -q -k "value_codes or sharded" 2>&1 | tail -2
