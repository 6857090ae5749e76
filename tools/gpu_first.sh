#!/bin/bash
# first GPU contact: environment facts, parity tests, micro-benchmarks (run via gpurun)
mkdir -p gpurun_out
{
  echo "== env"; nproc; free -g | head -2; rocm-smi --showproductname 2>/dev/null | head -8
  ls oracle/_ref/pyamg | head -3
  python -c "from pyamg_amd import _capi as c; print(c.version(), c.device_count())"
  echo "== torch coexistence (torch first, then lib first)"
  timeout 300 python - <<'PY'
import torch, numpy as np
print("torch", torch.__version__, torch.cuda.is_available())
from pyamg_amd import _capi as c
from pyamg_amd.multilevel import DeviceMatrix
from pyamg_amd.hierarchy import sparse_op
import scipy.sparse as sp
A = sp.random(500,500,density=0.02,format='csr',random_state=1)+sp.eye_array(500,format='csr'); A=sp.csr_array(A); A.sort_indices()
x=np.random.rand(500); dA=DeviceMatrix(sparse_op(A)); dx=c.DeviceArray.from_host(x); dy=c.DeviceArray(500,np.float64)
dA.spmv(0,dx,dy); print("torch-first spmv exact:", np.array_equal(dy.download(), A@x))
t = torch.ones(10, device='cuda'); print("torch cuda ok", float(t.sum()))
PY
  timeout 300 python - <<'PY'
import numpy as np
from pyamg_amd import _capi as c
from pyamg_amd.multilevel import DeviceMatrix
from pyamg_amd.hierarchy import sparse_op
import scipy.sparse as sp
A = sp.random(500,500,density=0.02,format='csr',random_state=1)+sp.eye_array(500,format='csr'); A=sp.csr_array(A); A.sort_indices()
x=np.random.rand(500); dA=DeviceMatrix(sparse_op(A)); dx=c.DeviceArray.from_host(x); dy=c.DeviceArray(500,np.float64)
dA.spmv(0,dx,dy); print("lib-first spmv exact:", np.array_equal(dy.download(), A@x))
import torch
t = torch.ones(10, device='cuda'); print("lib-first torch cuda ok", float(t.sum()))
PY
} > gpurun_out/first_env.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/first_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/first_pytest.log
timeout 900 python tools/microbench.py --sweep --tag r01_first > gpurun_out/first_microbench.log 2>&1
echo "microbench exit $?" >> gpurun_out/first_microbench.log
tail -5 gpurun_out/first_env.log; tail -15 gpurun_out/first_pytest.log; tail -20 gpurun_out/first_microbench.log
