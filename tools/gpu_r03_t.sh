#!/bin/bash
export TMPDIR=/tmp
cd /tmp
timeout 200 python $GRAFT_REPO_ROOT/tools/spmv_pmc_level1.py 2>&1 | tail -2
timeout 900 python $GRAFT_REPO_ROOT/tools/pmc_stall_probe.py --level1 2>&1 | tail -3
