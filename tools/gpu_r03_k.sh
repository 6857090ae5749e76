#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 1200 python $GRAFT_REPO_ROOT/tools/pmc_variants.py 2>&1 | tail -12
