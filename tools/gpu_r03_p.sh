#!/bin/bash
export TMPDIR=/tmp
cd /tmp
timeout 1500 python $GRAFT_REPO_ROOT/tools/pmc_stall_probe.py 2>&1 | tail -6
