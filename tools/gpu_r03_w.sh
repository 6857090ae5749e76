#!/bin/bash
export TMPDIR=/tmp
cd /tmp
timeout 600 python $GRAFT_REPO_ROOT/tools/pmc_stall_probe.py "--val8=1" 2>&1 | tail -2
