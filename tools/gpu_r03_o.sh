#!/bin/bash
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null > $GRAFT_REPO_ROOT/gpurun_out/r03_counters_avail.txt || rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/r03_counters_avail.txt 2>&1
wc -l $GRAFT_REPO_ROOT/gpurun_out/r03_counters_avail.txt
grep -o -E "\b(TA_[A-Z_a-z0-9]+|TCP_[A-Z_a-z0-9]+|SQ_[A-Z_a-z0-9]+|TCC_[A-Z_a-z0-9]+|GRBM_[A-Z_a-z0-9]+|[A-Za-z]+Busy|[A-Za-z]+Stalled|MemUnit[A-Za-z]+)\b" $GRAFT_REPO_ROOT/gpurun_out/r03_counters_avail.txt | sort -u | tr '\n' ' ' | head -c 6000
