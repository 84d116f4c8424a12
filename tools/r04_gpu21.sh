#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr -o t -f csv -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py 65536 > $GRAFT_REPO_ROOT/gpurun_out/r04_qp_trace.log 2>&1 || exit 1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/bulk_launch_rates.py $f > $GRAFT_REPO_ROOT/gpurun_out/r04_bulk_launch_rates.txt 2>&1 || exit 1
head -30 $GRAFT_REPO_ROOT/gpurun_out/r04_bulk_launch_rates.txt
grep it1 $GRAFT_REPO_ROOT/gpurun_out/r04_qp_trace.log | cut -c1-200
