#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr -o t -f csv -- python3 $GRAFT_REPO_ROOT/tools/rank_slice.py 65536 8 7 > $GRAFT_REPO_ROOT/gpurun_out/r04_tl.log 2>&1 || exit 1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/dist_timeline.py $f 0 4.0 > $GRAFT_REPO_ROOT/gpurun_out/r04_dist_timeline_end.txt 2>&1 || exit 1
grep "N=" $GRAFT_REPO_ROOT/gpurun_out/r04_tl.log | cut -c1-120
