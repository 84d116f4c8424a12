#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr -o t -f csv -- python3 $GRAFT_REPO_ROOT/tools/rank_slice.py 65536 8 7 > $GRAFT_REPO_ROOT/gpurun_out/r04_tl.log 2>&1 || exit 1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/dist_timeline.py $f 100 4.0 > $GRAFT_REPO_ROOT/gpurun_out/r04_dist_timeline_end2.txt 2>&1 || exit 1
grep "N=" $GRAFT_REPO_ROOT/gpurun_out/r04_tl.log | cut -c1-220
cd $GRAFT_REPO_ROOT && timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_api_dist.py tests/test_gpu_core.py -m gpu -x -q > gpurun_out/r04_gputest_18.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04_gputest_18.log
