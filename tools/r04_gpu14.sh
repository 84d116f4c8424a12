#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for ov in 0 1; do
  rm -rf /tmp/tr$ov
  TGP_DIST_OVERLAP=$ov timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr$ov -o t -f csv -- python3 $GRAFT_REPO_ROOT/tools/rank_slice.py 65536 8 7 > $GRAFT_REPO_ROOT/gpurun_out/r04_tl_$ov.log 2>&1 || exit 1
  f=$(find /tmp/tr$ov -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/dist_timeline.py $f > $GRAFT_REPO_ROOT/gpurun_out/r04_dist_timeline_ov$ov.txt 2>&1 || exit 1
  grep "N=" $GRAFT_REPO_ROOT/gpurun_out/r04_tl_$ov.log | cut -c1-120
  tail -4 $GRAFT_REPO_ROOT/gpurun_out/r04_dist_timeline_ov$ov.txt
done
