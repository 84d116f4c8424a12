#!/bin/bash
# round-4 GPU call 1: new hand-off / fused-launch tests first (short leash), then the whole GPU suite, then the rank slices
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "hand_offs or fused_update or update_entry" > gpurun_out/r04_gputest_0.log 2>&1
rc=$?; echo "pytest0 rc=$rc" >> gpurun_out/r04_gputest_0.log; tail -5 gpurun_out/r04_gputest_0.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_1.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r04_gputest_1.log; tail -5 gpurun_out/r04_gputest_1.log
[ $rc -eq 0 ] || exit 1
R=gpurun_out/r04_rank_slice_1.txt; : > $R
for g in 0 7; do timeout -k 10 120 python tools/rank_slice.py 65536 8 $g >> $R 2>&1 || exit 1; done
echo "# TGP_DIST_FUSED=0" >> $R; TGP_DIST_FUSED=0 timeout -k 10 120 python tools/rank_slice.py 65536 8 7 >> $R 2>&1 || exit 1
echo "# TGP_DIST_QUEUE=0" >> $R; TGP_DIST_QUEUE=0 timeout -k 10 120 python tools/rank_slice.py 65536 8 7 >> $R 2>&1 || exit 1
echo "# TGP_DIST_QUEUE=0 launches" >> $R; TGP_DIST_QUEUE=0 timeout -k 10 120 python tools/rank_slice.py 65536 8 0 launches >> $R 2>&1 || exit 1
timeout -k 10 120 python tools/rank_slice.py 65536 1 0 >> $R 2>&1 || exit 1
grep "N=\|^#" $R
