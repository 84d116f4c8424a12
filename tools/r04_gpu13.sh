#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_api_dist.py -m gpu -x -q > gpurun_out/r04_gputest_13.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r04_gputest_13.log
[ $rc -eq 0 ] || exit 1
R=gpurun_out/r04_overlap_ab.txt; : > $R
for rep in 1 2; do
  for ov in 0 1; do
    echo "# TGP_DIST_OVERLAP=$ov" >> $R; TGP_DIST_OVERLAP=$ov timeout -k 10 150 python tools/rank_slice.py 65536 8 7 2>&1 | grep "N=" >> $R || exit 1
  done
done
for G in 4 2; do for ov in 0 1; do
    echo "# TGP_DIST_OVERLAP=$ov" >> $R; TGP_DIST_OVERLAP=$ov timeout -k 10 200 python tools/rank_slice.py 65536 $G $((G-1)) 2>&1 | grep "N=" >> $R || exit 1
done; done
cut -c1-250 $R
