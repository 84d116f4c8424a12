#!/bin/bash
mkdir -p gpurun_out
R=gpurun_out/r04_strips_ab.txt; : > $R
run() { echo "# TGP_DIST_STRIPS=$1 : $2 $3 $4" >> $R; TGP_DIST_STRIPS=$1 timeout -k 10 200 python tools/rank_slice.py $2 $3 $4 2>&1 | grep "N=" >> $R || exit 1; }
for rep in 1 2 3; do run left 65536 8 7; run right 65536 8 7; done
run left 32768 8 7; run right 32768 8 7
run left 16384 4 3; run right 16384 4 3
run left 65536 4 3; run right 65536 4 3
cut -c1-240 $R
TGP_DIST_STRIPS=right timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/r04_gputest_19.log 2>&1; echo "pytest (right) rc=$?"; tail -2 gpurun_out/r04_gputest_19.log
