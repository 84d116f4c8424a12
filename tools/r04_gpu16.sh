#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_api_dist.py -m gpu -x -q > gpurun_out/r04_gputest_16.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r04_gputest_16.log
[ $rc -eq 0 ] || exit 1
R=gpurun_out/r04_chain_latency_ab.txt; : > $R
run() { echo "# TGP_TRSM_SMALL_TILES=$1 TGP_DIST_HEAD_HALF=$2 : $3 $4 $5" >> $R; TGP_TRSM_SMALL_TILES=$1 TGP_DIST_HEAD_HALF=$2 timeout -k 10 200 python tools/rank_slice.py $3 $4 $5 2>&1 | grep "N=" >> $R || exit 1; }
for rep in 1 2; do
  run 0 0 65536 8 7
  run 64 0 65536 8 7
  run 0 256 65536 8 7
  run 64 256 65536 8 7
done
run 64 512 65536 8 7
run 64 1024 65536 8 7
run 0 0 32768 8 7
run 64 256 32768 8 7
run 0 0 65536 4 3
run 64 256 65536 4 3
run 128 256 65536 4 3
cut -c1-250 $R
