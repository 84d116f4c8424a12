#!/bin/bash
mkdir -p gpurun_out
R=gpurun_out/r04_chain_latency_ab2.txt; : > $R
run() { echo "# TGP_TRSM_SMALL_TILES=$1 TGP_DIST_HEAD_HALF=$2 : $3 $4 $5" >> $R; TGP_TRSM_SMALL_TILES=$1 TGP_DIST_HEAD_HALF=$2 timeout -k 10 200 python tools/rank_slice.py $3 $4 $5 2>&1 | grep "N=" >> $R || exit 1; }
for rep in 1 2; do
  run 0 0 65536 8 7
  run 16 256 65536 8 7
  run 24 256 65536 8 7
  run 32 256 65536 8 7
  run 48 256 65536 8 7
  run 32 1024 65536 8 7
done
run 16 256 32768 8 7
run 32 256 32768 8 7
run 32 1024 32768 8 7
run 64 1024 32768 8 7
run 0 0 16384 4 3
run 32 256 16384 4 3
run 64 256 16384 4 3
run 32 256 65536 4 3
run 32 256 65536 2 1
run 0 0 65536 2 1
cut -c1-150 $R
