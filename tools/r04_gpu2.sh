#!/bin/bash
# round-4 GPU call 2: independent-solver forward-error test; rank-slice sweeps over group size / queue rule
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "independent" > gpurun_out/r04_gputest_2.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r04_gputest_2.log; grep "N=\|passed\|failed\|rc=" gpurun_out/r04_gputest_2.log
R=gpurun_out/r04_rank_slice_2.txt; : > $R
run() { echo "# $*" >> $R; env "$@" timeout -k 10 150 python tools/rank_slice.py $N $G $g 2>&1 | grep "N=" >> $R || exit 1; }
N=65536; G=8; g=7
run TGP_DIST_GROUP=2 TGP_DIST_QUEUE=0
run TGP_DIST_GROUP=2 TGP_DIST_QUEUE=-1
run TGP_DIST_GROUP=3 TGP_DIST_QUEUE=0
run TGP_DIST_GROUP=4 TGP_DIST_QUEUE=0
run TGP_DIST_GROUP=4 TGP_DIST_QUEUE=0 TGP_DIST_FUSED=0
G=4; g=3
run TGP_DIST_GROUP=4 TGP_DIST_QUEUE=0
run TGP_DIST_GROUP=4 TGP_DIST_QUEUE=-1
G=2; g=1
run TGP_DIST_GROUP=4 TGP_DIST_QUEUE=-1
N=131072; G=8; g=7
run TGP_DIST_GROUP=4 TGP_DIST_QUEUE=-1
cat $R
