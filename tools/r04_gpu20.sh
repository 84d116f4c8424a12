#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_api_dist.py -m gpu -x -q > gpurun_out/r04_gputest_20.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/r04_gputest_20.log
[ $rc -eq 0 ] || exit 1
R=gpurun_out/r04_chain_fold_ab.txt; : > $R
run() { echo "# $1 : $2 $3 $4" >> $R; if [ $1 = prev ]; then export TGP_LIB_PATH=$PWD/treegp_amd/csrc/libtgp_prev.so; else unset TGP_LIB_PATH; fi; timeout -k 10 200 python tools/rank_slice.py $2 $3 $4 2>&1 | grep "N=" >> $R || exit 1; }
for rep in 1 2 3; do run prev 65536 8 7; run new 65536 8 7; done
run prev 32768 8 7; run new 32768 8 7
run prev 32768 8 0; run new 32768 8 0
run prev 16384 4 3; run new 16384 4 3
cut -c1-240 $R
