#!/bin/bash
mkdir -p gpurun_out
R=gpurun_out/r04_gj16_group_ab.txt; : > $R
run() { echo "# asm groups of $1 : N=$2" >> $R; if [ $1 = 8 ]; then unset TGP_LIB_PATH; else export TGP_LIB_PATH=$PWD/treegp_amd/csrc/libtgp_g$1.so; fi; timeout -k 10 200 python tools/quick_perf.py $2 2>&1 | grep "it1" | cut -c1-150 >> $R || exit 1; }
for rep in 1 2 3; do for g in 8 4 2; do run $g 8192; done; done
for g in 8 4 2; do run $g 2048; done
cat $R
