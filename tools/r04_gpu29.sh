#!/bin/bash
mkdir -p gpurun_out
export TGP_LIB_PATH=$PWD/treegp_amd/csrc/libtgp_seg8.so
R=gpurun_out/r04_depth2048_probe.txt; : > $R
for rep in 1 2; do for d in 4 8; do
  echo "# TGP_DEBUG_SEGS=$d (depth $((d*256)))" >> $R
  TGP_DEBUG_SEGS=$d timeout -k 10 200 python tools/syrk_loop.py 49152 12 2>&1 | grep "chunk [1-4]" >> $R || exit 1
done; done
cat $R
