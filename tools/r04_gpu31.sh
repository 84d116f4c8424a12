#!/bin/bash
mkdir -p gpurun_out
R=gpurun_out/r04_chain_prio_ab.txt; : > $R
run() { echo "# $1 : N=$2" >> $R; if [ $1 = noprio ]; then export TGP_LIB_PATH=$PWD/treegp_amd/csrc/libtgp_noprio.so; else unset TGP_LIB_PATH; fi; timeout -k 10 200 python tools/quick_perf.py $2 2>&1 | grep "it1" | cut -c1-200 >> $R || exit 1; }
for rep in 1 2 3; do run prio 65536; run noprio 65536; done
for n in 8192 32768; do run prio $n; run noprio $n; done
cat $R
