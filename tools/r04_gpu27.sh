#!/bin/bash
mkdir -p gpurun_out
R=gpurun_out/r04_segs_tpw_ab.txt; : > $R
for rep in 1 2; do for t in 1 2 4; do
  echo "# TGP_SEGS_TPW=$t" >> $R; TGP_SEGS_TPW=$t timeout -k 10 200 python tools/quick_perf.py 65536 2>&1 | grep "it1" | cut -c1-260 >> $R || exit 1
done; done
for t in 1 2; do echo "# TGP_SEGS_TPW=$t N=32768" >> $R; TGP_SEGS_TPW=$t timeout -k 10 200 python tools/quick_perf.py 32768 2>&1 | grep "it1" | cut -c1-260 >> $R || exit 1; done
cat $R
