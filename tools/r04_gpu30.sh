#!/bin/bash
mkdir -p gpurun_out
R=gpurun_out/r04_group8_ab.txt; : > $R
for rep in 1 2 3; do for g in 999999999 49152; do
  echo "# TGP_GROUP8_FROM=$g" >> $R; TGP_GROUP8_FROM=$g timeout -k 10 200 python tools/quick_perf.py 65536 2>&1 | grep "it1" | cut -c1-260 >> $R || exit 1
done; done
for g in 999999999 49152; do echo "# TGP_GROUP8_FROM=$g N=49152" >> $R; TGP_GROUP8_FROM=$g timeout -k 10 200 python tools/quick_perf.py 49152 2>&1 | grep "it1" | cut -c1-260 >> $R || exit 1; done
for g in 999999999 32768; do echo "# TGP_GROUP8_FROM=$g N=32768" >> $R; TGP_GROUP8_FROM=$g timeout -k 10 200 python tools/quick_perf.py 32768 2>&1 | grep "it1" | cut -c1-260 >> $R || exit 1; done
cat $R
