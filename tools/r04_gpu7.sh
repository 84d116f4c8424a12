#!/bin/bash
# round-4 GPU call 7: clear-CU count sweep for the chain-bound sizes (forced TGP_QUEUE_RES), same box
mkdir -p gpurun_out
A=gpurun_out/r04_queue_res_sweep.txt; : > $A
for r in 0 1 2 3 4 5 6 0; do
  echo "# TGP_QUEUE_RES=$r (0 = the rule)" >> $A; TGP_QUEUE_RES=$r timeout -k 10 200 python tools/quick_perf.py 3072 4096 6144 8192 12288 16384 2>&1 | grep "it1" | cut -c1-110 >> $A || exit 1
done
cat $A
