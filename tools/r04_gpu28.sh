#!/bin/bash
mkdir -p gpurun_out
R=gpurun_out/r04_hw_queues_ab.txt; : > $R
for rep in 1 2; do for q in default 8 2; do
  echo "# GPU_MAX_HW_QUEUES=$q" >> $R
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 200 python tools/rank_slice.py 65536 8 7 2>&1 | grep "N=" | cut -c1-240 >> $R || exit 1
done; done
for q in default 8; do
  echo "# GPU_MAX_HW_QUEUES=$q (single-GPU path)" >> $R
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 200 python tools/quick_perf.py 65536 2>&1 | grep "it1" | cut -c1-200 >> $R || exit 1
  timeout -k 10 200 python tools/quick_perf.py 8192 2>&1 | grep "it1" | cut -c1-200 >> $R || exit 1
done
cat $R
