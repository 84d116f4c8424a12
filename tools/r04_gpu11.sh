#!/bin/bash
# round-4 GPU call 11: hand-offs published by the last kernel in front of them (TGP_TAIL_SIGNAL): tests, same-box A/B, soak
mkdir -p gpurun_out
L=gpurun_out/r04_gputest_11.log
timeout -k 10 400 python -m pytest tests/test_gpu_core.py tests/test_gpu_edge_cases.py tests/test_gpu_soak.py -m gpu -x -q > $L 2>&1
rc=$?; echo "pytest rc=$rc" >> $L; grep "passed\|failed\|rc=" $L
[ $rc -eq 0 ] || exit 1
A=gpurun_out/r04_tail_signal_ab.txt; : > $A
for rep in 1 2 3; do
  for f in 0 1; do echo "# TGP_TAIL_SIGNAL=$f" >> $A; TGP_TAIL_SIGNAL=$f timeout -k 10 200 python tools/quick_perf.py 2048 3072 4096 6144 8192 12288 16384 24576 2>&1 | grep "it1" | cut -c1-100 >> $A || exit 1; done
done
cat $A
timeout -k 10 300 python tools/soak_handoffs.py 4 > gpurun_out/r04_soak_tail.txt 2>&1; echo "soak rc=$?"; tail -4 gpurun_out/r04_soak_tail.txt
