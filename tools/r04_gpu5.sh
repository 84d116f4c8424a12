#!/bin/bash
# round-4 GPU call 5: alt-path tests; same-box A/Bs (fused bulk launch, panel overlap); rank slices after the stream fix; API profile
mkdir -p gpurun_out
L=gpurun_out/r04_gputest_5.log
timeout -k 10 900 python -m pytest tests/test_gpu_core.py tests/test_gpu_dist.py -m gpu -x -q -k "alternative or virtual_ranks or fused or hand_offs or world_of_one" > $L 2>&1
rc=$?; echo "pytest rc=$rc" >> $L; grep "passed\|failed\|rc=" $L
[ $rc -eq 0 ] || exit 1
A=gpurun_out/r04_fused_bulk_ab.txt; : > $A
for rep in 1 2; do
  for f in 1 0; do echo "# TGP_FUSED_BULK=$f" >> $A; TGP_FUSED_BULK=$f timeout -k 10 200 python tools/quick_perf.py 65536 32768 2>&1 | grep "it1" >> $A || exit 1; done
done
cat $A
B=gpurun_out/r04_panel_overlap_ab.txt; : > $B
for rep in 1 2; do
  for f in 0 1; do echo "# TGP_PANEL_OVERLAP=$f" >> $B; TGP_PANEL_OVERLAP=$f timeout -k 10 200 python tools/quick_perf.py 4096 8192 12288 16384 2>&1 | grep "it1" | cut -c1-150 >> $B || exit 1; done
done
cat $B
R=gpurun_out/r04_rank_slice_5.txt; : > $R
run() { echo "# $*" >> $R; env "$@" timeout -k 10 150 python tools/rank_slice.py $N $G $g 2>&1 | grep "N=" >> $R || exit 1; }
N=65536; G=8; g=7
run TGP_DIST_QUEUE=0 TGP_DIST_FINISH=0
run TGP_DIST_QUEUE=0 TGP_DIST_FINISH=32
run TGP_DIST_QUEUE=0 TGP_DIST_FINISH=16
run TGP_DIST_QUEUE=-1 TGP_DIST_FINISH=32
run TGP_DIST_QUEUE=-1 TGP_DIST_FINISH=0
cat $R
timeout -k 10 200 python tools/api_overhead.py 65536 > gpurun_out/r04_api_overhead_65536.txt 2>&1; head -45 gpurun_out/r04_api_overhead_65536.txt
