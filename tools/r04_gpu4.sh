#!/bin/bash
# round-4 GPU call 4: tests of the changed paths; KB=32 A/B; rank slices with half tiles / replicated finish / adaptive rule;
# API host-tax profile; the bench line
mkdir -p gpurun_out
L=gpurun_out/r04_gputest_4.log
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_api_dist.py -m gpu -x -q -s > $L 2>&1
rc=$?; echo "pytest rc=$rc" >> $L; grep "N=\|passed\|failed\|rc=" $L
[ $rc -eq 0 ] || exit 1
A=gpurun_out/r04_kb32_ab.txt; : > $A
for rep in 1 2; do
  echo "# default (KB=16)" >> $A; timeout -k 10 200 python tools/quick_perf.py 65536 32768 2>&1 | grep "it1" >> $A || exit 1
  echo "# TGP_KB32" >> $A; TGP_LIB_PATH=$PWD/treegp_amd/csrc/libtgp_kb32.so timeout -k 10 200 python tools/quick_perf.py 65536 32768 2>&1 | grep "it1" >> $A || exit 1
done
cat $A
R=gpurun_out/r04_rank_slice_4.txt; : > $R
run() { echo "# $*" >> $R; env "$@" timeout -k 10 150 python tools/rank_slice.py $N $G $g 2>&1 | grep "N=" >> $R || exit 1; }
N=65536; G=8; g=7
run TGP_DIST_QUEUE=-1
run TGP_DIST_QUEUE=0
run TGP_DIST_QUEUE=0 TGP_DIST_HALF_TILES=0
run TGP_DIST_QUEUE=0 TGP_DIST_FINISH=0
run TGP_DIST_QUEUE=0 TGP_DIST_FINISH=16
run TGP_DIST_QUEUE=0 TGP_DIST_FINISH=48
run TGP_DIST_QUEUE=0 TGP_DIST_HALF_TILES=64
g=0
run TGP_DIST_QUEUE=-1
G=1
run TGP_DIST_QUEUE=-1
cat $R
timeout -k 10 200 python tools/api_overhead.py 8192 > gpurun_out/r04_api_overhead_8192.txt 2>&1; head -40 gpurun_out/r04_api_overhead_8192.txt
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err; tail -c 1500 gpurun_out/r04_bench_a.json
