#!/bin/bash
mkdir -p gpurun_out
R=gpurun_out/r04_lds_stride_ab.txt; : > $R
run() { echo "# $1 : N=$2" >> $R; if [ $1 = stride18 ]; then export TGP_LIB_PATH=$PWD/treegp_amd/csrc/libtgp_prev.so; else unset TGP_LIB_PATH; fi; timeout -k 10 200 python tools/quick_perf.py $2 2>&1 | grep "it1" | cut -c1-200 >> $R || exit 1; }
for rep in 1 2 3; do run stride18 65536; run stride20 65536; done
for n in 8192 16384 32768; do run stride18 $n; run stride20 $n; done
cat $R
unset TGP_LIB_PATH
timeout -k 10 600 python -m pytest tests/test_gpu_core.py tests/test_gpu_dist.py -m gpu -x -q > gpurun_out/r04_gputest_22.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04_gputest_22.log
