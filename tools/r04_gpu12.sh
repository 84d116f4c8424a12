#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_api_dist.py -m gpu -x -q -k "threshold_route" > gpurun_out/r04_gputest_12.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04_gputest_12.log
R=gpurun_out/r04_head_fence_ab.txt; : > $R
for rep in 1 2 3; do
  echo "# every wave fences (shipped)" >> $R; timeout -k 10 150 python tools/rank_slice.py 65536 8 7 2>&1 | grep "N=" | cut -c1-200 >> $R
  echo "# one wave fences after the barrier" >> $R; TGP_LIB_PATH=$PWD/treegp_amd/csrc/libtgp_onewave.so timeout -k 10 150 python tools/rank_slice.py 65536 8 7 2>&1 | grep "N=" | cut -c1-200 >> $R
done
cat $R
