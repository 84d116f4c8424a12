#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "world_of_one or fused" > gpurun_out/r04_gputest_9.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04_gputest_9.log
W=gpurun_out/r04_world_of_one.txt
echo "# single-GPU path (tools/quick_perf.py 65536)" > $W; timeout -k 10 200 python tools/quick_perf.py 65536 2>&1 | grep it1 >> $W
echo "# the multi-GPU driver with a world of one (tools/world_of_one.py 65536 2)" >> $W; timeout -k 10 200 python tools/world_of_one.py 65536 2 2>&1 | grep -v amdgpu >> $W
echo "# the same with the fused bulk launch forced (TGP_DIST_FUSED=1)" >> $W; TGP_DIST_FUSED=1 timeout -k 10 200 python tools/world_of_one.py 65536 2 2>&1 | grep -v amdgpu >> $W
cat $W
TGP_BENCH_PROFILE_API=1 timeout -k 10 400 python bench.py --cpu-sample 0 --steps 2 > gpurun_out/r04_bench_apiprof.json 2> gpurun_out/r04_bench_apiprof.err; grep -v amdgpu gpurun_out/r04_bench_apiprof.err | head -60
