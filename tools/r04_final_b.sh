#!/bin/bash
# round-4 final collection, call B: bench line with the PMC traffic of call A, rank slices, world of one vs single GPU, soak
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r04_bench_n65536_with_traffic.json 2> gpurun_out/r04_bench_b.err || exit 1
echo "bench done"
bash tools/r04_rank_slice.sh || exit 1
W=gpurun_out/r04_world_of_one.txt
echo "# single-GPU path (tools/quick_perf.py 65536)" > $W; timeout -k 10 200 python tools/quick_perf.py 65536 2>&1 | grep it1 >> $W
echo "# the multi-GPU driver with a world of one (tools/world_of_one.py 65536 2)" >> $W; timeout -k 10 200 python tools/world_of_one.py 65536 2 2>&1 | grep -v amdgpu >> $W
cat $W
timeout -k 10 200 python tools/api_overhead.py 65536 2>&1 | grep "^rep" > gpurun_out/r04_api_overhead.txt
timeout -k 10 200 python tools/api_overhead.py 65536 profiling 2>&1 | grep "^rep" | sed 's/^/profiling on: /' >> gpurun_out/r04_api_overhead.txt
cat gpurun_out/r04_api_overhead.txt
timeout -k 10 600 python tools/soak_handoffs.py 8 > gpurun_out/r04_soak_handoffs.txt 2>&1; echo "soak rc=$?"; tail -12 gpurun_out/r04_soak_handoffs.txt
