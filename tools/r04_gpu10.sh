#!/bin/bash
mkdir -p gpurun_out
TGP_HOST_PHASES=1 timeout -k 10 400 python bench.py --cpu-sample 0 --steps 2 > gpurun_out/r04_bench_hostphases.json 2> gpurun_out/r04_bench_hostphases.err; grep "tgp_gp_solve n=65536\|factor_and_solve" gpurun_out/r04_bench_hostphases.err | tail -12
TGP_HOST_PHASES=1 timeout -k 10 200 python tools/api_overhead.py 65536 2>&1 | grep "tgp_gp_solve n=\|^rep" | head -8
