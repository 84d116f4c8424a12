#!/bin/bash
# API-route trace: kernels, copies and HIP API calls of one bench.py run (for the host-tax question of DESIGN 6)
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out/api_tr
rm -rf $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d $O -o t -f csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/r04_api_trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04_api_trace.err || exit 1
ls -la $O | head
