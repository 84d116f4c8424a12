#!/bin/bash
# HBM-side traffic of the dominant kernels for bench.py's default workload, measured on the shipped sources:
#  1. tools/probes/fetch_calib.hip under FETCH_SIZE / WRITE_SIZE passes -> calibration factors for this box's counters
#     in the library's own access shapes (16-B and 8-B per lane, accumulator-shaped 8-B reads);
#  2. bench.py (one step, no warm-up, no CPU leg) under separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit
#     one pass), kernel-trace only;
#  3. tools/pmc_derive.py -> gpurun_out/pmc_bench_n65536.json (copy to profiles/rNN_pmc_bench_n65536.json).
# Run on the GPU box:  bash tools/pmc_bench.sh
set -e
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcb_* $R/gpurun_out/pmccal_*   # the derivation sums every CSV it finds
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/fetch_calib $R/tools/probes/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
# Under counter collection rocprofv3 serialises the dispatches of the intercepted queues, and a stream parked in
# hipStreamWaitValue32 (the factorisation's cross-stream hand-offs) then never gets going: these passes once hung for 7 minutes
# until the box's watchdog ended them.  The library tries the mechanism once per process (csrc/handoff.hip: consumer parked
# first, producer second, 2 s host timeout) and hands over by events where the trial fails; TGP_SYNC_EVENTS=1 is the documented
# profiler setting and skips the trial.  The kernels and their traffic are the same (the fused bulk launches run as two).
export TGP_SYNC_EVENTS=1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmccal_$C -- /tmp/fetch_calib > $R/gpurun_out/pmccal_$C.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$T -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-configs > $R/gpurun_out/pmcb_$T.json 2> $R/gpurun_out/pmcb_$T.err
done
python3 $R/tools/pmc_derive.py
