#!/bin/bash
# round-4 final collection, call A: the whole GPU suite on the final sources, then tools/round_profiles.sh (bench line,
# per-kernel tools, rocprofv3 stats, PMC passes)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_final.log 2>&1
rc=$?; echo "pytest rc=$rc" >> gpurun_out/r04_gputest_final.log; grep "passed\|failed\|rc=" gpurun_out/r04_gputest_final.log
[ $rc -eq 0 ] || exit 1
bash tools/round_profiles.sh r04 2>&1 | tail -8
