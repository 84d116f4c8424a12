#!/bin/bash
# round-4 final collection, call A: tools/round_profiles.sh (bench line, per-kernel tools, rocprofv3 stats, PMC passes)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash tools/round_profiles.sh r04 2>&1 | tail -20
ls -la gpurun_out/ | grep r04_ | tail -20
