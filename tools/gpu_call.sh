#!/bin/bash
# One GPU call: runs the command lines of a job file under a per-step timeout, joined with && (a failed or killed step ends the
# call: no further GPU step is started), everything logged to gpurun_out/<name>.log.
# usage (through gpurun): bash tools/gpu_call.sh <name> <jobfile>      jobfile: one shell command per line, '#' comments
set -o pipefail
name=$1; job=$2
mkdir -p gpurun_out
log=gpurun_out/$name.log
: > $log
while IFS= read -r line; do
    case "$line" in ''|'#'*) continue;; esac
    echo "### $line" >> $log
    if ! timeout -k 10 ${STEP_TIMEOUT:-300} bash -c "$line" >> $log 2>&1; then
        echo "### FAILED (rc $?): $line" >> $log
        tail -30 $log
        exit 1
    fi
done < "$job"
tail -${TAIL:-60} $log
