#!/bin/bash
# kernel trace of the last of three solves at size N (tools/one_solve.py) as a listing: bash tools/trace_call.sh <N> <out.txt> [lines] [VAR=value ...]
N=$1; out=$2; lines=${3:-400}; shift 3 || true
R=$PWD
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/_trace_tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/_trace_tmp -- python3 $R/tools/one_solve.py $N > /dev/null 2>&1 || exit 1
python3 $R/tools/trace_list.py $R/gpurun_out/_trace_tmp $lines > $R/$out
rm -rf $R/gpurun_out/_trace_tmp
tail -16 $R/$out
