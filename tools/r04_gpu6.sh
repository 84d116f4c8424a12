#!/bin/bash
# round-4 GPU call 6: whole GPU suite on the accumulated changes; tilemap A/B against the previous build; N = 8192 traces
mkdir -p gpurun_out
R=$PWD
L=gpurun_out/r04_gputest_6.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $L 2>&1
rc=$?; echo "pytest rc=$rc" >> $L; grep "passed\|failed\|rc=" $L
[ $rc -eq 0 ] || exit 1
A=gpurun_out/r04_tilemap_ab.txt; : > $A
for rep in 1 2; do
  echo "# previous tile map (whole super-tiles dealt in turn)" >> $A; TGP_LIB_PATH=$R/treegp_amd/csrc/libtgp_pretilemap.so timeout -k 10 300 python tools/ab_quick_perf.py 4096 8192 16384 32768 65536 2>&1 | grep "it1" | cut -c1-170 >> $A || exit 1
  echo "# remainder super-tiles sliced over the XCDs" >> $A; timeout -k 10 300 python tools/quick_perf.py 4096 8192 16384 32768 65536 2>&1 | grep "it1" | cut -c1-170 >> $A || exit 1
done
cat $A
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/r04_trace8192_$v
  TGP_PANEL_OVERLAP=$v timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_trace8192_$v -- python3 $R/tools/one_solve.py 8192 > /dev/null 2>&1 || exit 1
  python3 $R/tools/trace_list.py $R/gpurun_out/r04_trace8192_$v 260 > $R/gpurun_out/r04_trace_n8192_overlap$v.txt
  rm -rf $R/gpurun_out/r04_trace8192_$v
done
tail -25 $R/gpurun_out/r04_trace_n8192_overlap0.txt; tail -25 $R/gpurun_out/r04_trace_n8192_overlap1.txt
