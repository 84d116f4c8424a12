#!/bin/bash
# round-4 GPU call 8: the sweeps with the next step's diagonal product riding in the bulk launch: whole GPU suite, then A/B
mkdir -p gpurun_out
L=gpurun_out/r04_gputest_8.log
timeout -k 10 300 python -m pytest tests/test_gpu_core.py tests/test_gpu_edge_cases.py -m gpu -x -q > $L 2>&1
rc=$?; echo "pytest(core) rc=$rc" >> $L; grep "passed\|failed\|rc=" $L
[ $rc -eq 0 ] || exit 1
A=gpurun_out/r04_trsv_fused_ab.txt; : > $A
for rep in 1 2; do
  echo "# TGP_POTRS_UNFUSED=1 (two launches per step)" >> $A; TGP_POTRS_UNFUSED=1 timeout -k 10 300 python tools/trsv_bench.py 1000 2300 4096 8192 16384 32768 65536 2>/dev/null | grep '^{' >> $A || exit 1
  echo "# default (next diagonal product rides in the bulk launch)" >> $A; timeout -k 10 300 python tools/trsv_bench.py 1000 2300 4096 8192 16384 32768 65536 2>/dev/null | grep '^{' >> $A || exit 1
done
python - <<'PY'
import json
for ln in open("gpurun_out/r04_trsv_fused_ab.txt"):
    if ln.startswith("#"): print(ln.strip())
    else:
        d=json.loads(ln); print("  N=%6d step %4d  big %.3f ms  frac %.3f  diff %.1e" % (d["n"], d["step"], d["big_ms"], d["big_frac_hbm"], d["max_rel_diff"]))
PY
timeout -k 10 1000 python -m pytest tests -m gpu -x -q >> $L 2>&1
rc=$?; echo "pytest(all) rc=$rc" >> $L; grep "passed\|failed\|rc=" $L
[ $rc -eq 0 ] || exit 1
