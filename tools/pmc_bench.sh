#!/bin/bash
# HBM traffic of the dominant kernel for bench.py's default workload: separate --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only.  Run on the GPU box.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$T -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/pmcb_$T.json 2> $R/gpurun_out/pmcb_$T.err
done
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for fn in glob.glob(R + "/gpurun_out/pmcb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        nm = r["Kernel_Name"]
        k = "syrk_segs_kernel<4>" if ("syrk_segs_kernel<4>" in nm) else ("kbuild_lower_kernel" if "kbuild_lower" in nm else None)
        if k is None: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {}
for k in agg:
    out[k] = {c: {"sum": v, "dispatches": len(disp[(k, c)])} for c, v in agg[k].items()}
print(json.dumps(out, indent=1))
json.dump(out, open(R + "/gpurun_out/pmc_bench_n65536.json", "w"), indent=1)
PY
