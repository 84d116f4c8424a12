#!/bin/bash
# usage: tools/pmc_run.sh <tag> <N> <counter list...>   (one rocprofv3 --pmc pass; run on the GPU box)
set -e
TAG=$1; N=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/quick_perf.py $N > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for fn in f:
    for r in csv.DictReader(open(fn)):
        import re
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"^void ", "", k).split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key); cnt[k] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:8]:
    print(k, "dispatches", cnt[k], {c: "%.4g" % v for c, v in agg[k].items()})
PY
