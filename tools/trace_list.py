#!/usr/bin/env python3
"""List the dispatches of the LAST solve in a `rocprofv3 --kernel-trace --output-format csv` run of tools/one_solve.py:
start (us, from the first dispatch of that solve), duration, stream/queue, workgroups, kernel -- and per-kernel totals.
usage: trace_list.py <dir with *kernel_trace.csv> [max lines]"""
import csv
import glob
import re
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for fn in files:
    for r in csv.DictReader(open(fn)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).split("(")[0]
        name = re.sub(r"potrf_v2::", "", name)
        wg = 1
        for ax in "XYZ":
            g, w = int(r.get("Grid_Size_" + ax, 1) or 1), int(r.get("Workgroup_Size_" + ax, 1) or 1)
            wg *= max(g // max(w, 1), 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id") or r.get("Queue_Id"), wg, name))
rows.sort()
# solves are separated by host round trips: split at gaps > 300 us and keep the last run with more than 50 dispatches
runs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 300000:
        runs.append(cur)
        cur = []
    cur.append(b)
runs.append(cur)
big = [r for r in runs if len(r) > 50]
run = big[-1] if big else runs[-1]
t0 = run[0][0]
print("# run of %d dispatches, span %.3f ms" % (len(run), (run[-1][1] - t0) / 1e6))
streams = {}
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
tot = defaultdict(lambda: [0, 0.0])
for i, (s, e, q, wg, name) in enumerate(run):
    sid = streams.setdefault(q, "s%d" % (len(streams) + 1))
    tot[name][0] += 1
    tot[name][1] += (e - s) / 1e3
    if i < limit:
        print("%9.1f %8.1f us  %s  wg %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, sid, wg, name))
print("# totals (us) of that run")
for name, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("# %8.1f us  %4d x  %s" % (us, c, name))
