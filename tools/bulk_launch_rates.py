#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a single-GPU factorisation: every whole-chip bulk-update launch of the LAST
factorisation with its workgroups, duration and time per 512 workgroups -- is the first launch slower than the rest?
usage: bulk_launch_rates.py <kernel_trace.csv> [name fragment, default syrk_segs_kernel]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frag = sys.argv[2] if len(sys.argv) > 2 else "syrk_segs_kernel"
starts = [i for i, r in enumerate(rows) if "kbuild" in r["Kernel_Name"]]
rows = rows[starts[-1]:] if starts else rows
t0 = int(rows[0]["Start_Timestamp"])
print("  #  start ms   dur ms     wgs   us per 512 wgs   idle before (us)")
prev = None
n = 0
for r in rows:
    if frag not in r["Kernel_Name"]:
        continue
    wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    if wgs < 2048:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%3d  %8.3f  %7.3f  %6d   %8.2f        %8.1f" % (n, (s - t0) / 1e6, (e - s) / 1e6, wgs, (e - s) / 1e3 / (wgs / 512.0),
                                                       (s - prev) / 1e3 if prev else 0.0))
    prev = e
    n += 1
