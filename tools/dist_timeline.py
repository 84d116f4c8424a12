#!/usr/bin/env python3
"""Per-launch timeline of the multi-GPU driver's bulk updates from a rocprofv3 --kernel-trace CSV of tools/rank_slice.py:
for every bulk launch of the LAST factorisation in the trace its start, duration, workgroups, how long it ran beside the
launch before it (overlapped launches), the time no bulk kernel was running before it, and what the panel chain did meanwhile.
usage: dist_timeline.py <kernel_trace.csv> [rows to print, default all]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"]                      # noqa: E731
    # factorisations are separated by the K build
    starts = [i for i, r in enumerate(rows) if "kbuild" in name(r)]
    rows = rows[starts[-1]:] if starts else rows
    t0 = int(rows[0]["Start_Timestamp"])
    bulk = [r for r in rows if "syrk_distn_kernel" in name(r) and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) > 600]
    other = [r for r in rows if r not in bulk and "kbuild" not in name(r)]
    print("last factorisation: %d kernels, %d bulk launches, span %.2f ms" % (
        len(rows), len(bulk), (max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e6))
    print("  #   start ms   dur ms   wgs   beside-prev ms   no-bulk gap ms   chain kernels busy in gap/launch ms")
    prev_end, tot_gap, tot_ov, tot_dur = None, 0.0, 0.0, 0.0
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
    for i, r in enumerate(bulk):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wgs = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        ov = max(0, (prev_end or s) - s) / 1e6 if prev_end else 0.0
        gap = max(0, s - prev_end) / 1e6 if prev_end else 0.0
        lo = prev_end if prev_end and prev_end < s else s
        busy = sum(min(int(o["End_Timestamp"]), e) - max(int(o["Start_Timestamp"]), lo) for o in other
                   if int(o["End_Timestamp"]) > lo and int(o["Start_Timestamp"]) < e) / 1e6
        tot_gap += gap
        tot_ov += ov
        tot_dur += (e - s) / 1e6
        if i < lim:
            print("%3d  %8.3f  %7.3f  %5d   %8.3f        %8.3f         %8.3f" % (i, (s - t0) / 1e6, (e - s) / 1e6, wgs, ov, gap, busy))
        prev_end = max(prev_end or 0, e)
    print("bulk kernels: sum of durations %.2f ms, ran beside the previous one %.2f ms, no bulk kernel running %.2f ms "
          "(between the first bulk start and the last bulk end)" % (tot_dur, tot_ov, tot_gap))
    last = max(int(r["End_Timestamp"]) for r in bulk)
    print("after the last bulk launch: %.2f ms" % ((max(int(r["End_Timestamp"]) for r in rows) - last) / 1e6))
    print("before the first bulk launch: %.2f ms" % ((int(bulk[0]["Start_Timestamp"]) - t0) / 1e6))
    if len(sys.argv) > 3:                                   # every kernel from this many ms before the last bulk launch's end
        lo = last - int(float(sys.argv[3]) * 1e6)
        print("kernels from %.2f ms before the end of the last bulk launch on (start ms rel. to it, dur us, workgroups, queue, name):" % float(sys.argv[3]))
        for r in rows:
            if int(r["End_Timestamp"]) >= lo:
                print("  %8.3f  %8.1f  %6d  q%-3s %s" % ((int(r["Start_Timestamp"]) - last) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                   int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r.get("Queue_Id", "?"), name(r).split("(")[0][-70:]))


if __name__ == "__main__":
    main()
