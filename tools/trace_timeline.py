#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name count / total / mean, the busy time of every queue and
the span of the trace.  usage: trace_timeline.py <kernel_trace.csv> [skip_first_n_dispatches]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rows = rows[skip:]
    t0 = int(rows[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    print("dispatches %d, span %.3f ms" % (len(rows), (t1 - t0) / 1e6))
    per = defaultdict(lambda: [0, 0])
    queues = defaultdict(list)
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        name = r["Kernel_Name"].split("(")[0][-60:]
        per[name][0] += 1
        per[name][1] += d
        queues[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    for name, (c, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-62s n=%5d total %9.3f ms  mean %8.2f us" % (name, c, tot / 1e6, tot / c / 1e3))
    for q, iv in queues.items():
        iv.sort()
        busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
        for s, e in iv[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        print("  queue %s: %d dispatches, busy %.3f ms, first %.3f last %.3f ms" %
              (q, len(iv), busy / 1e6, (iv[0][0] - t0) / 1e6, (max(e for _, e in iv) - t0) / 1e6))
    # union of all queues
    iv = sorted(x for v in queues.values() for x in v)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    gaps = []
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("  any queue busy %.3f ms; idle gaps: %d totalling %.3f ms (max %.1f us)" %
          (busy / 1e6, len(gaps), sum(gaps) / 1e6, max(gaps) / 1e3 if gaps else 0.0))


if __name__ == "__main__":
    main()
