#!/usr/bin/env python3
"""List the dispatches of the LAST of `reps` identical runs in a rocprofv3 results .db (kernel trace): start (us), duration,
stream, workgroups, kernel.  usage: trace_list.py results.db [reps=3] [first] [count]"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    rows = c.execute("select name,start,end,stream_id,grid_x,workgroup_x from kernels order by start").fetchall()
    n = len(rows) // reps
    last = rows[(reps - 1) * n:]
    t0 = last[0][1]

    def short(nm):
        m = re.search(r"(\w+)(<[^(]*>)?\(", nm)
        return (m.group(1) + (m.group(2) or ""))[:44] if m else nm[:44]
    print("run of %d dispatches, span %.3f ms" % (len(last), (max(r[2] for r in last) - t0) / 1e6))
    for name, s, e, st, gx, wx in last[first:first + count]:
        print("%9.1f  %7.1f us  s%d  wg %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, gx // wx, short(name)))


if __name__ == "__main__":
    main()
