"""Prints the kernel timeline of one steady-state step from a rocprofv3 rocpd database: start offset, duration and
the idle gap since the previous kernel ended.  Usage: prof_timeline.py results.db [anchor-kernel-substring] [step-index]"""
import sqlite3
import sys

db = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "CopyMat"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 10
c = sqlite3.connect(db)
rows = sorted(c.execute("select start, end, name from kernels"))
starts = [i for i, r in enumerate(rows) if anchor in r[2]]
a, b = starts[which], starts[which + 1]
t0 = rows[a][0]
prev_end = rows[a][0]
busy = gaps = 0.0
for s, e, n in rows[a:b]:
    n = n.replace("(anonymous namespace)::", "").replace("aslp::", "").replace("void ", "")[:60]
    print("%9.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
    busy += (e - s) / 1e3
    gaps += max(0.0, (s - prev_end) / 1e3)
    prev_end = max(prev_end, e)
print("step span %.1f us, kernel time %.1f us, idle gaps %.1f us" % ((rows[b][0] - t0) / 1e3, busy, gaps))
