"""GPU operations around one randomizer refill of a tool run (rocprofv3 db with --kernel-trace --memory-copy-trace): prof_tool_refill.py db [which]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
ev = [(s, e, n, q) for s, e, n, q in c.execute("select start, end, name, %s from kernels" % (qcol or "0"))]
try:
    mc = [r[1] for r in c.execute("pragma table_info(memory_copies)")]
    q2 = "queue_id" if "queue_id" in mc else ("stream_id" if "stream_id" in mc else "0")
    ev += [(s, e, "COPY " + str(n), q) for s, e, n, q in c.execute("select start, end, name, %s from memory_copies" % q2)]
except Exception as ex:
    print("no copies:", ex)
ev.sort()
g = [i for i, x in enumerate(ev) if "row_gather" in x[2]]
i0 = g[min(which, len(g) - 1)]
t0 = ev[i0][0]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("aslp::", "").replace("void ", "").split("(")[0][:44]
prev = None
for s, e, n, q in ev:
    if s < t0 - 300e3 or s > t0 + 3500e3: continue
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%9.1f us  dur %7.1f  gap %7.1f  q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, short(n)))
    prev = max(prev or 0, e)
