"""First and last milliseconds of a tool run's kernel + copy trace, runs of one kernel name merged: prof_tool_head.py results.db [head_ms] [tail_ms]"""
import sqlite3, sys
db = sys.argv[1]; head = float(sys.argv[2]) if len(sys.argv) > 2 else 130.0; tail = float(sys.argv[3]) if len(sys.argv) > 3 else 80.0
c = sqlite3.connect(db)
ev = [(s, e, n) for s, e, n in c.execute("select start, end, name from kernels")]
try:
    ev += [(s, e, "COPY " + str(n)) for s, e, n in c.execute("select start, end, name from memory_copies")]
except Exception:
    pass
ev.sort()
t0, t1 = ev[0][0], ev[-1][1]
short = lambda n: n.replace("(anonymous namespace)::", "").replace("aslp::", "").replace("void ", "").split("(")[0][:48]
def dump(lo, hi):
    run = None
    for s, e, n in ev:
        if s < lo or s > hi: continue
        n = short(n)
        if run and run[0] == n and s - run[2] < 200e3:
            run[2] = e; run[3] += 1; run[4] += e - s
        else:
            if run: print("%10.2f ms  +%8.2f ms  x%-5d busy %8.2f ms  %s" % ((run[1] - t0) / 1e6, (run[2] - run[1]) / 1e6, run[3], run[4] / 1e6, run[0]))
            run = [n, s, e, 1, e - s]
    if run: print("%10.2f ms  +%8.2f ms  x%-5d busy %8.2f ms  %s" % ((run[1] - t0) / 1e6, (run[2] - run[1]) / 1e6, run[3], run[4] / 1e6, run[0]))
print("---- head"); dump(t0, t0 + head * 1e6)
print("---- tail"); dump(t1 - tail * 1e6, t1)
