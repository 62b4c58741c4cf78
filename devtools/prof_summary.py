"""Summarises a rocprofv3 rocpd sqlite database: per-kernel calls / total / average duration
(the same columns as `--stats`' kernel_stats.csv) and, if present, PMC counters per kernel."""
import collections
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("aslp::", "")
    if n.startswith("void "):
        n = n[5:]
    depth, out = 0, []
    for ch in n:           # cut the argument list: first "(" at template depth 0
        if ch == "<": depth += 1
        if ch == ">": depth -= 1
        if ch == "(" and depth == 0: break
        out.append(ch)
    return "".join(out)[:110]
rows = list(c.execute("select name, end - start from kernels"))
agg = collections.defaultdict(lambda: [0, 0.0])
for name, dur in rows:
    a = agg[short(name)]
    a[0] += 1
    a[1] += dur
tot = sum(a[1] for a in agg.values())
print("%-112s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-112s %8d %12.1f %10.2f %6.2f" % (k, n, t / 1e3, t / 1e3 / n, 100.0 * t / tot))
print("TOTAL kernel time us: %.1f" % (tot / 1e3))
try:
    pm = list(c.execute("select kernel_name, counter_name, value from counters_collection"))
except sqlite3.Error:
    pm = []
if pm:
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for kn, cn, v in pm:
        d[short(kn)][cn] += v
    for kn, in set((short(k),) for k, _, _ in pm):
        cnt[kn] = len(set(r[0] for r in c.execute("select dispatch_id from counters_collection where kernel_name like ?", (kn[:40] + "%",))))
    print("\nPMC counters (sum over dispatches; per-dispatch average uses the dispatch count of the kernel trace):")
    for kn, cs in sorted(d.items(), key=lambda kv: -agg[kv[0]][1] if kv[0] in agg else 0):
        n = agg[kn][0] if kn in agg else 0
        print("%s   [%d dispatches, avg %.2f us]" % (kn, n, agg[kn][1] / 1e3 / n if n else 0.0))
        for cn, v in sorted(cs.items()):
            print("    %-32s %18.0f   per-dispatch %14.1f" % (cn, v, v / n if n else 0.0))
