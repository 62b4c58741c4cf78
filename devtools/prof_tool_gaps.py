"""GPU idle time of a whole tool run from a rocprofv3 rocpd database (--kernel-trace [--memory-copy-trace]): span from the
first to the last kernel, time covered by kernels, and the largest idle gaps with the kernels either side.
Usage: prof_tool_gaps.py results.db [min-gap-us]"""
import sqlite3
import sys

db = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
c = sqlite3.connect(db)
rows = sorted(c.execute("select start, end, name from kernels"))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("aslp::", "").replace("void ", "")[:70]
span = (rows[-1][1] - rows[0][0]) / 1e3
busy, prev_end, gaps = 0.0, rows[0][0], []
for i, (s, e, n) in enumerate(rows):
    if s > prev_end:
        gaps.append(((s - prev_end) / 1e3, (prev_end - rows[0][0]) / 1e3, short(rows[i - 1][2]), short(n)))
        busy += (e - s) / 1e3
    else:
        busy += max(0, e - prev_end) / 1e3
    prev_end = max(prev_end, e)
print("kernels %d, span %.1f ms, covered by kernels %.1f ms (%.1f %%), idle %.1f ms" % (len(rows), span / 1e3, busy / 1e3, 100 * busy / span, (span - busy) / 1e3))
big = [g for g in gaps if g[0] >= min_gap]
print("gaps >= %.0f us: %d, total %.1f ms; gaps below: %d, total %.1f ms" % (min_gap, len(big), sum(g[0] for g in big) / 1e3, len(gaps) - len(big),
                                                                           sum(g[0] for g in gaps if g[0] < min_gap) / 1e3))
for g in sorted(big, reverse=True)[:25]:
    print("  %8.1f us at %9.1f us   after %-50s before %s" % g)
try:
    cp = list(c.execute("select start, end, name from memory_copies"))
    tot = sum(e - s for s, e, _ in cp) / 1e6
    print("memory copies: %d, %.1f ms in total" % (len(cp), tot))
except Exception as ex:  # table absent without --memory-copy-trace
    print("no memory-copy table (%s)" % ex)
