#!/bin/bash
# effective shader clock per kernel variant of devtools/micro/s16_pc (group $1): GRBM_GUI_ACTIVE / kernel duration, one PMC pass (counters only)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R
rm -rf /tmp/clk
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/clk -o clk -- $R/devtools/micro/s16_pc ${1:-1} > /tmp/clk.log 2>&1
python3 - <<'PY'
import sqlite3, glob, collections
db = glob.glob('/tmp/clk/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
dur = collections.defaultdict(list)
for name, d, did in c.execute("select name, end - start, dispatch_id from kernels"):
    dur[name].append(d)
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for kn, cn, v, did in c.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
    cnt[kn][cn] += v
    n[kn].add(did)
print("%-100s %6s %9s %12s %9s %9s" % ("kernel", "calls", "avg_us", "gui_cycles", "GHz(/8)", "mfma_busy"))
for kn in sorted(cnt, key=lambda k: -sum(dur[k])):
    if 'gemm_s16' not in kn: continue
    k = len(n[kn]); us = sum(dur[kn]) / len(dur[kn]) / 1e3
    gui = cnt[kn]['GRBM_GUI_ACTIVE'] / k
    short = kn[kn.find('gemm_s16'):][:96]
    print("%-100s %6d %9.2f %12.0f %9.3f %9.0f" % (short, k, us, gui, gui / 8 / us / 1e3, cnt[kn]['SQ_VALU_MFMA_BUSY_CYCLES'] / k))
PY
