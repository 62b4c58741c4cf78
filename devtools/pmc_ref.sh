#!/bin/bash
# Kernel names + PMC counters of our GEMM (TILE) next to the rocBLAS kernel torch.mm picks, same shape.
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R
for SH in ${SHAPES:-0,1,1024,2048,2048 0,0,1024,2048,2048 1,0,2048,2048,1024}; do
  rm -rf /tmp/ref
  REF=1 SHAPE=$SH TILE=${TILE:-12} REPS=5 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d /tmp/ref -o r -- python3 $R/devtools/one_gemm.py > /tmp/ref.log 2>&1 || tail -5 /tmp/ref.log
  echo "=== shape $SH"
  python3 - <<PY
import sqlite3
c=sqlite3.connect("/tmp/ref/r_results.db")
names=set(r[0] for r in c.execute("select name from kernels"))
for n in names:
    if "gemm_f32" in n or "Cijk" in n:
        rows=list(c.execute("select end-start from kernels where name=?",(n,)))
        print(n[:300]); print("   calls",len(rows),"avg_us %.1f"%(sum(r[0] for r in rows)/len(rows)/1e3))
        for cn,v in c.execute("select counter_name,sum(value) from counters_collection where kernel_name=? group by counter_name",(n,)):
            print("      %-28s %14.0f per-dispatch"%(cn,v/len(rows)))
PY
done
