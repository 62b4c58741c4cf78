"""Round 6, VERDICT item 1(a): where do the waves of a split-fp16 product wait?  rocprofv3 --pmc passes (counters only, no other trace
domain) on ONE product shape per pass group, with the LDS / texture-path / wait counters this stack has (the candidates are filtered
against `rocprofv3 -L` first, so an unknown name costs nothing).  Writes per-shape summaries under gpurun_out/r6/.

    python3 devtools/r6_pmc_gemm.py [out_dir] [shape ...]      shape = tA,tB,M,N,K

This driver never touches the GPU itself; every pass is a child `rocprofv3 ... -- python3 devtools/one_gemm.py`."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6")
SHAPES = sys.argv[2:] or ["0,1,1024,2048,2048", "0,1,4096,4096,4096", "1,0,2048,2048,1024"]
os.makedirs(OUT, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")

lst = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, env=env)
listing = lst.stdout + lst.stderr
open(os.path.join(OUT, "counters_list.txt"), "w").write(listing)
known = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", listing))

# candidate passes (<= 8 SQ, <= 4 TCC-class counters each; GRBM rides along)
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_MEM_VIOLATIONS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_FLAT", "SQ_ACTIVE_INST_FLAT", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU", "SQ_WAIT_INST_VMEM"],
    ["SQ_INSTS_LDS_DMA", "SQ_INSTS_VMEM_LDS", "SQ_IFETCH", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_MISC", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_BRANCH", "SQ_WAVE_DEP_WAIT"],
    # (a pass with the TA_* counters never returned on this stack -- ten minutes of the first run -- and is left out)
    ["TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TA_TCP_STATE_READ_sum"],
    ["TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_TOTAL_READ_sum", "TCP_UTCL1_REQUEST_sum", "TD_TD_BUSY_sum", "TD_TC_STALL_sum", "TD_LOAD_WAVEFRONT_sum", "TCP_TOTAL_ACCESSES_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_TAG_STALL_sum", "TCC_BUSY_sum"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]


def summarise(db):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "devtools", "prof_summary.py"), db], capture_output=True, text=True)
    keep, on = [], False
    for line in r.stdout.splitlines():
        if re.match(r"^(gemm_s16|gemm_f32)", line) and "dispatches" in line:
            on = True
            keep.append(line[:150])
            continue
        if on:
            if line.startswith("    "):
                keep.append(line)
            else:
                on = False
    return "\n".join(keep)


for shape in SHAPES:
    tag = shape.replace(",", "_")
    out_txt = os.path.join(OUT, "pmc_gemm_%s.txt" % tag)
    with open(out_txt, "w") as f:
        f.write("# one_gemm.py SHAPE=%s REPS=10: rocprofv3 --kernel-trace --pmc <set>; counter values are SUMS over the 10 dispatches of the kernel\n" % shape)
        for cand in PASSES:
            names = [c for c in cand if c in known]
            missing = [c for c in cand if c not in known]
            if not names:
                f.write("# pass skipped, no such counters here: %s\n" % " ".join(missing))
                continue
            d = "/tmp/og_%s" % tag
            subprocess.run(["rm", "-rf", d])
            e = dict(env, SHAPE=shape, TILE=os.environ.get("TILE", "0"), REPS="10")
            cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + names + ["-d", d, "-o", "og", "--", "python3", os.path.join(ROOT, "devtools", "one_gemm.py")]
            try:
                p = subprocess.run(cmd, capture_output=True, text=True, env=e, cwd=ROOT, timeout=150)
            except subprocess.TimeoutExpired:
                f.write("## pass: %s\n# TIMED OUT after 150 s\n" % " ".join(names))
                continue
            dbs = subprocess.run(["find", d, "-name", "*.db"], capture_output=True, text=True).stdout.split()
            f.write("## pass: %s%s\n" % (" ".join(names), ("   (not on this stack: %s)" % " ".join(missing)) if missing else ""))
            if p.returncode != 0 or not dbs:
                f.write("# FAILED rc=%d: %s\n" % (p.returncode, (p.stderr or p.stdout)[-400:].replace("\n", " | ")))
                continue
            f.write(summarise(dbs[0]) + "\n")
            f.flush()
    print(open(out_txt).read())
