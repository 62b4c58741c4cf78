"""fp32 GEMM on the fp16 matrix instruction with two-piece operands (csrc/gemm_split16.hip) against the fp32-instruction kernels:
accuracy against a float64 product and time per launch.  Usage: python devtools/bench_split16.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import aslp_import  # noqa: E402

aslp = aslp_import.load()
aslp.ops.use_torch_stream()
dev = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator(device="cpu")
g.manual_seed(1)


def run(tA, tB, M, N, K, scale_a=1.0, scale_b=1.0):
    A = (torch.randn((K, M) if tA else (M, K), generator=g) * scale_a).to(dev)
    B = (torch.randn((N, K) if tB else (K, N), generator=g) * scale_b).to(dev)
    # a few rows far below the rest: per-matrix scaling must keep them
    if not tA:
        A[3] *= 1e-6
    ref = (A.double().T if tA else A.double()) @ (B.double().T if tB else B.double())
    mag = (A.double().abs().T if tA else A.double().abs()) @ (B.double().abs().T if tB else B.double().abs())
    out = {}
    for name, on in (("fp32", 0), ("split", 1)):
        aslp.lib.aslp_gemm_split16(on)
        C = torch.zeros(M, N, device=dev)
        aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C)
        torch.cuda.synchronize()
        err = ((C.double() - ref).abs() / mag).max().item()
        t0 = time.perf_counter()
        for _ in range(REPS):
            aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / REPS * 1e6
        out[name] = (err, us)
    # the product alone, from planes made once (what a training step pays per product)
    aslp.lib.aslp_gemm_split16(1)
    pa, pb = aslp.ops.Planes(A), aslp.ops.Planes(B)
    C = torch.zeros(M, N, device=dev)
    aslp.ops.sgemm_planes(tA, tB, 1.0, A, pa, B, pb, 0.0, C)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        aslp.ops.sgemm_planes(tA, tB, 1.0, A, pa, B, pb, 0.0, C)
    torch.cuda.synchronize()
    us_planes = (time.perf_counter() - t0) / REPS * 1e6
    aslp.lib.aslp_gemm_split16(-1)
    f = 2.0 * M * N * K
    print("   product alone from prepared planes: %7.1f us  %6.1f TF-equivalent (%.2f of 839)" % (us_planes, f / us_planes / 1e6, f / us_planes / 1e6 / 839))
    print("%s%s %5d x %5d x %5d  scale %g/%g:  fp32 err %.2e %7.1f us %6.1f TF | split err %.2e %7.1f us %6.1f TF-equivalent"
          % ("T" if tA else "N", "T" if tB else "N", M, N, K, scale_a, scale_b, out["fp32"][0], out["fp32"][1], f / out["fp32"][1] / 1e6,
             out["split"][0], out["split"][1], f / out["split"][1] / 1e6))


for shape in ((0, 1, 1024, 2048, 2048), (0, 0, 1024, 2048, 2048), (1, 0, 2048, 2048, 1024), (0, 1, 1024, 2048, 440), (0, 1, 1024, 3000, 2048),
              (1, 0, 3000, 2048, 1024), (0, 1, 4096, 4096, 4096), (0, 1, 1000, 1000, 1000)):
    run(*shape)
run(0, 1, 1024, 2048, 2048, 1e-7, 3e3)
run(1, 0, 2048, 2048, 1024, 1e-12, 1.0)
