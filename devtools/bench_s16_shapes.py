"""time split-fp16 products of given shapes from prepared planes: python devtools/bench_s16_shapes.py  (env switches select the tile)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aslp_import
aslp = aslp_import.load()
aslp.ops.use_torch_stream()
dev = torch.device("cuda:0")
SHAPES = [(0, 1, 1920, 2048, 1024), (0, 1, 1920, 2048, 512), (0, 1, 1920, 512, 512), (0, 0, 1920, 1024, 2048), (0, 0, 1920, 512, 2048),
          (1, 0, 2048, 1024, 1920), (1, 0, 2048, 512, 1920), (1, 0, 512, 512, 1920), (0, 1, 1920, 3000, 1024), (0, 1, 4096, 4096, 4096), (0, 1, 2048, 2048, 2048),
          (0, 1, 8192, 2048, 2048)]
for tA, tB, M, N, K in SHAPES:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((N, K) if tB else (K, N), device=dev)
    pa, pb = aslp.ops.Planes(A), aslp.ops.Planes(B)
    Cm = torch.zeros(M, N, device=dev)
    for _ in range(3):
        aslp.ops.sgemm_planes(tA, tB, 1.0, A, pa, B, pb, 0.0, Cm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        aslp.ops.sgemm_planes(tA, tB, 1.0, A, pa, B, pb, 0.0, Cm)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    print("%s%s %5d x %5d x %5d  %7.1f us  %6.1f TF-eq  tile %d" % ("T" if tA else "N", "T" if tB else "N", M, N, K, us, 2.0 * M * N * K / us / 1e6, aslp.lib.aslp_gemm_last_tile() if hasattr(aslp.lib, "aslp_gemm_last_tile") else -1))
