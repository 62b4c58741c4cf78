"""Correctness of forced GEMM tile configs against a float64 torch product (max rel err), all four transpose
variants, incl. ragged M/N.  Usage: TILES=212,213 python devtools/check_gemm.py"""
import os, sys, torch
sys.path.insert(0, '.')
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
cfgs = [int(c) for c in os.environ.get('TILES', '212').split(',')]
shapes = [(1024, 2048, 2048), (64, 128, 64), (100, 260, 96), (1024, 3000, 2048), (2048, 440, 1024), (192, 128, 32), (60 * 32, 2048, 512),
          (1024, 2048, 440), (1024, 2048, 3000), (128, 256, 24), (64, 64, 4), (200, 132, 100), (96, 64, 36)]
ok = True
for cfg in cfgs:
    aslp.lib.aslp_gemm_force_tile(cfg)
    for (M, N, K) in shapes:
        for tA in (0, 1):
            for tB in (0, 1):
                torch.manual_seed(M + N + K + tA * 2 + tB)
                A = torch.randn((K, M) if tA else (M, K), device=dev)
                B = torch.randn((N, K) if tB else (K, N), device=dev)
                C = torch.randn(M, N, device=dev)
                ref = 0.5 * ((A.t() if tA else A).double() @ (B.t() if tB else B).double()) + 0.25 * C.double()
                for rep in range(3):   # repeat: races show up as run-to-run differences
                    Cc = C.clone()
                    aslp.ops.sgemm(tA, tB, 0.5, A, B, 0.25, Cc)
                    err = ((Cc.double() - ref).abs().max() / ref.abs().max()).item()
                    if err > 1e-5:
                        ok = False
                        print("FAIL cfg", cfg, (M, N, K), "tA", tA, "tB", tB, "rep", rep, "err %.3e" % err)
                        break
aslp.lib.aslp_gemm_force_tile(0)
print("ALL OK" if ok else "FAILURES")
