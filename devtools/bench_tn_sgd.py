"""The cfg2 weight-gradient product with its fused SGD epilogue (momentum on the gradient buffer, W += -lr G), alone on the
chip, per tile configuration: TILES=212,213,207 python devtools/bench_tn_sgd.py"""
import os, sys, torch
sys.path.insert(0, '.')
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
_w = torch.randn(4096, 4096, device=dev)
for _ in range(800): torch.mm(_w, _w)  # ~1 s: the clocks take several hundred ms to settle after idle; shorter warm-ups bias whatever is timed first
torch.cuda.synchronize()
M, N, K = 2048, 2048, 1024
A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev)
G = torch.zeros(M, N, device=dev); W = torch.randn(M, N, device=dev)
ep = aslp._lib.GemmEpilogue(None, 0.0, W.data_ptr(), N, -1e-5, None, 0, 0)
for cfg in [int(c) for c in os.environ.get('TILES', '212,213,207').split(',')]:
    for name, beta, e in (("plain", 0.0, None), ("momentum + SGD step", 0.9, ep)):
        aslp.lib.aslp_gemm_force_tile(cfg)
        f = lambda: aslp.ops.sgemm(1, 0, 1.0, A, B, beta, G, e)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print("cfg %d -> ran %d  %-20s %6.1f us  %6.1f TFLOP/s" % (cfg, aslp.lib.aslp_gemm_last_tile(), name, us, 2.0 * M * N * K / us / 1e6))
