"""One product shape under several split-fp16 tile configurations, interleaved rounds, HIP events around 200 launches each (clock settled by
the 300 launches in front).     python3 devtools/r6_shape_tiles.py tA,tB,M,N,K cfg [cfg ...]      (cfg 0 = the launcher's own choice)"""
import sys
import torch
sys.path.insert(0, '.')
import aslp_import
aslp = aslp_import.load()
aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
tA, tB, M, N, K = [int(x) for x in sys.argv[1].split(',')]
cfgs = [int(x) for x in sys.argv[2:]] or [0]
A = torch.randn((K, M) if tA else (M, K), device=dev)
B = torch.randn((N, K) if tB else (K, N), device=dev)
C = torch.empty(M, N, device=dev)
ref = None
for rnd in range(3):
    for cfg in cfgs:
        aslp.lib.aslp_gemm_split16_tile(cfg)
        for _ in range(300):
            aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        if ref is None:
            ref = C.clone()
        same = bool(torch.equal(ref, C))
        print("round %d  shape %s  tile %3d  %7.2f us  %6.1f TF/s  %s" % (rnd, sys.argv[1], cfg, us, 2.0 * M * N * K / us * 1e-6, "same bits as first" if same else "bits differ from first"))
aslp.lib.aslp_gemm_split16_tile(0)
