"""Runs one GEMM shape N times (for rocprofv3 counter collection)."""
import os, sys, torch
sys.path.insert(0, '.')
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
tA, tB, M, N, K = [int(x) for x in os.environ.get('SHAPE', '0,1,1024,2048,2048').split(',')]
aslp.lib.aslp_gemm_force_tile(int(os.environ.get('TILE', '0')))
A = torch.randn((K, M) if tA else (M, K), device=dev)
B = torch.randn((N, K) if tB else (K, N), device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(int(os.environ.get('REPS', '10'))):
    aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C)
    if os.environ.get('REF') == '1':
        torch.mm(A.t() if tA else A, B.t() if tB else B, out=C)
torch.cuda.synchronize()
