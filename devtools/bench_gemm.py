"""Times aslp_sgemm on the BASELINE cfg shapes (HIP events on torch's stream)."""
import sys, torch
sys.path.insert(0, '.')
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
shapes = []
for mb in (1024, 256):
    shapes += [('NT fwd  ', 0, 1, mb, 2048, 2048), ('NT fwd1 ', 0, 1, mb, 2048, 440), ('NT fwdL ', 0, 1, mb, 3000, 2048),
               ('NN bwd  ', 0, 0, mb, 2048, 2048), ('NN bwdL ', 0, 0, mb, 2048, 3000),
               ('TN wgrad', 1, 0, 2048, 2048, mb), ('TN wgrdL', 1, 0, 3000, 2048, mb), ('TN wgrd1', 1, 0, 2048, 440, mb)]
shapes += [('NT lstm x', 0, 1, 60 * 32, 2048, 512), ('NT lstm r', 0, 1, 32, 2048, 256), ('NT lstm p', 0, 1, 32, 256, 512)]
for name, tA, tB, M, N, K in shapes:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(3): aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    # torch (rocBLAS/hipBLASLt) comparison line
    At = A.t() if tA else A; Bt = B.t() if tB else B
    for _ in range(3): torch.mm(At, Bt, out=C)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): torch.mm(At, Bt, out=C)
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / n
    fl = 2.0 * M * N * K
    print(f'{name} M={M:5d} N={N:5d} K={K:5d}  aslp {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF | torch.mm {ms2*1e3:8.1f} us {fl/ms2/1e9:7.1f} TF')
