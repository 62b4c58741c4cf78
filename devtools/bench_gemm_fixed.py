"""Fixed cost of a layer product: the same M x N at K = 64, 256, 1024, 2048 -- time against K extrapolates to the K = 0 intercept
(launch + prologue + epilogue), for this repo's kernel and the vendor library's (torch.mm).  Back to back, warm chip."""
import sys, torch
sys.path.insert(0, '.')
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
w = torch.randn(4096, 4096, device=dev)
for _ in range(600): torch.mm(w, w)
torch.cuda.synchronize()


def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, tA, tB, M, N in (("NN", 0, 0, 1024, 2048), ("NT", 0, 1, 1024, 2048), ("TN", 1, 0, 2048, 2048)):
    line = name
    for K in (64, 256, 1024, 2048):
        A = torch.randn((K, M) if tA else (M, K), device=dev)
        B = torch.randn((N, K) if tB else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        ours = timeit(lambda: aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C))
        At, Bt = (A.t() if tA else A), (B.t() if tB else B)
        lib = timeit(lambda: torch.mm(At, Bt, out=C))
        line += "  K=%4d: %6.1f / %6.1f us" % (K, ours, lib)
    print(line, " (this repo / torch.mm)")
