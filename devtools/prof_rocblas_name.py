"""Which kernel does the vendor library pick for the layer products?  (rocprofv3 --kernel-trace of a few torch.mm calls; the kernel
names carry the macro tile and the main Tensile parameters.)  Usage under rocprofv3: python3 devtools/prof_rocblas_name.py"""
import torch
dev = torch.device("cuda:0")
w = torch.randn(4096, 4096, device=dev)
for _ in range(300): torch.mm(w, w)
torch.cuda.synchronize()
for name, (M, N, K, ta, tb) in {"NN": (1024, 2048, 2048, 0, 0), "NT": (1024, 2048, 2048, 0, 1), "TN": (2048, 2048, 1024, 1, 0)}.items():
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    for _ in range(20):
        torch.mm(A.t() if ta else A, B.t() if tb else B, out=C)
torch.cuda.synchronize()
