"""Does a power-of-two leading dimension cost the K-contiguous GEMM operands anything (channel camping of the 8-row
LDS-DMA units)?  Times NT / NN / TN products of the cfg2 layer with operands whose rows are padded by PAD floats."""
import os, sys, torch
sys.path.insert(0, '.')
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
_w = torch.randn(4096, 4096, device=dev)
for _ in range(800): torch.mm(_w, _w)  # ~1 s: the clocks take several hundred ms to settle after idle; shorter warm-ups bias whatever is timed first
torch.cuda.synchronize()


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def padded(r, c, pad):
    return torch.randn(r, c + pad, device=dev)[:, :c]


for name, tA, tB, M, N, K in [('NT fwd', 0, 1, 1024, 2048, 2048), ('NT fwdL', 0, 1, 1024, 3000, 2048), ('NN bwd', 0, 0, 1024, 2048, 2048), ('NN bwdL', 0, 0, 1024, 2048, 3000), ('TN wgrad', 1, 0, 2048, 2048, 1024)]:
    line = '%-9s' % name
    for pad in [int(x) for x in os.environ.get('PADS', '0,16,32,64,128').split(',')]:
        A = padded(*((K, M) if tA else (M, K)), pad)
        B = padded(*((N, K) if tB else (K, N)), pad)
        C = padded(M, N, pad)
        for cfg in [int(c) for c in os.environ.get('TILES', '0').split(',')]:
            aslp.lib.aslp_gemm_force_tile(cfg)
            ms = timeit(lambda: aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C))
            line += ' pad%-3d cfg%d %6.1fus %5.1fTF |' % (pad, cfg, ms * 1e3, 2.0 * M * N * K / ms / 1e9)
    print(line)
