"""Times aslp_sgemm on the BASELINE cfg shapes (HIP events on torch's stream).
TILES=0,3,4 python devtools/bench_gemm.py  sweeps forced tile configs (0 = heuristic)."""
import os, sys, torch
sys.path.insert(0, '.')
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
_w = torch.randn(4096, 4096, device=dev)
for _ in range(800): torch.mm(_w, _w)  # ~1 s: the clocks take several hundred ms to settle after idle; shorter warm-ups bias whatever is timed first
torch.cuda.synchronize()
shapes = []
for mb in [int(m) for m in os.environ.get('MBS', '1024,256').split(',')]:
    shapes += [('NT fwd  ', 0, 1, mb, 2048, 2048), ('NT fwd1 ', 0, 1, mb, 2048, 440), ('NT fwdL ', 0, 1, mb, 3000, 2048),
               ('NN bwd  ', 0, 0, mb, 2048, 2048), ('NN bwdL ', 0, 0, mb, 2048, 3000),
               ('TN wgrad', 1, 0, 2048, 2048, mb), ('TN wgrdL', 1, 0, 3000, 2048, mb), ('TN wgrd1', 1, 0, 2048, 440, mb)]
if os.environ.get('LSTM', '1') == '1':
    shapes += [('NT lstm x', 0, 1, 60 * 32, 2048, 512), ('NT lstm r', 0, 1, 32, 2048, 256), ('NT lstm p', 0, 1, 32, 256, 512)]
if os.environ.get('LC', '0') == '1':  # the whole-sequence products of one LC-BLSTM direction (cfg3: T*S = 1920 rows, C = 512, R = 256)
    shapes = [('NT x->gates ', 0, 1, 1920, 2048, 512), ('NT m->r     ', 0, 1, 1920, 256, 512), ('NN d_m      ', 0, 0, 1920, 512, 256),
              ('NN d_r      ', 0, 0, 1920, 256, 2048), ('NN in_diff  ', 0, 0, 1920, 512, 2048), ('TN w_x grad ', 1, 0, 2048, 512, 1920),
              ('TN w_r grad ', 1, 0, 2048, 256, 1920), ('TN w_rm grad', 1, 0, 256, 512, 1920), ('NN W_eff    ', 0, 0, 2048, 512, 256),
              ('TT W_eff^T  ', 1, 1, 512, 2048, 256)]
cfgs = [int(c) for c in os.environ.get('TILES', '0').split(',')]
ref = os.environ.get('REF', '1') == '1'


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, tA, tB, M, N, K in shapes:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    line = f'{name} M={M:5d} N={N:5d} K={K:5d} |'
    for cfg in cfgs:
        aslp.lib.aslp_gemm_force_tile(cfg)
        ms = timeit(lambda: aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C))
        line += f' cfg{cfg}: {ms*1e3:7.1f}us {fl/ms/1e9:6.1f}TF |'
    if ref:
        At = A.t() if tA else A; Bt = B.t() if tB else B
        ms2 = timeit(lambda: torch.mm(At, Bt, out=C))
        line += f' torch.mm {ms2*1e3:7.1f}us {fl/ms2/1e9:6.1f}TF'
    print(line)
