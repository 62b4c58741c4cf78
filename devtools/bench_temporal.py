"""Times CompactFsmn (D=512, 30+30 taps, T=800 per utterance) and RowConvolution (D=512, FutureContext 20, T=800, S=32)
training steps through the engine (SURVEY 8d cfg5 swaps) and reports achieved algorithmic GB/s."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


D, T = 512, 800
net = aslp.Nnet.Init("<NnetProto>\n<CompactFsmn> <InputDim> 512 <OutputDim> 512 <PastContext> 30 <FutureContext> 30 <LearnRateCoef> 1.0\n</NnetProto>\n")
net.SetTrainOptions(learn_rate=1e-5)
x = torch.randn(T, D, device=dev); od = torch.randn(T, D, device=dev) * 0.01
def fsmn_step():
    net.Propagate(x); net.Backpropagate(od, want_in_diff=True)
s = timeit(fsmn_step)
bytes_alg = 4 * D * T * (2 + 5)   # 4*512*(2 fwd + 5 bwd) per frame
print("CompactFsmn  T=%d D=%d: %.1f us/step (fwd+bwd+update), %.0f k frames/s, algorithmic %.1f GB/s, %.1f GFLOP/s" % (
    T, D, s * 1e6, T / s / 1e3, bytes_alg / s / 1e9, 3 * 2 * 61 * D * T / s / 1e9))

S, K = 32, 20
net2 = aslp.Nnet.Init("<NnetProto>\n<RowConvolution> <InputDim> 512 <OutputDim> 512 <FutureContext> 20\n</NnetProto>\n")
net2.SetTrainOptions(learn_rate=1e-5, momentum=0.9)
lens = np.random.default_rng(0).integers(T // 2, T + 1, S); lens[0] = T
net2.SetSeqLengths(lens)
x2 = torch.randn(T * S, D, device=dev); od2 = torch.randn(T * S, D, device=dev) * 0.01
def rc_step():
    net2.Propagate(x2); net2.Backpropagate(od2, want_in_diff=True)
s2 = timeit(rc_step, 20)
print("RowConvolution T=%d S=%d D=%d K=%d: %.1f us/step, %.0f k rows/s, algorithmic (7 tensor passes) %.1f GB/s, %.1f GFLOP/s" % (
    T, S, D, K, s2 * 1e6, T * S / s2 / 1e3, 4 * D * T * S * 7 / s2 / 1e9, 3 * 2 * D * (K + 1) * T * S / s2 / 1e9))

# ---- the same passes at the kernel level (ops bindings): no engine copies of the input / output / diffs around them
def timeit_ev(fn, reps=10, batches=21):
    for _ in range(10): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(batches + 1)]
    torch.cuda.synchronize(); ev[0].record()
    for b in range(batches):
        for _ in range(reps): fn()
        ev[b + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[b].elapsed_time(ev[b + 1]) / reps for b in range(batches)])) * 1e-3

P = F = 30
coef = (torch.randn(P + F + 1, D, device=dev) * 0.05); corr = torch.zeros_like(coef)
o, idf = torch.empty_like(x), torch.empty_like(x)
def fsmn_ops():
    aslp.ops.fsmn_forward(o, x, coef, P, F)
    aslp.ops.fsmn_backward(idf, corr, coef, x, od, P, F, 0.0, 1e-5)
s = timeit_ev(fsmn_ops)
print("CompactFsmn  ops level (2 launches: forward; in-diff + tap gradients + update): %.1f us, algorithmic %.1f GB/s" % (s * 1e6, bytes_alg / s / 1e9))
w = torch.randn(D * (K + 1), device=dev) * 0.1; wd = torch.zeros_like(w); wc = torch.zeros_like(w)
sl = torch.from_numpy(lens.astype(np.int32)).to(dev)
o2, idf2 = torch.empty_like(x2), torch.empty_like(x2)
def rc_ops():
    aslp.ops.rowconv_forward(o2, x2, w, sl, K)
    aslp.ops.rowconv_backward(idf2, wd, x2, od2, w, sl, K, wc, 0.9, 1e-5)
s2 = timeit_ev(rc_ops, reps=5)
print("RowConvolution ops level (3 launches: forward; in-diff + tap partials; finish + update): %.1f us, algorithmic (7 tensor passes) %.1f GB/s = %.2f of 8 TB/s" % (
    s2 * 1e6, 4 * D * T * S * 7 / s2 / 1e9, 4 * D * T * S * 7 / s2 / 8e12))
