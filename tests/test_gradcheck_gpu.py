"""Finite-difference check of the backward passes, independent of the CPU oracle: for a random linear functional L = sum(out .* R)
of the network output, the gradient the engine applies (parameters before - after one Backpropagate at learn-rate 1, momentum 0, no
clipping) must equal (L(p + eps) - L(p - eps)) / 2 eps for individual parameters perturbed in place on the device.  The oracle
restates the reference's formulas; this test asks the calculus instead, through the same kernels the training step runs: the
persistent LSTM / GRU recurrences (forward for L, backward for the gradient), the batched products and fused epilogues around
them, and the BatchNormalization + Sigmoid fusions of the DNN path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NETS = {
    "GruStreams": ("<GruStreams> <InputDim> 10 <OutputDim> 16 <ParamScale> 0.3", 16, 16),
    "LstmProjectedStreams": ("<LstmProjectedStreams> <InputDim> 10 <OutputDim> 8 <CellDim> 16 <ParamScale> 0.3", 8, 16),
    "LstmCifgProjectedStreams": ("<LstmCifgProjectedStreams> <InputDim> 10 <OutputDim> 8 <CellDim> 16 <ParamScale> 0.3", 8, 16),
    "Lstm": ("<Lstm> <InputDim> 10 <OutputDim> 16 <ParamScale> 0.3", 16, 16),
    "BLstmProjectedStreams": ("<BLstmProjectedStreams> <InputDim> 10 <OutputDim> 16 <CellDim> 16 <ParamScale> 0.3", 16, 16),
    "BLstm": ("<BLstm> <InputDim> 10 <OutputDim> 32 <ParamScale> 0.3", 32, 16),
    # latency-controlled: with the streams reset and the whole batch as the chunk (no right context) the carried history is zero and
    # every frame counts -- the arithmetic of cfg3's component, where the gradient is exact
    "BLstmProjectedStreamsLC": ("<BLstmProjectedStreamsLC> <InputDim> 10 <OutputDim> 16 <CellDim> 16 <ParamScale> 0.3", 16, 16),
}


def check(aslp, dev, net, x, R, before_forward, n_probe=14, eps=1e-2, seed=0):
    from parallel_model import alias_device_params   # (tests/parallel_model.py: aliases device memory as torch tensors)
    params = alias_device_params(net.GetGpuParams())
    flat = lambda: torch.cat([p.reshape(-1) for p in params]).clone()

    def L():
        before_forward()
        return (net.Propagate(x).double() * R).sum().item()

    p0 = flat()
    net.SetTrainOptions(learn_rate=1.0, momentum=0.0)
    L()
    net.Backpropagate(R.float().contiguous())
    grad = (p0 - flat()).double()
    # put the parameters back
    off = 0
    for p in params:
        p.copy_(p0[off:off + p.numel()])
        off += p.numel()
    # GetGpuParams spans whole pitched rows (rows x stride, as the reference's: nnet-affine-transform.h:166-170): the padding floats behind
    # every row are never written by anybody and hold whatever the allocator's block held before -- NaN bit patterns when it was a
    # sentinel-filled recurrent buffer of an earlier test.  They are no parameters: leave them out (a parameter itself is always finite).
    real = torch.isfinite(p0)
    grad = torch.where(real, grad, torch.zeros_like(grad))
    assert torch.isfinite(grad).all() and grad.abs().max() > 1e-3
    big = (grad.abs() > 0.2 * grad.abs().max()).nonzero().flatten().cpu().numpy()
    small = (grad.abs() > 0).nonzero().flatten().cpu().numpy()
    rng = np.random.default_rng(seed)
    probes = list(rng.choice(big, min(n_probe // 2, len(big)), replace=False)) + list(rng.choice(small, min(n_probe // 2, len(small)), replace=False))
    bounds = np.cumsum([0] + [p.numel() for p in params])
    worst = 0.0
    for k in probes:
        t = int(np.searchsorted(bounds, k, side="right") - 1)
        j = int(k - bounds[t])
        orig = params[t][j].item()
        params[t][j] = orig + eps
        lp = L()
        params[t][j] = orig - eps
        lm = L()
        params[t][j] = orig
        fd, an = (lp - lm) / (2 * eps), grad[k].item()
        err = abs(fd - an) / (max(abs(fd), abs(an)) + 5e-2 * grad.abs().max().item())
        worst = max(worst, err)
        assert err < 3e-2, (k, fd, an)
    return worst


@pytest.mark.parametrize("name", list(NETS))
def test_recurrent_gradients_match_finite_differences(aslp, dev, name):
    line, out_dim, cells = NETS[name]
    S, T = 8, 7
    net = aslp.Nnet.Init("<NnetProto>\n%s\n</NnetProto>\n" % line, seed=3)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(T * S, 10, device=dev, generator=g)
    R = torch.randn(T * S, out_dim, device=dev, generator=g).double()
    lens = [T, T - 2, T, 3, T, T - 1, 1, T]

    if name.endswith("LC"):
        net.SetChunkSize(T)

    def before():
        net.ResetLstmStreams([1] * S)
        net.SetSeqLengths(lens)
    if name.startswith("BLstm") and not name.endswith("LC"):   # frames past an utterance's end carry no gradient (nnet-blstm-projected-streams.h:654-657)
        m = torch.zeros(T, S, 1, device=dev, dtype=torch.float64)
        for s, n in enumerate(lens):
            m[:n, s] = 1
        R = (R.view(T, S, -1) * m).view(T * S, -1)
    check(aslp, dev, net, x, R, before)


@pytest.mark.parametrize("rows,hidden", [(256, 64), (100, 32)])
def test_dnn_with_batchnorm_gradients_match_finite_differences(aslp, dev, rows, hidden):
    proto = """<NnetProto>
<AffineTransform> <InputDim> 12 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.5 <ParamStddev> 0.4
<BatchNormalization> <InputDim> %d <OutputDim> %d
<Sigmoid> <InputDim> %d <OutputDim> %d
<AffineTransform> <InputDim> %d <OutputDim> 9 <BiasMean> 0.0 <BiasRange> 0.5 <ParamStddev> 0.4
</NnetProto>
""" % (hidden, hidden, hidden, hidden, hidden, hidden)
    net = aslp.Nnet.Init(proto, seed=5)
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(rows, 12, device=dev, generator=g)
    R = torch.randn(rows, 9, device=dev, generator=g).double()
    check(aslp, dev, net, x, R, lambda: None, eps=5e-3)


def test_compact_fsmn_gradients_match_finite_differences(aslp, dev):
    proto = """<NnetProto>
<AffineTransform> <InputDim> 6 <OutputDim> 12 <BiasMean> 0.0 <BiasRange> 0.5 <ParamStddev> 0.4
<CompactFsmn> <InputDim> 12 <OutputDim> 12 <PastContext> 3 <FutureContext> 2
<AffineTransform> <InputDim> 12 <OutputDim> 5 <BiasMean> 0.0 <BiasRange> 0.5 <ParamStddev> 0.4
</NnetProto>
"""
    net = aslp.Nnet.Init(proto, seed=7)
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(40, 6, device=dev, generator=g)
    R = torch.randn(40, 5, device=dev, generator=g).double()
    check(aslp, dev, net, x, R, lambda: None, eps=5e-3)


def test_row_convolution_gradients_match_finite_differences(aslp, dev):
    """RowConvolution (nnet-row-convolution.cc): the reference drops the diff that lands on the replicated tail frames of a stream, so
    its tap gradient is exact only for functionals that ignore the last K frames -- R is zero there."""
    D, K, T, S = 8, 3, 12, 4
    proto = """<NnetProto>
<AffineTransform> <InputDim> 5 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.5 <ParamStddev> 0.4
<RowConvolution> <InputDim> %d <OutputDim> %d <FutureContext> %d
</NnetProto>
""" % (D, D, D, K)
    net = aslp.Nnet.Init(proto, seed=11)
    g = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn(T * S, 5, device=dev, generator=g)
    R = torch.randn(T, S, D, device=dev, generator=g).double()
    R[T - K:] = 0
    check(aslp, dev, net, x, R.view(T * S, D), lambda: net.SetSeqLengths([T] * S), eps=5e-3)
