"""BASELINE.json full-size checks through size-independent properties (the CPU oracle is too slow at these sizes):
products against a float64 product on the device, bit-exact gathers against index arithmetic, normalisation
invariants, linearity, and a full cfg2 training step that must reduce its loss."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tA,tB,M,N,K", [(0, 1, 1024, 2048, 2048), (0, 1, 1024, 2048, 440), (0, 1, 1024, 3000, 2048), (0, 0, 1024, 2048, 3000),
                                           (0, 0, 1024, 2048, 2048), (1, 0, 2048, 2048, 1024), (1, 0, 3000, 2048, 1024), (1, 0, 2048, 440, 1024),
                                           (0, 1, 1920, 2048, 512), (0, 0, 1920, 256, 2048),
                                           # cfg1 / cfg4's per-GPU minibatch of 256: the small-grid regime (a 256 x 2048 output is 128 tiles of
                                           # 64 x 64 for 256 CUs), all three layouts of a layer + the input and output layers
                                           (0, 1, 256, 2048, 2048), (0, 0, 256, 2048, 2048), (1, 0, 2048, 2048, 256), (0, 1, 256, 2048, 440),
                                           (0, 1, 256, 3000, 2048), (0, 0, 256, 2048, 3000), (1, 0, 3000, 2048, 256), (1, 0, 2048, 440, 256)])
def test_sgemm_fullsize_vs_float64(aslp, dev, tA, tB, M, N, K):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), device=dev, generator=g)
    B = torch.randn((N, K) if tB else (K, N), device=dev, generator=g)
    C0 = torch.randn(M, N, device=dev, generator=g)
    Cm = C0.clone()
    aslp.ops.sgemm(tA, tB, 0.75, A, B, 0.5, Cm)
    ref = 0.75 * ((A.t() if tA else A).double() @ (B.t() if tB else B).double()) + 0.5 * C0.double()
    rel = ((Cm.double() - ref).norm() / ref.norm()).item()
    assert rel < 2e-6, rel                      # fp32 accumulation error of a K-long dot product, far below 1e-4
    # linearity in A (exact arithmetic identity, fp32 rounding only)
    A2 = torch.randn_like(A)
    C1, C2, C12 = torch.empty_like(Cm), torch.empty_like(Cm), torch.empty_like(Cm)
    aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C1)
    aslp.ops.sgemm(tA, tB, 1.0, A2, B, 0.0, C2)
    aslp.ops.sgemm(tA, tB, 1.0, A + A2, B, 0.0, C12)
    assert ((C12 - (C1 + C2)).norm() / C12.norm()).item() < 1e-5
    # bit-reproducible run to run (no atomics, fixed summation order)
    C1b = torch.empty_like(Cm)
    aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C1b)
    assert torch.equal(C1, C1b)


def test_randomizer_cache_fullsize_bit_exact(aslp, dev):
    """cfg: randomizer cache 32768 x 440: the shuffled cache is an exact row gather"""
    rows, cols, mb = 32768, 440, 1024
    g = torch.Generator(device=dev).manual_seed(1)
    m = torch.randn(rows + 100, cols, device=dev, generator=g)
    r = aslp.MatrixRandomizer(randomizer_size=rows, minibatch_size=mb)
    r.AddData(m)
    assert r.IsFull()
    mask = aslp.randomizer_mask(rows + 100, seed=777)
    assert np.array_equal(np.sort(mask), np.arange(rows + 100))
    r.Randomize(mask)
    want = m[torch.from_numpy(mask.astype(np.int64)).to(dev)]
    i = 0
    while not r.Done():
        assert torch.equal(r.Value(), want[i * mb:(i + 1) * mb])
        r.Next()
        i += 1
    assert i == (rows + 100) // mb


def test_splice_fullsize_bit_exact(aslp, dev):
    rows, dim = 32768, 40
    offs = list(range(-5, 6))
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(rows, dim, device=dev, generator=g)
    y = torch.empty(rows, dim * len(offs), device=dev)
    aslp.ops.splice(y, x, torch.tensor(offs, dtype=torch.int32, device=dev))
    idx = torch.arange(rows, device=dev)
    want = torch.cat([x[(idx + o).clamp(0, rows - 1)] for o in offs], dim=1)
    assert torch.equal(y, want)


def test_softmax_and_bn_invariants_fullsize(aslp, dev):
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(1024, 3000, device=dev, generator=g) * 3
    y = torch.empty_like(x)
    aslp.ops.softmax(y, x)
    assert (y.sum(1) - 1).abs().max().item() < 1e-5 and (y >= 0).all()
    assert torch.equal(y.argmax(1), x.argmax(1))
    # BatchNorm forward: per-column mean 0 / variance 1 of xhat, out = xhat*scale + shift
    xb = torch.randn(1024, 2048, device=dev, generator=g) * 2.5 + 1.5
    scale = torch.rand(2048, device=dev, generator=g) + 0.5
    shift = torch.randn(2048, device=dev, generator=g)
    out, xhat = torch.empty_like(xb), torch.empty_like(xb)
    mean, inv_std = torch.empty(2048, device=dev), torch.empty(2048, device=dev)
    aslp.ops.bn_forward(xb, out, xhat, scale, shift, mean, inv_std)
    assert xhat.mean(0).abs().max().item() < 1e-5
    assert (xhat.var(0, unbiased=False) - 1).abs().max().item() < 1e-4
    assert ((out - (xhat * scale + shift)).abs().max().item()) < 1e-5
    assert ((mean - xb.mean(0)).abs().max().item()) < 1e-5


def test_cfg2_training_reduces_loss(aslp, dev):
    """Full BASELINE cfg2 net (5 x 2048 sigmoid + BatchNorm, 440 -> 3000, minibatch 1024): repeated steps on one
    minibatch must drive its cross-entropy down and keep every parameter finite."""
    lines = ["<NnetProto>"]
    d = 440
    for _ in range(5):
        lines += ["<AffineTransform> <InputDim> %d <OutputDim> 2048 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.04" % d,
                  "<BatchNormalization> <InputDim> 2048 <OutputDim> 2048", "<Sigmoid> <InputDim> 2048 <OutputDim> 2048"]
        d = 2048
    lines += ["<AffineTransform> <InputDim> 2048 <OutputDim> 3000 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04",
              "<Softmax> <InputDim> 3000 <OutputDim> 3000", "</NnetProto>"]
    net = aslp.Nnet.Init("\n".join(lines) + "\n", seed=777)
    assert net.NumParams() == 440 * 2048 + 4 * 2048 * 2048 + 2048 * 3000 + 5 * 2048 + 3000 + 5 * 2 * 2048
    net.SetTrainOptions(learn_rate=0.0005, momentum=0.0)
    g = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn(1024, 440, device=dev, generator=g)
    lab = torch.randint(0, 3000, (1024,), device=dev, generator=g, dtype=torch.int32)
    losses = []
    for _ in range(6):
        xe = aslp.Xent()
        net.TrainStepXent(xe, x, lab)
        st = xe.GetStats()
        losses.append((st["loss"] - st["entropy"]) / st["frames"])
    assert losses[-1] < losses[0] - 0.05, losses
    assert np.isfinite(net.GetParams()).all()


def test_ctc_fullsize_properties(aslp, dev):
    """32 utterances x up to 800 frames x alphabet 128 (SURVEY 8d CTC variant): gradient rows sum to ~0 over the
    alphabet (softmax Jacobian), padding frames get exactly zero, costs are positive and finite, and costs / gradients
    of an utterance do not depend on what else is in the minibatch."""
    A, mb, maxT = 128, 32, 800
    rng = np.random.default_rng(7)
    in_len = rng.integers(200, maxT + 1, mb).astype(np.int32)
    in_len[0] = maxT
    labels = [[int(v) for v in rng.integers(1, A, int(t) // 4)] for t in in_len]
    acts = torch.from_numpy(rng.random((maxT * mb, A)).astype(np.float32)).to(dev)
    costs, grads = aslp.ops.ctc_loss(acts, labels, in_len)
    assert np.isfinite(costs).all() and (costs > 0).all()
    g3 = grads.view(maxT, mb, A)
    assert g3.sum(2).abs().max().item() < 2e-2          # fp32 at |log p(z|x)| ~ 1e3
    for s in (1, 7, 31):
        assert torch.count_nonzero(g3[int(in_len[s]):, s]).item() == 0
    # utterance 5 alone
    a5 = acts.view(maxT, mb, A)[:int(in_len[5]), 5].contiguous()
    c5, g5 = aslp.ops.ctc_loss(a5, [labels[5]], [int(in_len[5])])
    assert abs(c5[0] - costs[5]) < 1e-4 * abs(costs[5])
    assert ((g5 - g3[:int(in_len[5]), 5]).norm() / g5.norm()).item() < 1e-4


@pytest.mark.parametrize("tA,tB,M,N,K,mmt", [(1, 0, 2048, 2048, 1024, 0.9), (1, 0, 3000, 2048, 1024, 0.9), (1, 0, 2048, 2048, 1024, 0.0),
                                             (0, 1, 2048, 2048, 1024, 0.5), (0, 0, 1984, 2176, 640, 0.9), (1, 1, 2048, 2048, 704, 0.9)])
def test_weight_gradient_gemm_with_full_epilogue(aslp, dev, tA, tB, M, N, K, mmt):
    """The weight-gradient product of cfg2 with everything AffineTransform::Update folds into it (momentum on the gradient
    buffer, clip, W += -lr * G, bias gradient + bias step from the column sums) at full size: against float64, and bit for bit
    across tile configurations (64x128 / 8 waves with 3 and 4 LDS stages, 64x64 / 4 waves): every kernel adds the K terms of an
    output in the same order and spells the epilogue with the same explicit fused multiply-adds (gemm_common.h), so the tile
    heuristic can change without changing a trained model.  Partial edge tiles (M = 3000), every operand layout."""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), device=dev, generator=g)
    B = torch.randn((N, K) if tB else (K, N), device=dev, generator=g)
    G0 = torch.randn(M, N, device=dev, generator=g)
    W0 = torch.randn(M, N, device=dev, generator=g)
    bc0 = torch.randn(M, device=dev, generator=g)
    b0 = torch.randn(M, device=dev, generator=g)
    clip, lr = 60.0, -0.01
    outs = []
    aslp.lib.aslp_gemm_split16(0)   # this test is about the fp32-instruction kernels (ASLP_GEMM_SPLIT_F16=1 in the environment would take the unforced call)
    for force in (0, 212, 213, 207):
        Gd, Wd, bc, b = G0.clone(), W0.clone(), bc0.clone(), b0.clone()
        if tA:
            ep = aslp._lib.GemmEpilogue(None, clip, Wd.data_ptr(), N, lr, None, 0, 0, bc.data_ptr(), 0.9, b.data_ptr(), -0.02)
        else:
            ep = aslp._lib.GemmEpilogue(None, clip, Wd.data_ptr(), N, lr, None, 0, 0)
        aslp.lib.aslp_gemm_force_tile(force)
        try:
            aslp.ops.sgemm(tA, tB, 1.0, A, B, mmt, Gd, ep)
        finally:
            aslp.lib.aslp_gemm_force_tile(0)
        if force:
            assert aslp.lib.aslp_gemm_last_tile() == force
        outs.append((Gd, Wd, bc, b))
    aslp.lib.aslp_gemm_split16(-1)
    opA, opB = (A.t() if tA else A).double(), (B.t() if tB else B).double()
    Gref = (opA @ opB + mmt * G0.double()).clamp(-clip, clip)
    assert (Gref.abs() == clip).any() and (Gref.abs() < clip).any()
    Gd, Wd, bc, b = outs[0]
    rel = lambda x, r: ((x.double() - r).norm() / r.norm()).item()
    assert rel(Gd, Gref) < 2e-6
    assert rel(Wd, W0.double() + lr * Gref) < 2e-6
    if tA:
        bc_ref = A.double().sum(0) + 0.9 * bc0.double()
        assert rel(bc, bc_ref) < 2e-6 and rel(b, b0.double() - 0.02 * bc_ref) < 2e-6
    for other in outs[1:]:
        for x, y in zip(outs[0][:2], other[:2]):
            assert torch.equal(x, y)
