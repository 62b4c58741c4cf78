"""csrc/gemm_split16.hip: aslp_sgemm_ex on the fp16 matrix instruction with every fp32 operand carried as two fp16 pieces behind a
power-of-two scale of its matrix (the default; aslp_gemm_split16(0) / ASLP_GEMM_SPLIT_F16=0 keep the fp32 instruction).  The claim is fp32 accuracy, so the bar is
the fp32-instruction kernels' own: against a float64 product the error relative to sum |a||b| must not exceed theirs by more than
rounding noise, at any magnitude of the operands, in every operand layout, with every epilogue feature, on ragged sizes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SPLIT_TILES = (304, 308, 311, 328, 351)   # gemm_s16_glds 32x64 / 64x128 / 128x128, gemm_s16_ks128, gemm_s16_pc 128x128 (csrc/gemm_split16.hip)


@pytest.fixture(autouse=True)
def split_off_afterwards(aslp):
    yield
    aslp.lib.aslp_gemm_split16(-1)
    aslp.lib.aslp_gemm_split16_tile(-1)


def products(aslp, tA, tB, A, B, alpha=1.0, beta=0.0, C0=None, ep=None):
    out = []
    for on in (0, 1):
        aslp.lib.aslp_gemm_split16(on)
        C = C0.clone() if C0 is not None else torch.zeros((A.shape[1] if tA else A.shape[0]), (B.shape[0] if tB else B.shape[1]), device=A.device)
        aslp.ops.sgemm(tA, tB, alpha, A, B, beta, C, ep(on) if ep else None)
        if on:
            assert aslp.lib.aslp_gemm_last_tile() in SPLIT_TILES      # a split kernel really ran
        out.append(C)
    aslp.lib.aslp_gemm_split16(-1)
    return out


def err_vs_double(C, A, B, tA, tB):
    opA, opB = (A.t() if tA else A).double(), (B.t() if tB else B).double()
    return (((C.double() - opA @ opB).abs()) / (opA.abs() @ opB.abs())).max().item()


@pytest.mark.parametrize("tA,tB,M,N,K", [(0, 1, 1024, 2048, 2048), (0, 0, 1024, 2048, 2048), (1, 0, 2048, 2048, 1024), (1, 1, 512, 640, 768),
                                         (0, 1, 1000, 3000, 440), (1, 0, 3000, 2048, 1024), (0, 0, 132, 260, 68), (1, 0, 436, 128, 2052),
                                         (0, 1, 256, 2048, 2048), (0, 0, 256, 2048, 3000), (0, 1, 192, 1920, 1028)])   # (minibatch 256: K split over workgroups)
def test_as_accurate_as_the_fp32_instruction(aslp, dev, tA, tB, M, N, K):
    g = torch.Generator(device=dev).manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((K, M) if tA else (M, K), device=dev, generator=g)
    B = torch.randn((N, K) if tB else (K, N), device=dev, generator=g)
    c32, c16 = products(aslp, tA, tB, A, B)
    e32, e16 = err_vs_double(c32, A, B, tA, tB), err_vs_double(c16, A, B, tA, tB)
    assert e16 <= max(e32, 2.5e-7), (e16, e32)          # measured: 1.0-1.4e-7 against 3-4e-7 of the fp32 instruction
    assert ((c16 - c32).norm() / c32.norm()).item() < 1e-6
    # bit-reproducible run to run (no atomics anywhere: two-level maximum, fixed summation order)
    aslp.lib.aslp_gemm_split16(1)
    again = torch.zeros_like(c16)
    aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, again)
    assert torch.equal(again, c16)


@pytest.mark.parametrize("tB", [1, 0])
def test_minibatch_256_products_run_whole_on_small_tiles(aslp, dev, tB):
    """256 x 2048 outputs with a long reduction: 256 workgroups of 32 x 64 (tile 304) carry the whole K in one launch instead of K chunks +
    a second launch.  Same k order and instruction sequence per output element as the 64 x 128 tile: the rows are bit-identical to the
    same rows inside a 1024-row product"""
    g = torch.Generator(device=dev).manual_seed(77 + tB)
    A = torch.randn(1024, 2048, device=dev, generator=g)
    B = torch.randn(2048, 2048, device=dev, generator=g) * 0.03
    aslp.lib.aslp_gemm_split16(1)
    pA, pB = aslp.ops.Planes(A), aslp.ops.Planes(B)
    big = torch.zeros(1024, 2048, device=dev)
    aslp.ops.sgemm_planes(0, tB, 1.0, A, pA, B, pB, 0.0, big)
    assert aslp.lib.aslp_gemm_last_tile() == 308
    A4 = A[:256].contiguous()
    pA4 = aslp.ops.Planes(A4)
    small = torch.zeros(256, 2048, device=dev)
    aslp.ops.sgemm_planes(0, tB, 1.0, A4, pA4, B, pB, 0.0, small)
    assert aslp.lib.aslp_gemm_last_tile() == 304
    assert err_vs_double(small, A4, B, 0, tB) <= 2.5e-7
    # (the 256 rows' own maximum may give their planes another power-of-two scale than the 1024 rows': exact either way, so the bits agree
    # whenever no piece leaves fp16's normal range -- N(0, 1) data)
    assert torch.equal(small, big[:256])


@pytest.mark.parametrize("M,N,K", [(1024, 2048, 2048), (1000, 3000, 440), (256, 2048, 2048), (1920, 512, 2048), (132, 260, 68), (1088, 1984, 1028)])
def test_one_pair_of_planes_serves_every_product(aslp, dev, M, N, K):
    """x [M x K], W [N x K], dy [M x N] as planes in their own layout: the forward product x W^T (both reduction-contiguous), the in-diff
    dy W (W read with the transposing LDS load) and the weight gradient dy^T x (both operands read that way) from the same three pairs
    of planes -- each bit-identical to the call that converts its operands itself, and as close to float64 as that one"""
    g = torch.Generator(device=dev).manual_seed(M + 5 * N + 11 * K)
    x = torch.randn(M, K, device=dev, generator=g)
    W = torch.randn(N, K, device=dev, generator=g) * 0.05
    dy = torch.randn(M, N, device=dev, generator=g) * 1e-3
    aslp.lib.aslp_gemm_split16(1)
    px, pW, pdy = aslp.ops.Planes(x), aslp.ops.Planes(W), aslp.ops.Planes(dy)
    for tA, tB, A, pa, B, pb, shape in ((0, 1, x, px, W, pW, (M, N)), (0, 0, dy, pdy, W, pW, (M, K)), (1, 0, dy, pdy, x, px, (N, K)),
                                        (1, 1, W, pW, dy, pdy, (K, M))):
        C1 = torch.zeros(shape, device=dev)
        aslp.ops.sgemm_planes(tA, tB, 1.0, A, pa, B, pb, 0.0, C1)
        assert aslp.lib.aslp_gemm_last_tile() in SPLIT_TILES or min(shape) < 128
        C2 = torch.zeros(shape, device=dev)
        aslp.ops.sgemm(tA, tB, 1.0, A, B, 0.0, C2)
        assert torch.equal(C1, C2), (tA, tB)
        C3 = torch.zeros(shape, device=dev)     # one operand prepared, the other converted inside the call
        aslp.ops.sgemm_planes(tA, tB, 1.0, A, pa, B, None, 0.0, C3)
        assert torch.equal(C1, C3), (tA, tB)
        assert err_vs_double(C1, A, B, tA, tB) <= 2.5e-7, (tA, tB, err_vs_double(C1, A, B, tA, tB))


@pytest.mark.parametrize("sa,sb", [(1e-12, 1.0), (1e-30, 1e8), (3e4, 2e-20), (1e15, 1e-15), (1.0, 1e-38)])
def test_any_magnitude(aslp, dev, sa, sb):
    """gradients of 1e-12, weights of 1e8, operands near the ends of fp32's range: the per-matrix power-of-two scale takes the magnitude out;
    rows a million times smaller than the matrix's largest keep their own 22 bits (fp16 is normal over 29 binades)"""
    g = torch.Generator(device=dev).manual_seed(11)
    A = torch.randn(512, 1024, device=dev, generator=g) * sa
    B = torch.randn(768, 1024, device=dev, generator=g) * sb
    A[5] *= 1e-6
    A[7, ::2] *= 3e-5
    B[9] *= 1e-7
    c32, c16 = products(aslp, 0, 1, A, B)
    assert torch.isfinite(c16).all()
    e32, e16 = err_vs_double(c32, A, B, 0, 1), err_vs_double(c16, A, B, 0, 1)
    assert e16 <= max(e32, 2.5e-7), (e16, e32)


def test_zero_and_nonfinite_operands(aslp, dev):
    A = torch.zeros(256, 512, device=dev)
    B = torch.randn(384, 512, device=dev)
    _, c16 = products(aslp, 0, 1, A, B)
    assert (c16 == 0).all()
    A[3, 4] = float("inf")
    A[9, 1] = float("nan")
    A[20:] = torch.randn(236, 512, device=dev)
    c32, c16 = products(aslp, 0, 1, A, B)
    assert torch.isnan(c16[9]).all() and not torch.isfinite(c16[3]).any()        # what the fp32 kernel gives for those rows
    ok = torch.ones(256, dtype=torch.bool, device=dev)
    ok[3] = ok[9] = False
    assert torch.isfinite(c16[ok]).all() and ((c16[ok] - c32[ok]).norm() / c32[ok].norm()).item() < 1e-6


@pytest.mark.parametrize("tA,tB,M,N,K,mmt", [(1, 0, 2048, 2048, 1024, 0.9), (1, 0, 3000, 2048, 1024, 0.0), (0, 1, 2048, 2048, 1024, 0.5), (0, 0, 1984, 2176, 640, 0.9),
                                             (0, 1, 256, 2048, 2048, 0.9), (1, 0, 2048, 2048, 256, 0.9)])
def test_full_epilogue(aslp, dev, tA, tB, M, N, K, mmt):
    """momentum on the gradient buffer, clip, W += -lr G, bias gradient + bias step from the column sums (tests/test_fullsize_gpu.py's case)"""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), device=dev, generator=g)
    B = torch.randn((N, K) if tB else (K, N), device=dev, generator=g)
    G0 = torch.randn(M, N, device=dev, generator=g)
    W0 = torch.randn(M, N, device=dev, generator=g)
    bc0, b0 = torch.randn(M, device=dev, generator=g), torch.randn(M, device=dev, generator=g)
    clip, lr = 60.0, -0.01
    res = []
    for on in (0, 1):
        Gd, Wd, bc, b = G0.clone(), W0.clone(), bc0.clone(), b0.clone()
        ep = (aslp._lib.GemmEpilogue(None, clip, Wd.data_ptr(), N, lr, None, 0, 0, bc.data_ptr(), 0.9, b.data_ptr(), -0.02) if tA else
              aslp._lib.GemmEpilogue(None, clip, Wd.data_ptr(), N, lr, None, 0, 0))
        aslp.lib.aslp_gemm_split16(on)
        aslp.ops.sgemm(tA, tB, 1.0, A, B, mmt, Gd, ep)
        res.append((Gd, Wd, bc, b))
    aslp.lib.aslp_gemm_split16(-1)
    opA, opB = (A.t() if tA else A).double(), (B.t() if tB else B).double()
    Gref = (opA @ opB + mmt * G0.double()).clamp(-clip, clip)
    rel = lambda x, r: ((x.double() - r).norm() / r.norm()).item()
    Gd, Wd, bc, b = res[1]
    assert rel(Gd, Gref) < 2e-6 and rel(Wd, W0.double() + lr * Gref) < 2e-6
    if tA:
        bc_ref = A.double().sum(0) + 0.9 * bc0.double()
        assert rel(bc, bc_ref) < 2e-6 and rel(b, b0.double() - 0.02 * bc_ref) < 2e-6
        assert rel(bc, res[0][2].double()) < 1e-6      # (column sums: a plain fp32 pass over A here, folded into the fp32 kernel's K loop there)
    assert rel(Gd, res[0][0].double()) < 1e-6


def test_cfg2_training_steps_agree(aslp, dev):
    """five training steps of a 3 x 1024 sigmoid DNN + BatchNorm at minibatch 1024 with the products on either instruction: the losses agree like
    two fp32 summation orders do"""
    proto = "<NnetProto>\n"
    d = 440
    for _ in range(3):
        proto += "<AffineTransform> <InputDim> %d <OutputDim> 1024 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.05\n" % d
        proto += "<BatchNormalization> <InputDim> 1024 <OutputDim> 1024\n<Sigmoid> <InputDim> 1024 <OutputDim> 1024\n"
        d = 1024
    proto += "<AffineTransform> <InputDim> 1024 <OutputDim> 3000 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.05\n<Softmax> <InputDim> 3000 <OutputDim> 3000\n</NnetProto>\n"
    losses = []
    for on in (0, 1):
        aslp.lib.aslp_gemm_split16(on)
        net = aslp.Nnet.Init(proto, seed=3)
        net.SetTrainOptions(learn_rate=1e-3, momentum=0.9)
        xe = aslp.Xent()
        g = torch.Generator(device="cpu").manual_seed(5)
        for step in range(5):
            x = torch.randn(1024, 440, generator=g).to(dev)
            lab = torch.randint(0, 3000, (1024,), generator=g, dtype=torch.int32).to(dev)
            net.TrainStepXent(xe, x, lab, torch.ones(1024, device=dev))
        st = xe.GetStats()
        losses.append((st["loss"] - st["entropy"]) / st["frames"])
        params = np.asarray(net.GetParams(), np.float32)
        losses.append(params)
    aslp.lib.aslp_gemm_split16(-1)
    l32, p32, l16, p16 = losses
    assert abs(l16 - l32) < 2e-6 * abs(l32), (l16, l32)
    assert np.linalg.norm(p16 - p32) < 2e-6 * np.linalg.norm(p32)


def _bn_dnn_proto(hidden=1024, layers=3, out=3000, bn=True):
    proto, d = "<NnetProto>\n", 440
    for _ in range(layers):
        proto += "<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.05\n" % (d, hidden)
        if bn:
            proto += "<BatchNormalization> <InputDim> %d <OutputDim> %d\n" % (hidden, hidden)
        proto += "<Sigmoid> <InputDim> %d <OutputDim> %d\n" % (hidden, hidden)
        d = hidden
    return proto + ("<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.05\n"
                    "<Softmax> <InputDim> %d <OutputDim> %d\n</NnetProto>\n" % (d, out, out, out))


@pytest.mark.parametrize("mmt,bn,mb", [(0.0, True, 1024), (0.9, True, 1024), (0.9, False, 1024), (0.0, False, 256), (0.9, False, 256)])
def test_weight_planes_kept_from_step_to_step(aslp, dev, mmt, bn, mb):
    """The weight-gradient product's epilogue writes the updated weights' planes (under a bound of |W - lr dW| known before the launch) and
    the next step's forward and in-diff products read them: six steps agree with six steps that convert the weights anew every time, and
    with the fp32 instruction, like two fp32 summation orders do.  Without BatchNormalization the sigmoid layers' planes come from the
    forward epilogue and from the Sigmoid's backward pass (scale from the maxima the in-diff product leaves); at minibatch 256 the layer
    products split K over workgroups and their second launch writes planes and maxima."""
    runs = {}
    try:
        for name, split, keep in (("fp32", 0, 0), ("convert", 1, 0), ("kept", 1, 1)):
            aslp.lib.aslp_gemm_split16(split)
            aslp.lib.aslp_keep_weight_planes(keep)
            net = aslp.Nnet.Init(_bn_dnn_proto(bn=bn), seed=3)
            net.SetTrainOptions(learn_rate=2e-3, momentum=mmt)
            xe = aslp.Xent()
            g = torch.Generator(device="cpu").manual_seed(5)
            for step in range(6):
                x = torch.randn(mb, 440, generator=g).to(dev)
                lab = torch.randint(0, 3000, (mb,), generator=g, dtype=torch.int32).to(dev)
                net.TrainStepXent(xe, x, lab)
            st = xe.GetStats()
            runs[name] = ((st["loss"] - st["entropy"]) / st["frames"], np.asarray(net.GetParams(), np.float32))
    finally:
        aslp.lib.aslp_gemm_split16(-1)
        aslp.lib.aslp_keep_weight_planes(-1)
    for a, b in (("kept", "convert"), ("kept", "fp32")):
        assert abs(runs[a][0] - runs[b][0]) < 2e-6 * abs(runs[b][0]), (a, b, runs[a][0], runs[b][0])
        assert np.linalg.norm(runs[a][1] - runs[b][1]) < 2e-6 * np.linalg.norm(runs[b][1]), (a, b)


@pytest.mark.parametrize("announce", [False, True])
def test_weights_written_through_the_raw_pointers(aslp, dev, announce):
    """GetGpuParams hands the weights to outside writers (model averaging).  Silent writers: the net stops keeping planes of its weights.
    Announcing writers (aslp_params_changed after every write, what the native sync workers do): the kept planes are dropped at the
    announcement.  Either way the next forward pass sees the written weights: with every weight matrix zeroed the posteriors no longer
    depend on the input."""
    aslp.lib.aslp_gemm_split16(1)
    aslp.lib.aslp_keep_weight_planes(1)
    try:
        net = aslp.Nnet.Init(_bn_dnn_proto(layers=2), seed=9)
        net.SetTrainOptions(learn_rate=1e-3, momentum=0.0)
        xe = aslp.Xent()
        g = torch.Generator(device="cpu").manual_seed(6)
        x = torch.randn(1024, 440, generator=g).to(dev)
        lab = torch.randint(0, 3000, (1024,), generator=g, dtype=torch.int32).to(dev)
        params = net.GetGpuParams(writers_announce=announce)
        for _ in range(3):
            net.TrainStepXent(xe, x, lab)     # (with announcing writers the planes of the weights are kept across these steps)
        torch.cuda.synchronize()
        from parallel_model import alias_device_params
        for t in alias_device_params(params):
            if t.numel() > 4096:    # the weight matrices (biases and BatchNormalization vectors are shorter)
                t.zero_()
        torch.cuda.synchronize()
        if announce:
            aslp.lib.aslp_params_changed()
        post = net.Propagate(x)
        assert (post - post[0]).abs().max().item() == 0.0      # W = 0 in the forward pass: every row is softmax(bias)
    finally:
        aslp.lib.aslp_gemm_split16(-1)
        aslp.lib.aslp_keep_weight_planes(-1)


@pytest.mark.parametrize("M,N,K", [(2048, 2048, 2048), (1920, 2048, 512), (4096, 4096, 1024), (2000, 2176, 1000)])
def test_producer_consumer_kernel_forms_the_same_bits(aslp, dev, M, N, K):
    """gemm_s16_pc (tile 351: four consumer waves multiply, four producer waves issue the LDS-DMA requests; the default wherever the
    128 x 128 tile is chosen) against the one-role kernels: same instruction order per accumulator, so the results are bit-identical --
    to the 128 x 128 one-role kernel (312) and to the 64 x 128 one (308) -- whole tiles, a ragged last tile row / column, K not a multiple
    of the K tile; repeated launches agree bit for bit (the producers' surplus requests never reach the epilogue's LDS)."""
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev, generator=g)
    B = torch.randn(N, K, device=dev, generator=g)
    A[5] *= 1e-5
    outs = {}
    aslp.lib.aslp_gemm_split16(1)
    for tile in (351, 312, 308):
        aslp.lib.aslp_gemm_split16_tile(tile)
        C = torch.full((M, N), 7.0, device=dev)
        aslp.ops.sgemm(0, 1, 1.0, A, B, 0.0, C)
        assert aslp.lib.aslp_gemm_last_tile() == (311 if tile == 312 else tile)
        outs[tile] = C
    assert torch.equal(outs[351], outs[312]) and torch.equal(outs[351], outs[308])
    assert err_vs_double(outs[351], A, B, 0, 1) < 2.5e-7
    aslp.lib.aslp_gemm_split16_tile(351)
    for _ in range(5):
        C = torch.empty(M, N, device=dev)
        aslp.ops.sgemm(0, 1, 1.0, A, B, 0.0, C)
        assert torch.equal(C, outs[351])
    # with beta and a bias (the plain epilogue's other inputs)
    bias = torch.randn(N, device=dev, generator=g)
    res = []
    for tile in (351, 312):
        aslp.lib.aslp_gemm_split16_tile(tile)
        C = outs[308].clone()
        aslp.ops.sgemm(0, 1, 0.5, A, B, 0.25, C, aslp._lib.GemmEpilogue(bias.data_ptr()))
        res.append(C)
    assert torch.equal(res[0], res[1])
    # the heuristic picks it by itself on a grid of >= 224 tiles of 128 x 128
    aslp.lib.aslp_gemm_split16_tile(-1)
    if M * N >= 224 * 128 * 128 and M % 128 == 0 and N % 128 == 0:
        C = torch.empty(M, N, device=dev)
        aslp.ops.sgemm(0, 1, 1.0, A, B, 0.0, C)
        assert aslp.lib.aslp_gemm_last_tile() in (351, 308)
