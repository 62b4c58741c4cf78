"""The front-end components of the CNN / cFSMN recipes on the GPU (nnet/nnet-conv.h, csrc/conv_pool.hip): LinearTransform,
ConvolutionalComponent, MaxPoolingComponent, LengthNormComponent, Pnorm -- driven through the C ABI (aslp_nnet_*), against
  (1) the same op sequences issued on the REFERENCE's own CuMatrix library (tests/golden/component_ops.bin), no oracle in between,
  (2) the known answers of the reference's own unit test (nnet-component-test.cc:53-206, tests/golden/component_known_answers.json),
  (3) the oracle at the recipes' shapes (run_cnn.sh:60-80, run_cfsmn.sh:60-95): aslp-nnet-init's Nnet::Init accepts the protos as they
      are, and two training steps agree with the oracle chain -- applied gradients compared, not only parameters.
Tolerances: bit-exact for index / mask work, 2e-5 of max(1, |ref|) behind products (the bar is 1e-4)."""
import re

import numpy as np
import pytest
import torch

import cumatrix_golden
import nnet_io
from test_oracle_components_cpu import parse_component

pytestmark = pytest.mark.gpu


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol


def T(a, dev, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)


def text_mat(m):
    m = np.atleast_2d(np.asarray(m))
    return "[ " + "\n".join(" ".join(repr(float(v)) for v in row) for row in m) + " ]"


def text_vec(v):
    return "[ " + " ".join(repr(float(x)) for x in np.asarray(v).ravel()) + " ]"


def write_text_nnet(path, marker, din, dout, payload):
    """one component between the graph endpoints, in the TEXT form of the nnet format (nnet-component.cc:288-342)"""
    with open(path, "w") as f:
        f.write("<Nnet>\n<InputLayer> %d %d 0 [ -1 ] [ 0 ]\n" % (din, din))
        f.write("%s %d %d 1 [ 0 ] [ 0 ]\n%s\n" % (marker, dout, din, payload))
        f.write("<OutputLayer> %d %d 2 [ 1 ] [ 0 ]\n</Nnet>\n" % (dout, dout))


def conv_payload(filters, bias, pd, ps, pst, coef=1.0, bcoef=1.0, max_norm=0.0):
    return ("<PatchDim> %d <PatchStep> %d <PatchStride> %d <LearnRateCoef> %r <BiasLearnRateCoef> %r <MaxNorm> %r <Filters> %s <Bias> %s"
            % (pd, ps, pst, coef, bcoef, max_norm, text_mat(filters), text_vec(bias)))


def test_kernel_abi_ops_match_reference_library(aslp, dev):
    """cudaF_max / equal_element_mask / group_pnorm / calc_pnorm_deriv / group_max / calc_group_max_deriv / mul_rows_group_mat"""
    from kaldi_aslp_amd._lib import D3, check_error
    g = cumatrix_golden.load_components()
    ops, lib = aslp.ops, aslp.lib
    a, b = T(g["op_max_a"], dev), T(g["op_max_b"], dev)
    mask = torch.empty_like(a)
    lib.cudaF_equal_element_mask(D3, D3, ops.ptr(a), ops.ptr(b), ops.ptr(mask), ops.dim(a), ops.dim(b).stride, ops.dim(mask).stride)
    lib.cudaF_max(D3, D3, ops.ptr(a), ops.ptr(b), ops.dim(a), ops.dim(b).stride)
    check_error()
    assert np.array_equal(mask.cpu().numpy(), g["op_eqmask"]) and np.array_equal(a.cpu().numpy(), g["op_max_out"])
    x, od = T(g["grp_in"], dev), T(g["grp_od"], dev)
    group = x.shape[1] // od.shape[1]
    for i, p in enumerate((2.0, 1.0, 3.0)):
        y, d = torch.empty_like(od), torch.empty_like(x)
        lib.cudaF_group_pnorm(D3, D3, ops.ptr(y), ops.ptr(x), ops.dim(y), ops.dim(x).stride, group, C_float(p))
        lib.cudaF_calc_pnorm_deriv(D3, D3, ops.ptr(d), ops.ptr(x), ops.ptr(y), ops.dim(d), ops.dim(y).stride, group, C_float(p))
        check_error()
        assert close(y.cpu().numpy(), g["pnorm%d_out" % i], 2e-6) and close(d.cpu().numpy(), g["pnorm%d_deriv" % i], 5e-6), p
        lib.cudaF_mul_rows_group_mat(D3, D3, ops.ptr(d), ops.ptr(od), ops.dim(d), ops.dim(od).stride, group)
        assert close(d.cpu().numpy(), g["pnorm%d_id" % i], 5e-6), p
    y, d = torch.empty_like(od), torch.empty_like(x)
    lib.cudaF_group_max(D3, D3, ops.ptr(y), ops.ptr(x), ops.dim(y), ops.dim(x).stride, group)
    lib.cudaF_calc_group_max_deriv(D3, D3, ops.ptr(d), ops.ptr(x), ops.ptr(y), ops.dim(d), ops.dim(y).stride, group)
    check_error()
    assert np.array_equal(y.cpu().numpy(), g["gmax_out"]) and np.array_equal(d.cpu().numpy(), g["gmax_deriv"])


def C_float(v):
    import ctypes
    return ctypes.c_float(v)


def test_linear_transform_matches_reference_library(aslp, dev, tmp_path):
    """nnet-linear-transform.h:127-160, two minibatches with momentum 0.9, l2, l1 and <LearnRateCoef> 0.7"""
    g = cumatrix_golden.load_components()
    lr, mmt, l2, l1, coef = [float(v) for v in g["lin_opts"]]
    W = g["lin_W0"]
    path = tmp_path / "lin.nnet"
    write_text_nnet(path, "<LinearTransform>", W.shape[1], W.shape[0], "<LearnRateCoef> %r %s" % (coef, text_mat(W)))
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt, l2_penalty=l2, l1_penalty=l1)
    for step in (0, 1):
        assert close(net.Propagate(T(g["lin_in%d" % step], dev)).cpu().numpy(), g["lin_out%d" % step], 2e-6)
        assert close(net.Backpropagate(T(g["lin_od%d" % step], dev), want_in_diff=True).cpu().numpy(), g["lin_id%d" % step], 2e-6)
        assert close(net.GetParams(), g["lin_W%d" % (step + 1)].ravel(), 2e-6), step


def test_convolutional_component_matches_reference_library(aslp, dev, tmp_path):
    """nnet-convolutional-component.h:268-470: 4 overlapping patches, gradient summed over patches, max-norm; two minibatches.  The
    applied gradient (W_before - W_after) / lr is compared where the max-norm does not rescale the row."""
    g = cumatrix_golden.load_components()
    in_dim, F, pd, ps, pst, lr, coef, bcoef, max_norm = [float(v) for v in g["conv_geom"]]
    filt, bias = g["conv_filters0"], g["conv_bias0"]
    P = 1 + (int(pst) - int(pd)) // int(ps)
    path = tmp_path / "conv.nnet"
    write_text_nnet(path, "<ConvolutionalComponent>", int(in_dim), int(F) * P, conv_payload(filt, bias, int(pd), int(ps), int(pst), coef, bcoef, max_norm))
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr)
    for step in (0, 1):
        assert close(net.Propagate(T(g["conv_in%d" % step], dev)).cpu().numpy(), g["conv_out%d" % step], 5e-6)
        assert close(net.Backpropagate(T(g["conv_od%d" % step], dev), want_in_diff=True).cpu().numpy(), g["conv_id%d" % step], 5e-6)
        ref = np.concatenate([g["conv_filters%d" % (step + 1)].ravel(), g["conv_bias%d" % (step + 1)]])
        assert close(net.GetParams(), ref, 5e-6), step
        got_bias = net.GetParams()[filt.size:]
        assert close((g["conv_bias%d" % step] - got_bias) / (lr * bcoef), g["conv_bgrad%d" % step], 2e-5), step   # the bias gradient as applied


def test_max_pooling_and_length_norm_and_pnorm_match_reference_library(aslp, dev):
    g = cumatrix_golden.load_components()
    in_dim, size, step, stride = [int(v) for v in g["pool_geom"]]
    n_pools = 1 + (in_dim // stride - size) // step
    net = aslp.Nnet.Init("<NnetProto>\n<MaxPoolingComponent> <InputDim> %d <OutputDim> %d <PoolSize> %d <PoolStep> %d <PoolStride> %d\n</NnetProto>\n"
                         % (in_dim, n_pools * stride, size, step, stride))
    assert np.array_equal(net.Propagate(T(g["pool_in"], dev)).cpu().numpy(), g["pool_out"])
    assert np.array_equal(net.Backpropagate(T(g["pool_od"], dev), want_in_diff=True).cpu().numpy(), g["pool_id"])   # overlapping pools, ties: bit-exact
    for w in (0, 1):
        D = g["ln%d_in" % w].shape[1]
        net = aslp.Nnet.Init("<NnetProto>\n<LengthNormComponent> <InputDim> %d <OutputDim> %d\n</NnetProto>\n" % (D, D))
        assert close(net.Propagate(T(g["ln%d_in" % w], dev)).cpu().numpy(), g["ln%d_out" % w], 1e-6)
        assert close(net.Backpropagate(T(g["ln%d_od" % w], dev), want_in_diff=True).cpu().numpy(), g["ln%d_id" % w], 1e-6)
    x, od = g["grp_in"], g["grp_od"]
    for i, p in enumerate((2.0, 1.0, 3.0)):
        net = aslp.Nnet.Init("<NnetProto>\n<Pnorm> <InputDim> %d <OutputDim> %d <P> %r\n</NnetProto>\n" % (x.shape[1], od.shape[1], p))
        assert close(net.Propagate(T(x, dev)).cpu().numpy(), g["pnorm%d_out" % i], 2e-6), p
        assert close(net.Backpropagate(T(od, dev), want_in_diff=True).cpu().numpy(), g["pnorm%d_id" % i], 5e-6), p
    # "<Maxout>" reads as a Pnorm in the reference (nnet-component.cc:79-80): P defaults to 2
    net = aslp.Nnet.Init("<NnetProto>\n<Maxout> <InputDim> %d <OutputDim> %d\n</NnetProto>\n" % (x.shape[1], od.shape[1]))
    assert close(net.Propagate(T(x, dev)).cpu().numpy(), g["pnorm0_out"], 2e-6)


def test_reference_known_answers(aslp, dev, tmp_path):
    """src/aslp-nnet/nnet-component-test.cc:53-206 on the GPU components.  (That test's component strings predate the ASLP file format --
    Component::Read now expects `id [inputs] [offsets]` behind the dimensions, nnet-component.cc:303-315 --, so the fields are re-issued
    in today's format; the numbers are the reference's.)"""
    ka = cumatrix_golden.load_known_answers()
    net = aslp.Nnet.Init("<NnetProto>\n<LengthNormComponent> <InputDim> 5 <OutputDim> 5\n</NnetProto>\n")
    out = net.Propagate(T(ka["UnitTestLengthNorm"]["matrices"]["mat_in"], dev)).cpu().numpy().astype(np.float64)
    assert np.allclose(np.sqrt((out ** 2).sum(1)), 1.0, atol=1e-6)
    for name in ("UnitTestConvolutionalComponentUnity", "UnitTestConvolutionalComponent3x3"):
        t = ka[name]
        marker, dout, din, f = parse_component(t["component"])
        path = tmp_path / (name + ".nnet")
        write_text_nnet(path, marker, din, dout, conv_payload(f["<Filters>"], f["<Bias>"], int(f["<PatchDim>"]), int(f["<PatchStep>"]), int(f["<PatchStride>"]),
                                                              f["<LearnRateCoef>"], f["<BiasLearnRateCoef>"], f["<MaxNorm>"]))
        net = aslp.Nnet.Read(path)
        m = t["matrices"]
        out_ref = m.get("mat_out_ref", m["mat_in"])
        od = m.get("mat_out_diff", m["mat_in"])
        id_ref = m.get("mat_in_diff_ref", od)
        assert np.array_equal(net.Propagate(T(m["mat_in"], dev)).cpu().numpy(), out_ref), name
        net.SetTrainOptions(learn_rate=0.0)
        assert np.array_equal(net.Backpropagate(T(od, dev), want_in_diff=True).cpu().numpy(), id_ref), name
    t = ka["UnitTestMaxPoolingComponent"]
    net = aslp.Nnet.Init("<NnetProto>\n%s\n</NnetProto>\n" % t["component"])
    m = t["matrices"]
    out = net.Propagate(T(m["mat_in"], dev))
    assert np.array_equal(out.cpu().numpy(), m["mat_out_ref"])
    assert np.array_equal(net.Backpropagate(torch.ones_like(out), want_in_diff=True).cpu().numpy(), m["mat_in_diff_ref"])


CNN_PROTO = """<NnetProto>
<ConvolutionalComponent> <InputDim> 440 <OutputDim> 4096 <PatchDim> 9 <PatchStep> 1 <PatchStride> 40 <BiasMean> -2.000000 <BiasRange> 4.000000 <ParamStddev> 0.1 <MaxNorm> 30
<MaxPoolingComponent> <InputDim> 4096 <OutputDim> 1024 <PoolSize> 4 <PoolStep> 4 <PoolStride> 128
<Sigmoid> <InputDim> 1024 <OutputDim> 1024
<AffineTransform> <InputDim> 1024 <OutputDim> 300 <BiasMean> -2.000000 <BiasRange> 4.000000 <ParamStddev> 0.1
<Softmax> <InputDim> 300 <OutputDim> 300
</NnetProto>
"""


def test_run_cnn_proto_initialises_and_trains_like_the_oracle(aslp, oracle, dev):
    """aslp_scripts/aslp_nnet/run_cnn.sh:60-80 (num_tgt 300 here), minibatch 256, learn rate 0.008: two steps of Propagate + Xent +
    Backpropagate + Update against the oracle's Conv -> MaxPool -> Sigmoid -> Affine -> Softmax chain on the same weights."""
    import ctypes as C
    net = aslp.Nnet.Init(CNN_PROTO, seed=3)
    assert [net.Marker(c) for c in range(net.NumComponents())] == ["<InputLayer>", "<ConvolutionalComponent>", "<MaxPoolingComponent>", "<Sigmoid>",
                                                                    "<AffineTransform>", "<Softmax>", "<OutputLayer>"]
    p = net.GetParams()
    F, K, H, A = 128, 99, 1024, 300
    filt, cb = p[:F * K].reshape(F, K).copy(), p[F * K:F * K + F].copy()
    W, b = p[F * K + F:F * K + F + A * H].reshape(A, H).copy(), p[F * K + F + A * H:].copy()
    conv = oracle.Conv(filt, cb, 440, 9, 1, 40)
    lr, mb = 0.008, 256
    net.SetTrainOptions(learn_rate=lr)
    xent = aslp.Xent()
    rng = np.random.default_rng(11)
    Wc, bc = np.zeros_like(W), np.zeros_like(b)
    opts = oracle.AffineOpts(lr, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0)
    for step in range(2):
        x = rng.standard_normal((mb, 440)).astype(np.float32)
        lab = rng.integers(0, A, mb).astype(np.int32)
        # oracle chain
        c_out = conv.propagate(x)
        pool = oracle.max_pool(c_out, 4, 4, 128)
        h = oracle.unary("orc_sigmoid", pool)
        logits = np.empty((mb, A), np.float32)
        oracle.lib.orc_affine_propagate(logits, A, h, H, mb, W, H, b, H, A)
        y = oracle.unary("orc_softmax_rows", logits)
        tgt = np.zeros((mb, A), np.float32); tgt[np.arange(mb), lab] = 1.0
        diff = y - tgt
        d_h = np.empty((mb, H), np.float32)
        oracle.lib.orc_affine_backpropagate(d_h, H, diff, A, mb, W, H, H, A)
        W_before, filt_before, cb_before = W.copy(), conv.filters.copy(), conv.bias.copy()
        oracle.lib.orc_affine_update(W, H, b, Wc, H, bc, h, H, diff, A, mb, H, A, C.byref(opts))
        d_pool = oracle.binary("orc_diff_sigmoid", h, d_h)
        d_conv = oracle.max_pool_backprop(c_out, pool, d_pool, 4, 4, 128)
        conv.update(d_conv, lr, 1.0, 1.0, 30.0)
        # engine
        before = net.GetParams()
        net.TrainStepXent(xent, T(x, dev), T(lab, dev, torch.int32))
        after = net.GetParams()
        assert close(net.ComponentOutput(1, mb, 4096), c_out, 2e-5) and close(net.ComponentOutput(2, mb, 1024), pool, 2e-5)
        assert close(net.ComponentOutput(net.NumComponents() - 1, mb, A), y, 2e-5)
        # applied gradients: (before - after) / lr per tensor against the oracle's
        g_eng = (before - after) / lr
        g_ref = np.concatenate([((filt_before - conv.filters) / lr).ravel(), (cb_before - conv.bias) / lr, ((W_before - W) / lr).ravel(), bc])
        for lo, hi, nm in ((0, F * K, "filters"), (F * K, F * K + F, "conv bias"), (F * K + F, F * K + F + A * H, "W"), (F * K + F + A * H, p.size, "b")):
            e, r = g_eng[lo:hi].astype(np.float64), g_ref[lo:hi].astype(np.float64)
            assert np.linalg.norm(e - r) / np.linalg.norm(r) < 1e-4, (nm, step)
            assert np.max(np.abs(e - r)) / max(1.0, np.max(np.abs(r))) < 1e-3, (nm, step)   # (g / lr amplifies the fp32 rounding of W by 1 / lr)
        assert close(after, np.concatenate([conv.filters.ravel(), conv.bias, W.ravel(), b]), 2e-5), step


CFSMN_PROTO = """<NnetProto>
<AffineTransform> <InputDim> 120 <OutputDim> 1024 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<ReLU> <InputDim> 1024 <OutputDim> 1024
<AffineTransform> <InputDim> 1024 <OutputDim> 512 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<CompactFsmn> <InputDim> 512 <OutputDim> 512 <PastContext> 30 <FutureContext> 30 <LearnRateCoef> 1.0 
<AffineTransform> <InputDim> 512 <OutputDim> 1024 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<Sigmoid> <InputDim> 1024 <OutputDim> 1024
<AffineTransform> <InputDim> 1024 <OutputDim> 512 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<CompactFsmn> <InputDim> 512 <OutputDim> 512 <PastContext> 30 <FutureContext> 30 <LearnRateCoef> 1.0 
<AffineTransform> <InputDim> 512 <OutputDim> 1024 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<ReLU> <InputDim> 1024 <OutputDim> 1024
<AffineTransform> <InputDim> 1024 <OutputDim> 512 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<CompactFsmn> <InputDim> 512 <OutputDim> 512 <PastContext> 30 <FutureContext> 30 <LearnRateCoef> 1.0 
<AffineTransform> <InputDim> 512 <OutputDim> 1024 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<Sigmoid> <InputDim> 1024 <OutputDim> 1024
<AffineTransform> <InputDim> 1024 <OutputDim> 1024 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<ReLU> <InputDim> 1024 <OutputDim> 1024
<AffineTransform> <InputDim> 1024 <OutputDim> 1024 <BiasMean> 0 <BiasRange> 1 <ParamStddev> 0.1
<Sigmoid> <InputDim> 1024 <OutputDim> 1024
<LinearTransform> <InputDim> 1024 <OutputDim> 512 <ParamStddev> 0.1
<AffineTransform> <InputDim> 512 <OutputDim> 200 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.040000
<Softmax> <InputDim> 200 <OutputDim> 200
</NnetProto>
"""


def test_run_cfsmn_proto_initialises_and_its_linear_transform_trains_like_the_oracle(aslp, oracle, dev, tmp_path):
    """aslp_scripts/aslp_nnet/run_cfsmn.sh:60-95 (120-dim input, 200 targets): aslp-nnet-init's path takes the proto as it is, the model
    survives a write / read round trip, and over two per-utterance steps the <LinearTransform> (component 19) computes the oracle's
    output, in-diff and applied gradient from the tensors that reach it."""
    net = aslp.Nnet.Init(CFSMN_PROTO, seed=5)
    markers = [net.Marker(c) for c in range(net.NumComponents())]
    assert markers.count("<CompactFsmn>") == 3 and markers[19] == "<LinearTransform>" and markers[-2] == "<Softmax>"
    path = tmp_path / "cfsmn.nnet"
    net.Write(path)
    again = aslp.Nnet.Read(path)
    assert np.array_equal(net.GetParams(), again.GetParams())
    # where the LinearTransform's 1024 x 512 matrix sits in the flat parameter vector: walk the proto
    off = 0
    for line in CFSMN_PROTO.splitlines()[1:20]:
        din, dout = [int(v) for v in re.findall(r"<(?:Input|Output)Dim> (\d+)", line)]
        if line.startswith("<AffineTransform>"): off += din * dout + dout
        elif line.startswith("<CompactFsmn>"): off += 61 * din
    W0 = net.GetParams()[off:off + 512 * 1024].reshape(512, 1024).copy()
    lin = oracle.Linear(W0)
    lr = 0.01   # (the recipe's 0.008 / 1024 would bury the applied gradient (before - after) / lr under the fp32 rounding of the weights)
    net.SetTrainOptions(learn_rate=lr, momentum=0.9)
    xent = aslp.Xent()
    rng = np.random.default_rng(2)
    for step in range(2):
        Tn = 300 + 50 * step
        x = rng.standard_normal((Tn, 120)).astype(np.float32)
        lab = rng.integers(0, 200, Tn).astype(np.int32)
        before = net.GetParams()[off:off + W0.size]
        net.TrainStepXent(xent, T(x, dev), T(lab, dev, torch.int32))
        after = net.GetParams()[off:off + W0.size]
        h = net.ComponentOutput(18, Tn, 1024)
        od = net.ComponentOutDiff(19, Tn, 512)
        assert close(net.ComponentOutput(19, Tn, 512), lin.propagate(h), 2e-5)
        assert close(net.ComponentOutDiff(18, Tn, 1024), lin.backpropagate(od), 2e-5)
        Wb = lin.W.copy()
        lin.update(h, od, lr, 0.9)
        e, r = ((before - after) / lr).astype(np.float64), ((Wb - lin.W) / lr).ravel().astype(np.float64)
        assert np.linalg.norm(e - r) / np.linalg.norm(r) < 1e-4 and np.max(np.abs(e - r)) / max(1.0, np.max(np.abs(r))) < 1e-2, step
        assert close(after, lin.W.ravel(), 2e-6), step


def _aff(i, o, extra=""):
    return "<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> -2.000000 <BiasRange> 4.000000 <ParamStddev> 0.1%s" % (i, o, extra)


def _act(kind, d):
    return "<%s> <InputDim> %d <OutputDim> %d" % (kind, d, d)


def _cnn_front(num_feat):
    return ["<ConvolutionalComponent> <InputDim> %d <OutputDim> 4096 <PatchDim> 9 <PatchStep> 1 <PatchStride> 40 <BiasMean> -2.000000 <BiasRange> 4.000000 "
            "<ParamStddev> 0.1 <MaxNorm> 30" % num_feat,
            "<MaxPoolingComponent> <InputDim> 4096 <OutputDim> 1024 <PoolSize> 4 <PoolStep> 4 <PoolStride> 128"]


def recipe_protos(num_feat=440, num_tgt=120):
    """The network descriptions of the recipes in aslp_scripts/aslp_nnet that round 2's aslp-nnet-init refused, rebuilt from their layer lists
    (run_cnn.sh:69-77 + its hidden.conf :79-84, run_cnn_1dnn_2lstm.sh, run_ctc_cnn_1dnn_2blstm.sh:64-82, run_cfsmn.sh:68-93, run_cfsmn_pre.sh)."""
    out_layer = ["<AffineTransform> <InputDim> 512 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.040000" % num_tgt, _act("Softmax", num_tgt)]
    lstm = "<%s> <InputDim> 512 <OutputDim> 512 <CellDim> 1024 <ParamScale> 0.010000 <ClipGradient> 5.000000"
    p = {}
    p["run_cnn"] = _cnn_front(num_feat) + [_act("Sigmoid", 1024), _aff(1024, num_tgt), _act("Softmax", num_tgt)]
    p["run_cnn.hidden"] = [_aff(1024, 1024), _act("Sigmoid", 1024)]
    p["run_cnn_1dnn_2lstm"] = ([_cnn_front(num_feat)[0], _act("BatchNormalization", 4096), _cnn_front(num_feat)[1], _act("Sigmoid", 1024), _aff(1024, 512),
                                _act("BatchNormalization", 512), _act("Sigmoid", 512)] +
                               [lstm % "LstmProjectedStreams", _act("BatchNormalization", 512)] * 2 + out_layer)
    p["run_ctc_cnn_1dnn_2blstm"] = (_cnn_front(num_feat) + [_act("BatchNormalization", 1024), _act("Sigmoid", 1024), _aff(1024, 1024),
                                    _act("BatchNormalization", 1024), _act("Sigmoid", 1024), _aff(1024, 512), _act("BatchNormalization", 512), _act("Sigmoid", 512)] +
                                    [lstm % "BLstmProjectedStreams", _act("BatchNormalization", 512)] * 2 + out_layer)
    fsmn = "<CompactFsmn> <InputDim> 512 <OutputDim> 512 <PastContext> 30 <FutureContext> 30 <LearnRateCoef> 1.0 "
    p["run_cfsmn_pre.block"] = [_aff(1024, 512, " <MaxNorm> 20"), fsmn + " <ClipGradient> 10", _aff(512, 1024, " <MaxNorm> 20"), _act("ReLU", 1024)]
    p["run_cfsmn_pre.init"] = [_aff(num_feat, 1024, " <MaxNorm> 20"), _act("ReLU", 1024), "<LinearTransform> <InputDim> 1024 <OutputDim> 512 <ParamStddev> 0.1",
                               "<AffineTransform> <InputDim> 512 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.040000 <MaxNorm> 20" % num_tgt,
                               _act("Softmax", num_tgt)]
    return {k: "<NnetProto>\n" + "\n".join(v) + "\n</NnetProto>\n" for k, v in p.items()}


def test_aslp_nnet_init_accepts_the_cnn_and_cfsmn_recipes(aslp, dev, tmp_path):
    """The tool itself (bin/aslp-nnet-init, then aslp-nnet-info and a forward pass through the written model): the B7 row's "recipes drop in
    unchanged" for the recipes whose first stage died in round 2."""
    from test_tools_gpu import tool
    protos = dict(recipe_protos(), run_cfsmn=CFSMN_PROTO)
    for name, proto in protos.items():
        (tmp_path / (name + ".proto")).write_text(proto)
        out = tmp_path / (name + ".nnet")
        tool("aslp-nnet-init", "--binary=true", "--seed=777", str(tmp_path / (name + ".proto")), str(out))
        info = tool("aslp-nnet-info", str(out)).stdout.decode()
        for marker in set(re.findall(r"^<(\w+)>", proto, re.M)) - {"NnetProto"}:
            assert "<%s>" % marker in info, (name, marker)
        net = aslp.Nnet.Read(out)
        rows = 64
        if "Lstm" in proto:
            net.SetSeqLengths([16] * 4) if "BLstm" in proto else net.ResetLstmStreams([1] * 4)
        y = net.Propagate(torch.randn(rows, net.InputDim(), device=dev))
        assert y.shape == (rows, net.OutputDim()) and bool(torch.isfinite(y).all()), name
    # the reference's own refusal stays a refusal: <ClipGradient> is no AffineTransform option (nnet-affine-transform.h:70-83), so
    # run_eesen_ctc_cnn_1dnn_2blstm.sh:71 fails in the reference's aslp-nnet-init as well
    (tmp_path / "bad.proto").write_text("<NnetProto>\n" + _aff(64, 64, " <ClipGradient> 5.000000") + "\n</NnetProto>\n")
    p = tool("aslp-nnet-init", str(tmp_path / "bad.proto"), str(tmp_path / "bad.nnet"), ok=False)
    assert p.returncode != 0 and b"Unknown token <ClipGradient>" in p.stderr
