"""GPU parity of the host engine (Nnet / Component / Xent driven through the C ABI of
include/aslp_nnet.h) against the CPU oracle, for the DNN path of BASELINE.json configs 1-2:
Propagate outputs, per-layer buffers, loss statistics, updated weights after 1 and 2 SGD steps
(the second step exposes the momentum path), at small sizes; plus size-independent properties."""
import numpy as np
import pytest
import torch

import nnet_io

pytestmark = pytest.mark.gpu
TOL = 1e-4


def make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, bn, mb, seed):
    d = oracle.lib.orc_dnn_create(in_dim, hid, nh, out_dim, bn, mb, seed)
    L = oracle.lib.orc_dnn_num_layers(d)
    layers = []
    import ctypes as C
    for l in range(L):
        r, c = C.c_int(), C.c_int()
        wp = oracle.lib.orc_dnn_weight(d, l, C.byref(r), C.byref(c))
        W = np.ctypeslib.as_array(wp, shape=(r.value, c.value)).copy()
        b = np.ctypeslib.as_array(oracle.lib.orc_dnn_bias(d, l), shape=(r.value,)).copy()
        layers.append(("<AffineTransform>", c.value, r.value, nnet_io.affine(W, b)))
        if l < L - 1:
            if bn:
                layers.append(("<BatchNormalization>", r.value, r.value, nnet_io.batchnorm(np.zeros(r.value), np.ones(r.value))))
            layers.append(("<Sigmoid>", r.value, r.value, b""))
    layers.append(("<Softmax>", out_dim, out_dim, b""))
    path = tmp_path / "dnn.nnet"
    nnet_io.write_simple_nnet(path, layers)
    return d, path


def oracle_params(oracle, d, bn):
    import ctypes as C
    L = oracle.lib.orc_dnn_num_layers(d)
    out = []
    for l in range(L):
        r, c = C.c_int(), C.c_int()
        wp = oracle.lib.orc_dnn_weight(d, l, C.byref(r), C.byref(c))
        out.append(np.ctypeslib.as_array(wp, shape=(r.value * c.value,)).copy())
        out.append(np.ctypeslib.as_array(oracle.lib.orc_dnn_bias(d, l), shape=(r.value,)).copy())
        if bn and l < L - 1:  # GetParams order of BatchNormalization: shift then scale
            out.append(np.ctypeslib.as_array(oracle.lib.orc_dnn_bn_shift(d, l), shape=(r.value,)).copy())
            out.append(np.ctypeslib.as_array(oracle.lib.orc_dnn_bn_scale(d, l), shape=(r.value,)).copy())
    return np.concatenate(out)


@pytest.mark.parametrize("bn", [0, 1])
# (440, 2048, 5, 3000, 256): BASELINE cfg1 / the per-GPU leg of cfg4 at FULL size against the oracle chain (0.1 s per oracle step)
@pytest.mark.parametrize("dims", [(24, 32, 2, 10, 16), (44, 64, 3, 50, 128), (40, 96, 5, 300, 256), (440, 2048, 5, 3000, 256)])
@pytest.mark.parametrize("momentum", [0.0, 0.9])
def test_dnn_train_steps_match_oracle(aslp, oracle, dev, tmp_path, dims, bn, momentum):
    _train_steps_vs_oracle(aslp, oracle, dev, tmp_path, dims, bn, momentum, lr=0.002)


@pytest.mark.parametrize("momentum", [0.0, 0.9])
def test_cfg2_full_size_matches_oracle(aslp, oracle, dev, tmp_path, momentum):
    """BASELINE cfg2 itself -- 440 -> 5 x 2048 + BatchNormalization + Sigmoid -> 3000, minibatch 1024, learn rate 0.008 (run_bn_dnn.sh:64-101)
    -- on the default path (split-fp16 products from producer-made planes, weights' planes kept from step to step, updates on the side
    stream) against the oracle chain (nnet-batch-normalization.h:177-284, nnet-affine-transform.h:186-245): outputs, every parameter, the
    gradient each tensor was moved by, largest element.  ~0.4 s per oracle step."""
    _train_steps_vs_oracle(aslp, oracle, dev, tmp_path, (440, 2048, 5, 3000, 1024), 1, momentum, lr=0.008, steps=3)
    assert aslp.lib.aslp_gemm_last_tile() in (304, 308, 311, 328)     # the products ran on the split-fp16 kernels


def _train_steps_vs_oracle(aslp, oracle, dev, tmp_path, dims, bn, momentum, lr, steps=2):
    in_dim, hid, nh, out_dim, mb = dims
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, bn, mb, seed=5)
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=momentum)
    assert net.NumParams() == oracle_params(oracle, d, bn).size
    assert oracle.rel_err(net.GetParams(), oracle_params(oracle, d, bn)) == 0.0  # file round trip is exact
    rng = np.random.default_rng(3)
    xent = aslp.Xent()
    tot_loss = 0.0
    prev_e, prev_o = net.GetParams(), oracle_params(oracle, d, bn)
    for step in range(steps):
        x = rng.standard_normal((mb, in_dim)).astype(np.float32)
        lab = rng.integers(0, out_dim, mb).astype(np.int32)
        ref_loss = oracle.lib.orc_dnn_train_step(d, x, lab, lr, momentum)
        tot_loss += ref_loss
        xd, ld = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
        net.TrainStepXent(xent, xd, ld)
        # softmax output of this step (OutputLayer buffer) vs oracle
        out_ref = np.ctypeslib.as_array(oracle.lib.orc_dnn_output(d), shape=(mb, out_dim))
        try:
            out = net.ComponentOutput(net.NumComponents() - 1, mb, out_dim)
        except RuntimeError as e:   # at some shapes the executor leaves the final Softmax to the loss kernel: the posteriors exist only inside it
            assert "not materialised" in str(e)
            out = None              # (the loss below and the gradients are then what pins this step)
        if out is not None:
            assert oracle.rel_err(out, out_ref) < TOL and oracle.max_err(out, out_ref) < 10 * TOL, ("output", step)
        got, want = net.GetParams(), oracle_params(oracle, d, bn)
        assert oracle.rel_err(got, want) < TOL and oracle.max_err(got, want) < 10 * TOL, ("params", step)
        # what was applied in this step, engine vs oracle, both read back as (before - after) / lr: insensitive to |W| >> |lr g|
        g_e, g_o = (prev_e.astype(np.float64) - got) / lr, (prev_o.astype(np.float64) - want) / lr
        floor = 2.0 ** -23 * np.abs(want).max() / lr
        assert np.linalg.norm(g_e - g_o) / np.linalg.norm(g_o) < TOL + 2.0 ** -23 * np.linalg.norm(want) / lr / np.linalg.norm(g_o), ("applied gradient", step)
        assert np.max(np.abs(g_e - g_o)) / max(1.0, np.max(np.abs(g_o))) < 10 * TOL + 2 * floor, ("applied gradient, max element", step)
        prev_e, prev_o = got, want
    st = xent.GetStats()
    assert st["frames"] == steps * mb
    assert abs(st["loss"] - tot_loss) / tot_loss < 1e-5
    assert "AvgLoss:" in xent.Report() and "FRAME_ACCURACY >>" in xent.Report()
    oracle.lib.orc_dnn_destroy(d)


@pytest.mark.parametrize("reg", [dict(l2=1e-3), dict(l1=2e-4), dict(max_norm=0.9), dict(l2=5e-4, l1=1e-4, max_norm=1.1),
                                 dict(l1=2e-4, lr_coef=0.5, bias_lr_coef=2.0), dict(max_norm=0.9, momentum=0.0)])
@pytest.mark.parametrize("dims", [(40, 24, 32), (130, 257, 128)])
def test_affine_update_regularisers_match_oracle(aslp, oracle, dev, tmp_path, dims, reg):
    """AffineTransform::Update beyond the plain SGD step (nnet-affine-transform.h:200-245): l2 weight decay scaled by the
    frame count (:214-216), cu::RegularizeL1 on the weights AND the momentum buffer (:218-220, cu-math.cc:37-75), the
    <MaxNorm> row shrink (:231-243), and the two learn-rate coefficients -- three steps with momentum so the L1 clamp of
    the gradient buffer carries over."""
    in_dim, out_dim, mb = dims
    l2, l1, max_norm = reg.get("l2", 0.0), reg.get("l1", 0.0), reg.get("max_norm", 0.0)
    lr_coef, bias_lr_coef, mmt, lr = reg.get("lr_coef", 1.0), reg.get("bias_lr_coef", 1.0), reg.get("momentum", 0.9), 0.01
    rng = np.random.default_rng(11)
    W = (rng.standard_normal((out_dim, in_dim)) * 0.1).astype(np.float32)
    W[rng.random(W.shape) < 0.2] *= 1e-3          # weights near zero: the L1 step must clamp them to exactly 0
    b = rng.standard_normal(out_dim).astype(np.float32)
    path = tmp_path / "aff.nnet"
    nnet_io.write_simple_nnet(path, [("<AffineTransform>", in_dim, out_dim, nnet_io.affine(W, b, lr_coef, bias_lr_coef, max_norm))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt, l2_penalty=l2, l1_penalty=l1)
    Wc, bc = np.zeros_like(W), np.zeros_like(b)
    o = oracle.AffineOpts(lr, mmt, l2, l1, lr_coef, bias_lr_coef, max_norm)
    import ctypes as C
    b_prev = b.copy()
    for step in range(3):
        x = rng.standard_normal((mb, in_dim)).astype(np.float32)
        od = (rng.standard_normal((mb, out_dim)) * 0.05).astype(np.float32)
        out_ref = np.empty((mb, out_dim), np.float32)
        oracle.lib.orc_affine_propagate(out_ref, out_dim, x, in_dim, mb, W, in_dim, b, in_dim, out_dim)
        idf_ref = np.empty((mb, in_dim), np.float32)
        oracle.lib.orc_affine_backpropagate(idf_ref, in_dim, od, out_dim, mb, W, in_dim, in_dim, out_dim)
        oracle.lib.orc_affine_update(W, in_dim, b, Wc, in_dim, bc, x, in_dim, od, out_dim, mb, in_dim, out_dim, C.byref(o))
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, out_ref) < TOL and oracle.max_err(out, out_ref) < 10 * TOL, ("out", step)
        idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf, idf_ref) < TOL and oracle.max_err(idf, idf_ref) < 10 * TOL, ("in_diff", step)
        got = net.GetParams()
        assert oracle.rel_err(got, np.concatenate([W.ravel(), b])) < TOL and oracle.max_err(got, np.concatenate([W.ravel(), b])) < 10 * TOL, ("params", step)
        # the momentum-carrying gradient buffers themselves: bias_corr is what the bias moved by (no regulariser touches it, :227); the weight
        # gradient is read back through the step only where nothing but -lr * coef * W_corr acted on W
        assert oracle.max_err((b_prev - got[W.size:]) / (lr * bias_lr_coef), bc) < 10 * TOL + 2.0 ** -23 * np.abs(b_prev).max() / (lr * bias_lr_coef), ("bias_corr", step)
        b_prev = got[W.size:].copy()
        if l1:   # the exact zeros land on the same elements (an element whose sign test sits within rounding of 0 may go either way)
            mism = (got[:W.size] == 0) != (W.ravel() == 0)
            assert mism.sum() <= 3 and np.abs(got[:W.size][mism]).max(initial=0) < 1e-7 and np.abs(W.ravel()[mism]).max(initial=0) < 1e-7, step
    if l1:
        assert (W == 0).sum() > 0
    if max_norm:
        assert np.sqrt((got[:W.size].reshape(W.shape) ** 2).sum(1)).max() <= max_norm * (1 + 1e-5)


def test_link_aliasing_is_value_neutral(aslp, oracle, dev, tmp_path):
    """The engine's in-place links must give bit-identical results to the reference's zero+AddMat copies."""
    d, path = make_dnn(oracle, tmp_path, 24, 32, 2, 10, 1, 16, seed=9)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((16, 24)).astype(np.float32)).to(dev)
    lab = torch.from_numpy(rng.integers(0, 10, 16).astype(np.int32)).to(dev)
    res = []
    for alias in (1, 0):
        net = aslp.Nnet.Read(path)
        net.SetLinkAliasing(alias)
        net.SetTrainOptions(learn_rate=0.01, momentum=0.5)
        xe = aslp.Xent()
        for _ in range(3):
            net.TrainStepXent(xe, x, lab)
        res.append(net.GetParams())
    assert np.array_equal(res[0], res[1])
    oracle.lib.orc_dnn_destroy(d)


def test_propagate_backpropagate_api(aslp, oracle, dev, tmp_path):
    """Separate Propagate / Xent / Backpropagate calls (the reference's loop, aslp-nnet-train-frame.cc:109-131)
    including the in-diff w.r.t. the network input."""
    in_dim, hid, out_dim, mb = 20, 16, 7, 12
    rng = np.random.default_rng(2)
    W1 = rng.standard_normal((hid, in_dim)).astype(np.float32) * 0.3
    b1 = rng.standard_normal(hid).astype(np.float32)
    W2 = rng.standard_normal((out_dim, hid)).astype(np.float32) * 0.3
    b2 = rng.standard_normal(out_dim).astype(np.float32)
    path = tmp_path / "n.nnet"
    nnet_io.write_simple_nnet(path, [("<AffineTransform>", in_dim, hid, nnet_io.affine(W1, b1)), ("<Tanh>", hid, hid, b""),
                                     ("<AffineTransform>", hid, out_dim, nnet_io.affine(W2, b2)), ("<Softmax>", out_dim, out_dim, b"")])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.0)  # gradients only
    x = rng.standard_normal((mb, in_dim)).astype(np.float32)
    lab = rng.integers(0, out_dim, mb).astype(np.int32)
    out = net.Propagate(torch.from_numpy(x).to(dev))
    a1 = x @ W1.T + b1
    h = oracle.unary("orc_tanh", a1)
    y = oracle.unary("orc_softmax_rows", h @ W2.T + b2)
    assert oracle.rel_err(out.cpu().numpy(), y) < 1e-5
    tgt = np.zeros((mb, out_dim), np.float32)
    tgt[np.arange(mb), lab] = 1
    diff_ref, _ = oracle.xent_eval(np.ones(mb, np.float32), y, tgt)
    xe = aslp.Xent()
    diff = torch.empty_like(out)
    xe.Eval(torch.ones(mb, device=dev), out, diff, labels=torch.from_numpy(lab).to(dev))
    assert oracle.rel_err(diff.cpu().numpy(), diff_ref) < 1e-5
    in_diff = net.Backpropagate(diff, want_in_diff=True)
    dh = oracle.binary("orc_diff_tanh", h, diff_ref @ W2)
    assert oracle.rel_err(in_diff.cpu().numpy(), dh @ W1) < 1e-5
    # wrong dims raise like the reference's KALDI_ERR "Non-matching dims!"
    with pytest.raises(RuntimeError, match="Non-matching dims"):
        net.Propagate(torch.zeros(4, in_dim + 1, device=dev))


def test_graph_net_multi_input_links(aslp, oracle, dev, tmp_path):
    """Graph nets: a component's input is the zeroed buffer with each producer added at its column offset
    (nnet-nnet.cc:86-95); out-diffs are scatter-added back (:133-144)."""
    rng = np.random.default_rng(4)
    D, mb = 6, 9
    Wa = rng.standard_normal((D, D)).astype(np.float32); ba = rng.standard_normal(D).astype(np.float32)
    Wb = rng.standard_normal((D, D)).astype(np.float32); bb = rng.standard_normal(D).astype(np.float32)
    Wc = rng.standard_normal((4, 2 * D)).astype(np.float32); bc = rng.standard_normal(4).astype(np.float32)
    comps = [
        dict(marker="<InputLayer>", dim_in=D, dim_out=D, id=0, inputs=[-1], offsets=[0]),
        dict(marker="<AffineTransform>", dim_in=D, dim_out=D, id=1, inputs=[0], offsets=[0], data=nnet_io.affine(Wa, ba)),
        dict(marker="<AffineTransform>", dim_in=D, dim_out=D, id=2, inputs=[0], offsets=[0], data=nnet_io.affine(Wb, bb)),
        dict(marker="<Sigmoid>", dim_in=D, dim_out=D, id=3, inputs=[1, 2], offsets=[0, 0]),            # sum of two branches
        dict(marker="<AffineTransform>", dim_in=2 * D, dim_out=4, id=4, inputs=[3, 1], offsets=[0, D], data=nnet_io.affine(Wc, bc)),  # concat
        dict(marker="<OutputLayer>", dim_in=4, dim_out=4, id=5, inputs=[4], offsets=[0]),
    ]
    path = tmp_path / "g.nnet"
    nnet_io.write_graph_nnet(path, comps)
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.0)
    x = rng.standard_normal((mb, D)).astype(np.float32)
    out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
    a, b = x @ Wa.T + ba, x @ Wb.T + bb
    s = oracle.unary("orc_sigmoid", a + b)
    ref = np.concatenate([s, a], 1) @ Wc.T + bc
    assert oracle.rel_err(out, ref) < 1e-5
    od = rng.standard_normal((mb, 4)).astype(np.float32)
    in_diff = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
    dcat = od @ Wc
    ds = oracle.binary("orc_diff_sigmoid", s, dcat[:, :D])
    da = ds + dcat[:, D:]
    ref_in = da @ Wa + ds @ Wb
    assert oracle.rel_err(in_diff, ref_in) < 1e-5


def test_model_file_round_trip_text_and_binary(aslp, oracle, dev, tmp_path):
    proto = """<NnetProto>
<Splice> <InputDim> 8 <OutputDim> 24 <BuildVector> -1:1 </BuildVector>
<AffineTransform> <InputDim> 24 <OutputDim> 16 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.04
<BatchNormalization> <InputDim> 16 <OutputDim> 16
<Sigmoid> <InputDim> 16 <OutputDim> 16
<AffineTransform> <InputDim> 16 <OutputDim> 5 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04
<Softmax> <InputDim> 5 <OutputDim> 5
</NnetProto>
"""
    net = aslp.Nnet.Init(proto, seed=777)
    assert net.NumComponents() == 8  # + InputLayer / OutputLayer (AutoComplete)
    assert [net.Marker(i) for i in (0, 1, 7)] == ["<InputLayer>", "<Splice>", "<OutputLayer>"]
    p0 = net.GetParams()
    net.Write(tmp_path / "b.nnet", binary=True)
    net.Write(tmp_path / "t.nnet", binary=False)
    nb = aslp.Nnet.Read(tmp_path / "b.nnet")
    nt = aslp.Nnet.Read(tmp_path / "t.nnet")
    assert np.array_equal(nb.GetParams(), p0)
    assert oracle.rel_err(nt.GetParams(), p0) < 1e-6  # text keeps 7 significant digits
    x = torch.randn(10, 8, device=dev)
    assert torch.equal(nb.Propagate(x), net.Propagate(x))
    txt = open(tmp_path / "t.nnet").read()
    assert txt.startswith("<Nnet> \n<InputLayer> 8 8 0 [ -1 ]\n[ 0 ]\n\n")  # header, then the "\n" of nnet-component.cc:340
    assert "<Splice> 24 8 1 [ 0 ]\n[ 0 ]\n\n[ -1 0 1 ]\n" in txt and txt.rstrip().endswith("</Nnet>")


@pytest.mark.parametrize("bn", [1, 0])
def test_bn_sigmoid_fusion_is_bit_identical(aslp, oracle, dev, tmp_path, bn):
    """The executor folds a Sigmoid behind a BatchNormalization into the BN kernels, and the forward pass of a Sigmoid behind
    an AffineTransform (nets without BatchNormalization) into that layer's GEMM (Nnet::SetLayerFusion, default on).
    Same float operations in the same order: outputs and updated parameters must be bit-identical to the unfused run."""
    in_dim, hid, nh, out_dim, mb = 40, 96, 3, 50, 128
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, bn, mb, seed=11)
    nets = [aslp.Nnet.Read(path), aslp.Nnet.Read(path)]
    nets[1].SetLayerFusion(False)
    rng = np.random.default_rng(5)
    xents = [aslp.Xent(), aslp.Xent()]
    for step in range(3):
        x = torch.from_numpy(rng.standard_normal((mb, in_dim)).astype(np.float32)).to(dev)
        lab = torch.from_numpy(rng.integers(0, out_dim, mb).astype(np.int32)).to(dev)
        outs = []
        for net, xe in zip(nets, xents):
            net.SetTrainOptions(learn_rate=0.01, momentum=0.9)
            net.TrainStepXent(xe, x, lab)
            outs.append(net.ComponentOutput(net.NumComponents() - 1, mb, out_dim))
        assert np.array_equal(outs[0], outs[1]), step
        assert np.array_equal(nets[0].GetParams(), nets[1].GetParams()), step
    if not bn:  # the AffineTransform fold keeps both outputs: the pre-activation is there and equal
        aff = [c for c in range(nets[0].NumComponents()) if nets[0].Marker(c) == "<AffineTransform>"][0]
        assert np.array_equal(nets[0].ComponentOutput(aff, mb, hid), nets[1].ComponentOutput(aff, mb, hid))
        return
    # the folded intermediate is not materialised: asking for it is an error, not stale data
    bnc = [c for c in range(nets[0].NumComponents()) if nets[0].Marker(c) == "<BatchNormalization>"][0]
    with pytest.raises(RuntimeError):
        nets[0].ComponentOutput(bnc, mb, hid)
    assert nets[1].ComponentOutput(bnc, mb, hid).shape == (mb, hid)


def test_softmax_fold_and_update_overlap_are_bit_identical(aslp, oracle, dev, tmp_path):
    """TrainStepXent leaves a final Softmax (513..8192 classes) to the loss kernel and issues AffineTransform::Update on a
    side stream; both are pure scheduling changes: parameters and loss statistics must equal, bit for bit, those of a net
    with SetLayerFusion(False) / SetUpdateOverlap(False) driven through Propagate / Xent / Backpropagate."""
    in_dim, hid, nh, out_dim, mb = 40, 128, 2, 600, 256
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, 1, mb, seed=12)
    nets = [aslp.Nnet.Read(path) for _ in range(3)]
    nets[1].SetLayerFusion(False)
    nets[1].SetUpdateOverlap(False)
    nets[2].SetUpdateOverlap(False)
    rng = np.random.default_rng(6)
    xents = [aslp.Xent() for _ in range(3)]
    for step in range(4):
        x = torch.from_numpy(rng.standard_normal((mb, in_dim)).astype(np.float32)).to(dev)
        lab = torch.from_numpy(rng.integers(0, out_dim, mb).astype(np.int32)).to(dev)
        for net in nets:
            net.SetTrainOptions(learn_rate=0.01, momentum=0.9)
        nets[0].TrainStepXent(xents[0], x, lab)
        nets[2].TrainStepXent(xents[2], x, lab)
        # the unfused net goes through the three separate calls of the reference's training loop
        y = nets[1].Propagate(x)
        diff = torch.empty_like(y)
        xents[1].Eval(torch.ones(mb, device=dev), y, diff, labels=lab)
        nets[1].Backpropagate(diff)
        p0, p1, p2 = (n.GetParams() for n in nets)
        assert np.array_equal(p0, p1), step
        assert np.array_equal(p0, p2), step
    s0, s1 = xents[0].GetStats(), xents[1].GetStats()
    assert s0 == s1
    with pytest.raises(RuntimeError):
        nets[0].ComponentOutput(nets[0].NumComponents() - 1, mb, out_dim)


def test_dropout_component(aslp, dev, tmp_path):
    """<Dropout> (nnet-activation.h:203-273): out = in * mask / retention with mask ~ Bernoulli(retention), the same mask in
    the backward pass, a fresh one per Propagate; retention 1 is the identity; the retention survives a write / read."""
    D, N, r = 64, 4096, 0.6
    proto = "<NnetProto>\n<Dropout> <InputDim> %d <OutputDim> %d <DropoutRetention> %g\n</NnetProto>\n" % (D, D, r)
    net = aslp.Nnet.Init(proto, seed=3)
    x = torch.randn(N, D, device=dev) + 3.0          # keep away from 0 so that a zero output means "dropped"
    y = net.Propagate(x)
    kept = (y != 0)
    frac = kept.float().mean().item()
    assert abs(frac - r) < 4 * np.sqrt(r * (1 - r) / (N * D))
    assert torch.equal(y[kept], (x * (1.0 / np.float32(r)))[kept]) or torch.allclose(y[kept], x[kept] / r, rtol=1e-6)
    od = torch.randn(N, D, device=dev)
    idiff = net.Backpropagate(od, want_in_diff=True)
    assert torch.equal(idiff != 0, kept & (od != 0))
    assert torch.allclose(idiff[kept], od[kept] / r, rtol=1e-6)
    y2 = net.Propagate(x)
    assert not torch.equal(y2 != 0, kept)             # a new mask every call
    # columns and rows are all hit (no stripe pattern from the index hash)
    assert kept.float().mean(0).min() > r - 0.1 and kept.float().mean(1).min() > r - 0.35
    path = tmp_path / "d.nnet"
    net.Write(path, binary=False)
    assert "<DropoutRetention> 0.6" in open(path).read()
    net2 = aslp.Nnet.Read(path)
    assert net2.NumComponents() == net.NumComponents()
    one = aslp.Nnet.Init(proto.replace("%g" % r, "1.0"), seed=3)
    assert torch.equal(one.Propagate(x), x)
    assert torch.equal(one.Backpropagate(od, want_in_diff=True), od)
