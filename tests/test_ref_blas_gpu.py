"""The HIP engine against outputs of the REFERENCE's own CuMatrix / CuVector CPU branch linked to a real BLAS
(tests/golden/cumatrix_blas_ops.bin, generator oracle/gen_cumatrix_blas_golden.cpp): the product in four layouts, softmax, the column
sums, BatchNormalization forward / backward over two minibatches, and the AffineTransform / LstmProjectedStreams / GruStreams
components driven through the C ABI on the fixture's weights -- no oracle in between.  Tolerances as in
tests/test_oracle_ref_blas_cpu.py (the bar is 1e-4)."""
import numpy as np
import pytest
import torch

import cumatrix_golden
import nnet_io

pytestmark = pytest.mark.gpu


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def T(a, dev, dt=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)


def test_kernels_match_reference_blas_branch(aslp, dev):
    g = cumatrix_golden.load_blas()
    ops = aslp.ops
    for s in (0, 1):
        for lay, ta, tb in (("nn", 0, 0), ("nt", 0, 1), ("tn", 1, 0), ("tt", 1, 1)):
            k = "gemm%d_%s_" % (s, lay)
            c = T(g[k + "Cin"], dev)
            ops.sgemm(ta, tb, 0.7, T(g[k + "A"], dev), T(g[k + "B"], dev), 0.3, c)
            assert rel(c.cpu().numpy(), g[k + "Cout"]) < 2e-6, k
    x = T(g["softmax_in"], dev)
    y = torch.empty_like(x)
    ops.softmax(y, x)
    assert close(y.cpu().numpy(), g["softmax_out"], 1e-6)
    v = T(g["colsum_v_in"], dev)
    ops.add_row_sum_mat_vec(0.43, T(g["sum_in"], dev), 1.4, v)
    assert close(v.cpu().numpy(), g["colsum_v_out"], 2e-5)


def test_batchnorm_kernels_match_reference_library(aslp, dev):
    """nnet-batch-normalization.h:177-284 as the reference's library computes it, two minibatches."""
    g = cumatrix_golden.load_blas()
    ops = aslp.ops
    dim = g["bn_scale0"].shape[0]
    accm = torch.zeros(dim, dtype=torch.float64, device=dev)
    accv = torch.zeros(dim, dtype=torch.float64, device=dev)
    dsc, dsh = torch.zeros(dim, device=dev), torch.zeros(dim, device=dev)
    for step in (0, 1):
        x, od = T(g["bn_in%d" % step], dev), T(g["bn_od%d" % step], dev)
        scale, shift = T(g["bn_scale%d" % step], dev), T(g["bn_shift%d" % step], dev)
        out, xhat = torch.empty_like(x), torch.empty_like(x)
        mean, inv = torch.empty(dim, device=dev), torch.empty(dim, device=dev)
        ops.bn_forward(x, out, xhat, scale, shift, mean, inv, accm, accv)
        assert close(out.cpu().numpy(), g["bn_out%d" % step], 1e-5)
        assert close(mean.cpu().numpy(), g["bn_mean%d" % step], 2e-6) and close(inv.cpu().numpy(), g["bn_invstd%d" % step], 1e-5)
        assert np.allclose(accm.cpu().numpy(), g["bn_accm%d" % step], rtol=1e-12) and np.allclose(accv.cpu().numpy(), g["bn_accv%d" % step], rtol=1e-12)
        idf = torch.empty_like(x)
        ops.bn_backward(x, od, xhat, scale, mean, inv, dsc, dsh, 0.0 if step == 0 else 0.9, idf)
        assert close(idf.cpu().numpy(), g["bn_id%d" % step], 2e-5)
        assert close(dsc.cpu().numpy(), g["bn_dscale%d" % step], 2e-5) and close(dsh.cpu().numpy(), g["bn_dshift%d" % step], 2e-5)


def test_affine_component_matches_reference_library(aslp, dev, tmp_path):
    """nnet-affine-transform.h:186-245 with momentum 0.9, l2 1e-3, bias-learn-rate-coef 0.5, <MaxNorm> 0.9, two minibatches."""
    g = cumatrix_golden.load_blas()
    W, b = g["aff_W0"], g["aff_b0"]
    dout, din = W.shape
    path = tmp_path / "aff.nnet"
    nnet_io.write_simple_nnet(path, [("<AffineTransform>", din, dout, nnet_io.affine(W, b, 1.0, 0.5, 0.9))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.02, momentum=0.9, l2_penalty=1e-3, l1_penalty=0.0)
    for step in (0, 1):
        out = net.Propagate(T(g["aff_in%d" % step], dev)).cpu().numpy()
        assert close(out, g["aff_out%d" % step], 2e-5)
        idf = net.Backpropagate(T(g["aff_od%d" % step], dev), want_in_diff=True).cpu().numpy()
        assert close(idf, g["aff_id%d" % step], 2e-5)
        ref = np.concatenate([g["aff_W%d" % (step + 1)].ravel(), g["aff_b%d" % (step + 1)]])
        assert close(net.GetParams(), ref, 2e-5), step


def test_projected_lstm_component_matches_reference_library(aslp, oracle, dev, tmp_path):
    """nnet-lstm-projected-streams.h:313-617: output r(1..T), input diff, and parameters after one step (learn rate 0.1, momentum 0, no
    clipping) = W - 0.1 * the reference library's gradients."""
    g = {k[5:]: v for k, v in cumatrix_golden.load_blas().items() if k.startswith("lstm_")}
    Cc, D, R = g["Wx"].shape[0] // 4, g["Wx"].shape[1], g["Wr"].shape[1]
    S = 3
    Tn = g["in"].shape[0] // S
    d = oracle.LstmDir(D, Cc, R, False, zero=True)   # container of the fixture's tensors for the model writer (no oracle arithmetic runs)
    names = (("w_x", "Wx"), ("w_r", "Wr"), ("bias", "bias"), ("peep_i", "pi"), ("peep_f", "pf"), ("peep_o", "po"), ("w_rm", "Wrm"))
    for n, k in names:
        getattr(d, n)[...] = g[k]
    path = tmp_path / "lstm.nnet"
    nnet_io.write_simple_nnet(path, [("<LstmProjectedStreams>", D, R, nnet_io.lstm([d], 0.0, Cc))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.1, momentum=0.0)
    net.ResetLstmStreams([1] * S)
    out = net.Propagate(T(g["in"], dev)).cpu().numpy()
    assert close(out, g["fwd_buf"][S:(Tn + 1) * S, 7 * Cc:], 2e-6)
    idf = net.Backpropagate(T(g["od"], dev), want_in_diff=True).cpu().numpy()
    assert close(idf, g["in_diff"], 5e-6)
    ref = np.concatenate([(g[k] - 0.1 * g["g" + ("b" if k == "bias" else k)]).ravel() for _, k in names])
    assert close(net.GetParams(), ref, 5e-6)


def test_projected_lstm_two_training_steps_momentum_clip_match_reference_library(aslp, oracle, dev, tmp_path):
    """`lstm2`: two training steps at C = 64, R = 32, S = 8 streams (sizes the persistent recurrence kernels serve), momentum 0.9, element-wise
    clipping 0.5, learn rate 0.01, against the parameters the reference's library produces when the update is issued as
    nnet-blstm-projected-streams-lc.h:976-1016, 1085-1098 issues it (oracle/gen_cumatrix_blas_golden.cpp LstmProjectedTrain)."""
    g = {k[6:]: v for k, v in cumatrix_golden.load_blas().items() if k.startswith("lstm2_")}
    Cc, D, R = g["Wx0"].shape[0] // 4, g["Wx0"].shape[1], g["Wr0"].shape[1]
    S = 8
    d = oracle.LstmDir(D, Cc, R, False, zero=True)   # container of the fixture's tensors for the model writer (no oracle arithmetic runs)
    names = (("w_x", "Wx"), ("w_r", "Wr"), ("bias", "bias"), ("peep_i", "pi"), ("peep_f", "pf"), ("peep_o", "po"), ("w_rm", "Wrm"))
    for n, k in names:
        getattr(d, n)[...] = g[k + "0"]
    path = tmp_path / "lstm2.nnet"
    nnet_io.write_simple_nnet(path, [("<LstmProjectedStreams>", D, R, nnet_io.lstm([d], 0.5, Cc))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.01, momentum=0.9)
    for step in (0, 1):
        net.ResetLstmStreams([1] * S)
        out = net.Propagate(T(g["in%d" % step], dev)).cpu().numpy()
        assert close(out, g["out%d" % step], 5e-6), step
        idf = net.Backpropagate(T(g["od%d" % step], dev), want_in_diff=True).cpu().numpy()
        assert close(idf, g["in_diff%d" % step], 1e-5), step
        ref = np.concatenate([g["%s%d" % (k, step + 1)].ravel() for _, k in names])
        assert close(net.GetParams(), ref, 1e-5), step


def _fullwidth_case(oracle):
    """tensors of tests/golden/lstm_fullwidth.bin regenerated in the generator's own order (oracle/gen_cumatrix_blas_golden.cpp LstmProjectedTrain)"""
    g, state = cumatrix_golden.load_fullwidth()
    Tn, S, D, Cc, R = 60, 32, 512, 512, 256
    rng = oracle.GoldenRng(state)
    p = dict(w_x=rng.fill((4 * Cc, D), -0.02, 0.02), w_r=rng.fill((4 * Cc, R), -0.02, 0.02), w_rm=rng.fill((R, Cc), -0.02, 0.02),
             bias=rng.fill((4 * Cc,), -0.3, 0.3), peep_i=rng.fill((Cc,), -0.3, 0.3), peep_f=rng.fill((Cc,), -0.3, 0.3), peep_o=rng.fill((Cc,), -0.3, 0.3))
    steps = [(rng.fill((Tn * S, D), -1.5, 1.5), rng.fill((Tn * S, R), -1.0, 1.0)) for _ in range(2)]
    return g, p, steps, (Tn, S, D, Cc, R)


def _digest_close(x, g, key, tol):
    """x against the digest records `key` / `key#`: the sampled elements to tol (relative to max(1, |ref|) per element AND in norm), the sums
    to tol of the sum of magnitudes"""
    pick, sums = cumatrix_golden.digest_of(x)
    ref = g[key]
    assert pick.shape == ref.shape, (key, pick.shape, ref.shape)
    assert sums[2] == g[key + "#"][2], key
    ok = close(pick, ref, tol) and rel(pick, ref) <= tol
    scale = np.sqrt(sums[2] * g[key + "#"][1])     # >= sum |ref| (Cauchy-Schwarz): the size a sum's error is held against
    return ok and abs(sums[0] - g[key + "#"][0]) <= tol * max(scale, 1.0) and abs(sums[1] - g[key + "#"][1]) <= 2 * tol * g[key + "#"][1]


def test_projected_lstm_full_width_two_training_steps_match_reference_library(aslp, oracle, dev, tmp_path):
    """`lcfull` (tests/golden/lstm_fullwidth.bin): ONE projected-LSTM layer at BASELINE cfg3's widths -- 512 cells, 256-wide projection, 512
    inputs, S = 32 streams, T = 60 frames (chunk 40 + right context 20) -- through two training steps with momentum 0.9, element-wise
    clipping 5 (5-13 % of the W_rm gradients clipped) and learn rate 0.002, as nnet-lstm-projected-streams.h:313-617 /
    nnet-blstm-projected-streams-lc.h:976-1016, 1085-1098 issue them ON THE REFERENCE'S LIBRARY.  Here the full-width persistent recurrences
    (four chains of 32 workgroups), the 1920-row layer products from prepared planes and the fused update run; outputs, input diffs and every
    parameter tensor after each step are held against the digest (every 61st element + sums)."""
    g, p, steps, (Tn, S, D, Cc, R) = _fullwidth_case(oracle)
    d = oracle.LstmDir(D, Cc, R, False, zero=True)   # container of the tensors for the model writer (no oracle arithmetic runs)
    for n, v in p.items():
        getattr(d, n)[...] = v
    path = tmp_path / "lcfull.nnet"
    nnet_io.write_simple_nnet(path, [("<LstmProjectedStreams>", D, R, nnet_io.lstm([d], 5.0, Cc))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.002, momentum=0.9)
    names = (("w_x", "Wx"), ("w_r", "Wr"), ("bias", "bias"), ("peep_i", "pi"), ("peep_f", "pf"), ("peep_o", "po"), ("w_rm", "Wrm"))
    sizes = [p[n].size for n, _ in names]
    for step, (x, od) in enumerate(steps):
        net.ResetLstmStreams([1] * S)
        out = net.Propagate(T(x, dev)).cpu().numpy()
        assert _digest_close(out, g, "out%d" % step, 1e-4), ("out", step)
        idf = net.Backpropagate(T(od, dev), want_in_diff=True).cpu().numpy()
        assert _digest_close(idf, g, "in_diff%d" % step, 1e-4), ("in_diff", step)
        params = np.split(net.GetParams(), np.cumsum(sizes)[:-1])
        for (n, k), v in zip(names, params):
            key = "%s%d" % (k, step + 1)
            if key + "#" in g:
                assert _digest_close(v, g, key, 1e-4), (key, step)
            else:   # vectors are stored whole
                assert close(v, g[key], 1e-4) and rel(v, g[key]) <= 1e-4, (key, step)


@pytest.mark.parametrize("split16", [1, 0])
def test_dnn_cfg2_full_size_two_training_steps_match_reference_library(aslp, oracle, dev, tmp_path, split16):
    """tests/golden/dnn_cfg2_fullsize.bin: BASELINE cfg2 ITSELF -- 440 -> 5 x 2048 + BatchNormalization + Sigmoid -> 3000, minibatch 1024, learn
    rate 0.008 (run_bn_dnn.sh:64-101) -- two training steps on fresh minibatches, issued on the REFERENCE'S LIBRARY in the order of
    Nnet::Propagate / Backpropagate (oracle/ref_dnn_bench.cpp golden mode), no oracle in between: posteriors, loss and every parameter after
    each step against the digest (every 257th element + sums), on the default split-fp16 path and on the fp32 instruction."""
    g, state, stride = cumatrix_golden.load_cfg2_fullsize()
    W, batches = cumatrix_golden.replay_cfg2(oracle.GoldenRng(state))
    layers = []
    for l, w in enumerate(W):
        layers.append(("<AffineTransform>", w.shape[1], w.shape[0], nnet_io.affine(w, np.zeros(w.shape[0], np.float32))))
        if l < len(W) - 1:
            layers.append(("<BatchNormalization>", w.shape[0], w.shape[0], nnet_io.batchnorm(np.zeros(w.shape[0]), np.ones(w.shape[0]))))
            layers.append(("<Sigmoid>", w.shape[0], w.shape[0], b""))
    layers.append(("<Softmax>", 3000, 3000, b""))
    path = tmp_path / "cfg2.nnet"
    nnet_io.write_simple_nnet(path, layers)

    def digest_close(x, key, tol):
        pick, sums = cumatrix_golden.digest_of(x, stride)
        assert pick.shape == g[key].shape and sums[2] == g[key + "#"][2], key
        return close(pick, g[key], tol) and rel(pick, g[key]) <= tol and abs(sums[1] - g[key + "#"][1]) <= 4 * tol * g[key + "#"][1]
    aslp.lib.aslp_gemm_split16(split16)
    try:
        net = aslp.Nnet.Read(path)
        net.SetTrainOptions(learn_rate=0.008, momentum=0.0)
        net.SetLayerFusion(True)
        xent = aslp.Xent()
        seen = 0.0
        for s, (x, lab) in enumerate(batches):
            net.TrainStepXent(xent, T(x, dev), torch.from_numpy(lab).to(dev))
            st = xent.GetStats()
            loss = (st["loss"] - st["entropy"]) - seen
            seen += loss
            assert abs(loss - g["loss%d" % s]) <= 1e-4 * abs(g["loss%d" % s]), (s, loss, g["loss%d" % s])
            params = net.GetParams()
            off = 0
            for l, w in enumerate(W):   # GetParams order: W, b, then BatchNormalization's shift and scale
                n = w.size
                assert digest_close(params[off:off + n], "W%d_%d" % (l, s + 1), 1e-4), (s, l)
                off += n
                assert close(params[off:off + w.shape[0]], g["b%d_%d" % (l, s + 1)], 1e-4), (s, l, "bias")
                off += w.shape[0]
                if l < len(W) - 1:
                    assert close(params[off:off + 2048], g["sh%d_%d" % (l, s + 1)], 1e-4), (s, l, "shift")
                    assert close(params[off + 2048:off + 4096], g["sc%d_%d" % (l, s + 1)], 1e-4), (s, l, "scale")
                    off += 4096
            assert off == params.size
        # the posteriors of a third forward pass are not in the fixture; those of the two steps are checked through the loss and through
        # the parameters they produced (the final Softmax is folded into the loss kernel on this path)
    finally:
        aslp.lib.aslp_gemm_split16(-1)


def test_gru_component_matches_reference_library(aslp, oracle, dev, tmp_path):
    """nnet-gru-streams.h:238-450: output h(1..T), input diff, parameters after one step."""
    g = {k[4:]: v for k, v in cumatrix_golden.load_blas().items() if k.startswith("gru_")}
    H, D = g["Wg"].shape[0], g["Wx"].shape[1]
    S = 3
    Tn = g["in"].shape[0] // S
    u = oracle.Gru(D, H, zero=True)
    names = (("w_zrm_x", "Wx"), ("w_zr_h", "Wh"), ("w_m_g", "Wg"), ("bias", "bias"))
    for n, k in names:
        getattr(u, n)[...] = g[k]
    path = tmp_path / "gru.nnet"
    nnet_io.write_simple_nnet(path, [("<GruStreams>", D, H, nnet_io.gru(u, 0.0))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.1, momentum=0.0)
    net.ResetLstmStreams([1] * S)
    out = net.Propagate(T(g["in"], dev)).cpu().numpy()
    assert close(out, g["fwd_buf"][S:(Tn + 1) * S, 4 * H:], 2e-6)
    idf = net.Backpropagate(T(g["od"], dev), want_in_diff=True).cpu().numpy()
    assert close(idf, g["in_diff"], 5e-6)
    ref = np.concatenate([(g[k] - 0.1 * g["g" + ("b" if k == "bias" else k)]).ravel() for _, k in names])
    assert close(net.GetParams(), ref, 5e-6)


def test_gru_component_at_full_width_matches_reference_library(aslp, oracle, dev, tmp_path):
    """`grufull` (tests/golden/temporal_fullsize.bin): <GruStreams> 512 -> 512 with S = 32 streams and T = 60 frames -- the persistent GRU recurrence at
    cfg5's width -- against the reference's library directly: output h(1..T), input diff, and the update each tensor was moved by (learn rate
    0.001, momentum 0) against the library's gradients."""
    _, _, t = cumatrix_golden.load_temporal_fullsize(oracle.GoldenRng)
    g, stride = t["gru"], cumatrix_golden.DIR_STRIDE
    Tn, S, D, H = 60, 32, 512, 512
    u = oracle.Gru(D, H, zero=True)      # container of the tensors for the model writer (no oracle arithmetic runs)
    names = (("w_zrm_x", "gru_Wx", "gWx"), ("w_zr_h", "gru_Wh", "gWh"), ("w_m_g", "gru_Wg", "gWg"), ("bias", "gru_bias", "gb"))
    for n, k, _ in names:
        getattr(u, n)[...] = t[k]
    path = tmp_path / "gru.nnet"
    nnet_io.write_simple_nnet(path, [("<GruStreams>", D, H, nnet_io.gru(u, 0.0))])
    net = aslp.Nnet.Read(path)
    lr = 0.001
    net.SetTrainOptions(learn_rate=lr, momentum=0.0)
    net.ResetLstmStreams([1] * S)
    out = net.Propagate(T(t["gru_in"], dev)).cpu().numpy()
    # the digest is over the whole [(T + 2) S x 5H] buffer: rebuild the h columns of frames 1..T inside a zero buffer and compare what a digest of
    # THAT keeps with the library's samples that fall on those positions
    W5 = 5 * H
    pos = np.arange(0, (Tn + 2) * S * W5, stride)
    rows, cols = pos // W5, pos % W5
    sel = (rows >= S) & (rows < (Tn + 1) * S) & (cols >= 4 * H)
    assert sel.sum() > 1000
    assert close(out[rows[sel] - S, cols[sel] - 4 * H], g["fwd_buf"][sel], 1e-4) and rel(out[rows[sel] - S, cols[sel] - 4 * H], g["fwd_buf"][sel]) <= 1e-4
    idf = net.Backpropagate(T(t["gru_od"], dev), want_in_diff=True).cpu().numpy()
    pick, sums = cumatrix_golden.digest_of(idf, stride)
    assert close(pick, g["in_diff"], 1e-4) and rel(pick, g["in_diff"]) <= 1e-4 and abs(sums[1] - g["in_diff#"][1]) <= 4e-4 * g["in_diff#"][1]
    params = net.GetParams()
    off = 0
    for n, k, gk in names:
        w0 = t[k].ravel().astype(np.float64)
        applied = ((w0 - params[off:off + w0.size]) / lr).astype(np.float32)
        off += w0.size
        ref = g[gk]
        got = cumatrix_golden.digest_of(applied, stride)[0] if gk + "#" in g else applied
        assert rel(got, ref) <= 1e-4 + 2.0 ** -22 * np.abs(w0).max() / lr / np.abs(ref).mean(), (n, rel(got, ref))
    assert off == params.size


def test_rowconvolution_component_matches_reference_library(aslp, dev, tmp_path):
    """nnet-row-convolution.cc:90-169, ragged lengths: output, input diff, taps after one step (learn rate 0.1, momentum 0)."""
    g = cumatrix_golden.load_blas()
    w = g["rc_w"]
    D = w.shape[0]
    path = tmp_path / "rc.nnet"
    nnet_io.write_simple_nnet(path, [("<RowConvolution>", D, D, nnet_io.rowconv(w))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.1, momentum=0.0)
    net.SetSeqLengths(g["rc_lens"].astype(np.int32))
    lens, S = g["rc_lens"], len(g["rc_lens"])
    valid = np.array([(r // S) < lens[r % S] for r in range(g["rc_in"].shape[0])])   # rows past a stream's length are not defined
    out = net.Propagate(T(g["rc_in"], dev)).cpu().numpy()
    assert close(out[valid], g["rc_out"][valid], 2e-6)
    idf = net.Backpropagate(T(g["rc_od"], dev), want_in_diff=True).cpu().numpy()
    assert close(idf[valid], g["rc_in_diff"][valid], 2e-6)
    assert close(net.GetParams(), (w - 0.1 * g["rc_w_diff"]).ravel(), 5e-6)


def test_compact_fsmn_component_matches_reference_library(aslp, dev, tmp_path):
    """nnet-cfsmn-component.h:169-264: output, input diff, taps after one step."""
    g = cumatrix_golden.load_blas()
    coef = g["fsmn_coef"]
    D = coef.shape[1]
    path = tmp_path / "fsmn.nnet"
    nnet_io.write_simple_nnet(path, [("<CompactFsmn>", D, D, nnet_io.fsmn(coef, 3, 2))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.1, momentum=0.0)
    out = net.Propagate(T(g["fsmn_in"], dev)).cpu().numpy()
    assert close(out, g["fsmn_out"], 2e-6)
    idf = net.Backpropagate(T(g["fsmn_od"], dev), want_in_diff=True).cpu().numpy()
    assert close(idf, g["fsmn_in_diff"], 2e-6)
    assert close(net.GetParams(), (coef - 0.1 * g["fsmn_corr"]).ravel(), 5e-6)


def test_temporal_components_at_full_size_match_reference_library(aslp, oracle, dev, tmp_path):
    """tests/golden/temporal_fullsize.bin: the streaming RowConvolution kernels and the LDS-tiled CompactFsmn kernels (csrc/temporal.hip, round 5) at
    the sizes BASELINE cfg5 swaps the components in at, against the reference's library directly: output, input diff, and the taps after one
    step (learn rate 0.01, momentum 0) = w - 0.01 x the library's gradient."""
    rc, fs, t = cumatrix_golden.load_temporal_fullsize(oracle.GoldenRng)
    stride = cumatrix_golden.DIR_STRIDE

    def digest_close(a, g, key, tol):
        pick, sums = cumatrix_golden.digest_of(a, stride)
        assert pick.shape == g[key].shape and sums[2] == g[key + "#"][2], key
        return close(pick, g[key], tol) and rel(pick, g[key]) <= tol and abs(sums[1] - g[key + "#"][1]) <= 4 * tol * g[key + "#"][1]
    path = tmp_path / "rc.nnet"
    nnet_io.write_simple_nnet(path, [("<RowConvolution>", 512, 512, nnet_io.rowconv(t["rc_w"]))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.01, momentum=0.0)
    net.SetSeqLengths(t["lens"])
    assert digest_close(net.Propagate(T(t["rc_in"], dev)).cpu().numpy(), rc, "out", 1e-4)
    assert digest_close(net.Backpropagate(T(t["rc_od"], dev), want_in_diff=True).cpu().numpy(), rc, "in_diff", 1e-4)
    # the applied update, read back as (w_before - w_after) / lr, against the library's gradient digest
    applied = (t["rc_w"].ravel().astype(np.float64) - net.GetParams()) / 0.01
    pick, _ = cumatrix_golden.digest_of(applied.astype(np.float32), stride)
    assert rel(pick, rc["w_diff"]) <= 1e-4 + 2.0 ** -22 * np.abs(t["rc_w"]).max() / 0.01 / np.abs(rc["w_diff"]).mean(), rel(pick, rc["w_diff"])
    path = tmp_path / "fsmn.nnet"
    nnet_io.write_simple_nnet(path, [("<CompactFsmn>", 512, 512, nnet_io.fsmn(t["fsmn_coef"], 30, 30))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.01, momentum=0.0)
    assert digest_close(net.Propagate(T(t["fsmn_in"], dev)).cpu().numpy(), fs, "out", 1e-4)
    assert digest_close(net.Backpropagate(T(t["fsmn_od"], dev), want_in_diff=True).cpu().numpy(), fs, "in_diff", 1e-4)
    applied = (t["fsmn_coef"].ravel().astype(np.float64) - net.GetParams()) / 0.01
    pick, _ = cumatrix_golden.digest_of(applied.astype(np.float32), stride)
    assert rel(pick, fs["corr"]) <= 1e-4 + 2.0 ** -22 * np.abs(t["fsmn_coef"]).max() / 0.01 / np.abs(fs["corr"]).mean(), rel(pick, fs["corr"])


@pytest.mark.parametrize("tag,marker,cifg", [("cifg", "<LstmCifgProjectedStreams>", True), ("lstmnp", "<Lstm>", False)])
def test_other_lstm_components_match_reference_library(aslp, oracle, dev, tmp_path, tag, marker, cifg):
    """LstmCifgProjectedStreams (nnet-lstm-couple-if-projected-streams.h) and Lstm (nnet-recurrent-component.cc:235-420): output,
    input diff, parameters after one step (learn rate 0.1, momentum 0, no clipping)."""
    g = {k[len(tag) + 1:]: v for k, v in cumatrix_golden.load_blas().items() if k.startswith(tag + "_")}
    ng = 3 if cifg else 4
    Cc, D = g["Wx"].shape[0] // ng, g["Wx"].shape[1]
    R = g["Wrm"].shape[0] if "Wrm" in g else 0
    S = 3
    Tn = g["in"].shape[0] // S
    d = oracle.LstmDir(D, Cc, R, cifg, zero=True)   # container of the fixture's tensors for the model writer (no oracle arithmetic runs)
    key = {"w_x": "Wx", "w_r": "Wr", "bias": "bias", "peep_i": "pi", "peep_f": "pf", "peep_o": "po", "w_rm": "Wrm"}
    order = ["w_x", "w_r", "bias"] + ([] if cifg else ["peep_i"]) + ["peep_f", "peep_o"] + (["w_rm"] if R else [])   # file / GetParams order
    for n in order:
        getattr(d, n)[...] = g[key[n]]
    path = tmp_path / "l.nnet"
    nnet_io.write_simple_nnet(path, [(marker, D, R if R else Cc, nnet_io.lstm([d], 0.0, Cc if R else None))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=0.1, momentum=0.0)
    net.ResetLstmStreams([1] * S)
    out = net.Propagate(T(g["in"], dev)).cpu().numpy()
    off = (ng + 3) * Cc if R else (ng + 2) * Cc
    assert close(out, g["fwd_buf"][S:(Tn + 1) * S, off:off + (R if R else Cc)], 2e-6)
    idf = net.Backpropagate(T(g["od"], dev), want_in_diff=True).cpu().numpy()
    assert close(idf, g["in_diff"], 5e-6)
    ref = np.concatenate([(g[key[n]] - 0.1 * g["g" + ("b" if n == "bias" else key[n])]).ravel() for n in order])
    assert close(net.GetParams(), ref, 5e-6)


def test_xent_eval_matches_reference_library(aslp, dev):
    """Xent::Eval, nnet-loss.cc:63-156 as the reference's library computes it: diff and {frames, correct, loss, entropy, likelihood}."""
    g = cumatrix_golden.load_blas()
    y = T(g["xe_y"], dev)
    diff = torch.empty_like(y)
    stats = torch.zeros(5, dtype=torch.float64, device=dev)
    aslp.ops.xent_eval(y, T(g["xe_fw"], dev), diff, stats, targets=T(g["xe_tgt"], dev))
    assert close(diff.cpu().numpy(), g["xe_diff"], 1e-6)
    assert np.allclose(stats.cpu().numpy(), g["xe_stats"], rtol=2e-6, atol=1e-6), (stats.cpu().numpy(), g["xe_stats"])
