"""The C oracle against the REFERENCE's own CuMatrix / CuVector CPU branch linked to a real BLAS (tests/golden/cumatrix_blas_ops.bin,
generator oracle/gen_cumatrix_blas_golden.cpp; the BLAS is the OpenBLAS inside the image's scipy wheel).  Pins what
tests/golden/cumatrix_ops.bin could not reach: the product (row a1), softmax (a4), the column / row sums incl. the double ones of
BatchNormalization, the wide bias broadcasts -- and the oracle's AffineTransform (a2), BatchNormalization (a7) and projected-LSTM
(a8 / a9: the gate block all family members share) chains against the same op sequences issued on the reference's library.
Tolerance: where a BLAS reduction is involved the summation order differs, so those records are compared with the reference's own
AssertEqual metric (relative Frobenius error, cu-matrix.h:803-811) at 2e-6 for the bare products and element-wise at 2e-5 of
max(1, |ref|) for the short chains; 1e-6 element-wise otherwise; the bar is 1e-4."""
import ctypes as C

import numpy as np

import cumatrix_golden


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol


def test_blas_backed_operations(oracle):
    g = cumatrix_golden.load_blas()
    for s in (0, 1):
        for lay, ta, tb in (("nn", 0, 0), ("nt", 0, 1), ("tn", 1, 0), ("tt", 1, 1)):
            k = "gemm%d_%s_" % (s, lay)
            out = oracle.add_mat_mat(g[k + "Cin"], 0.7, g[k + "A"], ta, g[k + "B"], tb, 0.3)
            assert oracle.rel_err(out, g[k + "Cout"]) < 2e-6, k
    y = oracle.unary("orc_softmax_rows", g["softmax_in"])
    assert close(y, g["softmax_out"], 1e-6)
    assert abs(g["softmax_out"][5, 17] - 1.0) < 1e-6 and np.allclose(g["softmax_out"][6], 1.0 / 300, rtol=1e-6)
    m = g["sum_in"]
    # CuVector::AddRowSumMat / AddColSumMat: v = beta v + alpha * column (row) sums
    assert close(1.4 * g["colsum_v_in"] + 0.43 * m.astype(np.float64).sum(0), g["colsum_v_out"], 2e-5)
    assert close(0.5 * g["rowsum_v_in"] - 0.6 * m.astype(np.float64).sum(1), g["rowsum_v_out"], 2e-5)
    assert np.allclose(m.astype(np.float64).sum(0), g["colsum_d"], rtol=1e-13)
    assert np.allclose((m * m).astype(np.float64).sum(0), g["colsumsq_d"], rtol=1e-13)   # squares rounded to float first
    L, c32 = oracle.lib, oracle.c32
    cp = lambda a: np.array(a, dtype=np.float32, order="C", copy=True)
    r, c = g["bc_in"].shape
    d = cp(g["bc_in"]); L.orc_add_vec_to_rows(d, c, c32(g["bc_row"]), r, c, 0.5); assert close(d, g["add_vec_to_rows"], 1e-6)
    d = cp(g["bc_in"]); L.orc_add_vec_to_cols(d, c, c32(g["bc_col"]), r, c, -1.5); assert close(d, g["add_vec_to_cols"], 1e-6)
    assert np.array_equal(g["add_vec_to_rows_beta0"], np.broadcast_to(g["bc_row"], (r, c)))
    d = cp(g["grp_dst_in"])
    L.orc_add_row_sum_mat(d, d.shape[1], d.shape[0], d.shape[1], c32(g["grp_src"]), d.shape[1], g["grp_src"].shape[0], 0.8, 0.25)
    assert close(d, g["grp_dst_out"], 2e-6)


def test_affine_chain_matches_reference_library(oracle):
    """nnet-affine-transform.h:186-245 with momentum 0.9, l2 1e-3, bias-learn-rate-coef 0.5, max-norm 0.9, two minibatches."""
    g = cumatrix_golden.load_blas()
    L, c32 = oracle.lib, oracle.c32
    W = np.array(g["aff_W0"], np.float32, copy=True); b = np.array(g["aff_b0"], np.float32, copy=True)
    dout, din = W.shape
    Wc = np.zeros_like(W); bc = np.zeros_like(b)
    o = oracle.AffineOpts(0.02, 0.9, 1e-3, 0.0, 1.0, 0.5, 0.9)
    for step in (0, 1):
        x, od = g["aff_in%d" % step], g["aff_od%d" % step]
        rows = x.shape[0]
        out = np.zeros((rows, dout), np.float32)
        L.orc_affine_propagate(out, dout, c32(x), din, rows, W, din, b, din, dout)
        assert close(out, g["aff_out%d" % step], 2e-5)
        idf = np.zeros((rows, din), np.float32)
        L.orc_affine_backpropagate(idf, din, c32(od), dout, rows, W, din, din, dout)
        assert close(idf, g["aff_id%d" % step], 2e-5)
        L.orc_affine_update(W, din, b, Wc, din, bc, c32(x), din, c32(od), dout, rows, din, dout, C.byref(o))
        assert close(Wc, g["aff_Wc%d" % (step + 1)], 2e-5) and close(bc, g["aff_bc%d" % (step + 1)], 2e-5)
        assert close(W, g["aff_W%d" % (step + 1)], 2e-5) and close(b, g["aff_b%d" % (step + 1)], 2e-5)
    nrm = np.sqrt((g["aff_W2"].astype(np.float64) ** 2).sum(1))
    assert nrm.max() <= 0.9 * (1 + 1e-5) and (nrm > 0.9 * (1 - 1e-5)).sum() > 0   # the max-norm branch did shrink some rows


def test_batchnorm_chain_matches_reference_library(oracle):
    """nnet-batch-normalization.h:177-284, two minibatches: running sums in double, momentum on the second gradient."""
    g = cumatrix_golden.load_blas()
    dim = g["bn_scale0"].shape[0]
    bn = oracle.Bn(dim)
    bn.scale[:] = g["bn_scale0"]; bn.shift[:] = g["bn_shift0"]
    for step in (0, 1):
        x, od = g["bn_in%d" % step], g["bn_od%d" % step]
        out = bn.propagate(x)
        assert close(out, g["bn_out%d" % step], 1e-5)
        assert close(bn.xs, g["bn_xhat%d" % step], 1e-5)
        assert close(bn.mean, g["bn_mean%d" % step], 2e-6) and close(bn.var, g["bn_invstd%d" % step], 1e-5)
        assert np.allclose(bn.acc_means, g["bn_accm%d" % step], rtol=1e-12) and np.allclose(bn.acc_vars, g["bn_accv%d" % step], rtol=1e-12)
        idf = bn.backpropagate(x, od, 0.0 if step == 0 else 0.9)
        assert close(idf, g["bn_id%d" % step], 2e-5)
        assert close(bn.dscale, g["bn_dscale%d" % step], 2e-5) and close(bn.dshift, g["bn_dshift%d" % step], 2e-5)
        bn.update(0.05)
        assert close(bn.scale, g["bn_scale%d" % (step + 1)], 2e-5) and close(bn.shift, g["bn_shift%d" % (step + 1)], 2e-5)


import pytest


@pytest.mark.parametrize("tag,reverse,carried,cifg", [("lstm", False, False, False), ("lcf", False, True, False), ("lcb", True, False, False),
                                                      ("cifg", False, False, True), ("lstmnp", False, False, False), ("blstmnp", True, False, False),
                                                      ("bmask", True, False, False)])
def test_lstm_family_chains_match_reference_library(oracle, tag, reverse, carried, cifg):
    """The LSTM gate blocks: forward buffer (every gate of every frame), backward buffer, input diff and all gradients (momentum 0, no
    clipping) of T = 5 frames x S = 3 streams.  `lstm` = LstmProjectedStreams (nnet-lstm-projected-streams.h:313-617); `lcf` / `lcb` = the
    two directions of BLstmProjectedStreamsLC (nnet-blstm-projected-streams-lc.h:503-1040): forward in time from a carried state,
    backward in time from zero; `cifg` = LstmCifgProjectedStreams (nnet-lstm-couple-if-projected-streams.h); `lstmnp` / `blstmnp` = Lstm
    and the backward-in-time direction of BLstm (nnet-recurrent-component.cc:235-554, 912-1450; no projection)."""
    g = {k[len(tag) + 1:]: v for k, v in cumatrix_golden.load_blas().items() if k.startswith(tag + "_")}
    Wx, Wr = g["Wx"], g["Wr"]
    ng = 3 if cifg else 4
    Cc, D = Wx.shape[0] // ng, Wx.shape[1]
    R = g["Wrm"].shape[0] if "Wrm" in g else 0
    x, od = g["in"], g["od"]
    S = 3
    T = x.shape[0] // S
    d = oracle.LstmDir(D, Cc, R, cifg, zero=True)
    names = [("w_x", "Wx"), ("w_r", "Wr"), ("bias", "bias"), ("peep_f", "pf"), ("peep_o", "po")]
    if R:
        names.append(("w_rm", "Wrm"))
    if not cifg:
        names.append(("peep_i", "pi"))
    for n, k in names:
        getattr(d, n)[...] = g[k]
    # `bmask`: ragged utterance lengths -- the backward-in-time direction of BLstmProjectedStreams clears the rows of streams that have ended
    lens = g.get("lens")
    buf = d.forward(x, T, S, reverse=reverse, init_state=g["state"] if carried else None, seq_len=lens)
    assert buf.shape == g["fwd_buf"].shape
    assert close(buf[S:(T + 1) * S], g["fwd_buf"][S:(T + 1) * S], 2e-6)
    dbuf, idf = d.backward(od, T, S, buf, reverse=reverse)
    assert close(dbuf[S:(T + 1) * S], g["bwd_buf"][S:(T + 1) * S], 5e-6)
    assert close(idf, g["in_diff"], 5e-6)
    gr = oracle.LstmDir(D, Cc, R, cifg, zero=True)
    d.grads(gr, x, T, S, buf, dbuf, 0.0, 0.0, reverse=reverse)
    for n, k in names:
        assert close(getattr(gr, n), g["g" + ("b" if k == "bias" else k)], 5e-6), n


def test_lstm_two_training_steps_with_momentum_and_clipping_match_reference_library(oracle):
    """`lstm2` (round 4): two training steps of the projected LSTM at C = 64, R = 32, S = 8 streams, T = 6 with momentum 0.9, element-wise
    gradient clipping at 0.5 and the component's Update (learn rate 0.01), every operation issued on the reference's library in the order
    nnet-blstm-projected-streams-lc.h:976-1016, 1085-1098 (= nnet-lstm-projected-streams.h:560-630) issues them.  The oracle's restatement
    (orc_lstm_grads: corr = grad + mmt corr, clip; W -= lr corr) must land on the same parameters after each step."""
    g = {k[6:]: v for k, v in cumatrix_golden.load_blas().items() if k.startswith("lstm2_")}
    Cc, D, R = g["Wx0"].shape[0] // 4, g["Wx0"].shape[1], g["Wr0"].shape[1]
    S, mmt, clip, lr = 8, 0.9, 0.5, 0.01
    T = g["in0"].shape[0] // S
    names = [("w_x", "Wx"), ("w_r", "Wr"), ("bias", "bias"), ("peep_i", "pi"), ("peep_f", "pf"), ("peep_o", "po"), ("w_rm", "Wrm")]
    d, corr = oracle.LstmDir(D, Cc, R, False, zero=True), oracle.LstmDir(D, Cc, R, False, zero=True)
    for n, k in names:
        getattr(d, n)[...] = g[k + "0"]
    clipped = 0
    for step in (0, 1):
        x, od = g["in%d" % step], g["od%d" % step]
        buf = d.forward(x, T, S)
        assert close(d.out_of(buf, T, S), g["out%d" % step], 2e-6)
        dbuf, idf = d.backward(od, T, S, buf)
        assert close(idf, g["in_diff%d" % step], 5e-6)
        d.grads(corr, x, T, S, buf, dbuf, mmt, clip)       # corr = grad + mmt * corr, clipped
        assert close(corr.w_x, g["cWx%d" % step], 5e-6) and close(corr.w_rm, g["cWrm%d" % step], 5e-6)
        clipped += int((np.abs(g["cWx%d" % step]) == np.float32(clip)).sum())
        for n, k in names:
            getattr(d, n)[...] = getattr(d, n) - np.float32(lr) * getattr(corr, n)
            assert close(getattr(d, n), g["%s%d" % (k, step + 1)], 5e-6), (step, n)
    assert clipped > 0      # the clip is active in the fixture


def test_lc_directions_at_full_width_match_reference_library_digest(oracle):
    """`lcff` / `lcfb` (tests/golden/lstm_fullwidth.bin): the two directions of BLstmProjectedStreamsLC (nnet-blstm-projected-streams-lc.h:503-1040)
    at BASELINE cfg3's widths -- C 512, R 256, 512 inputs, S = 32, T = 60 -- as the small `lcf` / `lcb` records pin them: forward in time from a
    CARRIED state, backward in time from zero; every gate of every frame (forward and backward buffers), input diff and all seven gradients,
    on the reference's library, as a digest.  With `lcfull` this pins at full width every function the oracle chain of
    tests/test_cfg3_step_gpu.py is composed of."""
    recs, state = cumatrix_golden.load_fullwidth_directions()
    rng = oracle.GoldenRng(state)
    T, S, D, Cc, R = 60, 32, 512, 512, 256
    stride = cumatrix_golden.DIR_STRIDE
    for tag, reverse, carried in (("lcff", False, True), ("lcfb", True, False)):   # (the generator's order: its state runs on from one to the other)
        g = recs[tag]
        d = oracle.LstmDir(D, Cc, R, False, zero=True)
        d.w_x[...] = rng.fill((4 * Cc, D), -0.02, 0.02); d.w_r[...] = rng.fill((4 * Cc, R), -0.02, 0.02); d.w_rm[...] = rng.fill((R, Cc), -0.02, 0.02)
        d.bias[...] = rng.fill((4 * Cc,), -0.3, 0.3)
        d.peep_i[...] = rng.fill((Cc,), -0.3, 0.3); d.peep_f[...] = rng.fill((Cc,), -0.3, 0.3); d.peep_o[...] = rng.fill((Cc,), -0.3, 0.3)
        x, od = rng.fill((T * S, D), -1.5, 1.5), rng.fill((T * S, R), -1.0, 1.0)
        st = rng.fill((S, d.width), -0.8, 0.8) if carried else None

        def digest_close(a, key, tol):
            pick, sums = cumatrix_golden.digest_of(a, stride)
            return pick.shape == g[key].shape and sums[2] == g[key + "#"][2] and close(pick, g[key], tol) and \
                abs(sums[1] - g[key + "#"][1]) <= 4 * tol * g[key + "#"][1]
        buf = d.forward(x, T, S, reverse=reverse, init_state=st)
        # (the boundary row blocks differ by construction where the generator leaves zeros: compare what both define, the frames 1..T -- the
        #  digest is over the whole buffer, so the oracle's boundary blocks are set to the generator's first)
        ref_like = np.zeros_like(buf)
        ref_like[S:(T + 1) * S] = buf[S:(T + 1) * S]
        if carried:
            ref_like[:S] = st
        assert digest_close(ref_like, "fwd_buf", 1e-5), tag
        dbuf, idf = d.backward(od, T, S, buf, reverse=reverse)
        dref = np.zeros_like(dbuf)
        dref[S:(T + 1) * S] = dbuf[S:(T + 1) * S]
        assert digest_close(dref, "bwd_buf", 2e-5), tag
        assert digest_close(idf, "in_diff", 2e-5), tag
        gr = oracle.LstmDir(D, Cc, R, False, zero=True)
        d.grads(gr, x, T, S, buf, dbuf, 0.0, 0.0, reverse=reverse)
        for n, k in (("w_x", "gWx"), ("w_r", "gWr"), ("w_rm", "gWrm")):
            assert digest_close(getattr(gr, n), k, 5e-5), (tag, n)
        for n, k in (("bias", "gb"), ("peep_i", "gpi"), ("peep_f", "gpf"), ("peep_o", "gpo")):
            assert close(getattr(gr, n), g[k], 5e-5), (tag, n)


def test_lstm_full_width_two_training_steps_match_reference_library_digest(oracle):
    """`lcfull` (tests/golden/lstm_fullwidth.bin): the same two training steps at BASELINE cfg3's widths (C 512, R 256, D 512, S 32, T 60), on the
    reference's library, kept as a digest (every 61st element of a tensor + its sum, sum of squares and size); parameters, inputs and out-diffs
    are replayed from the recorded generator state.  Pins the oracle's LSTM chain -- the checker of tests/test_cfg3_step_gpu.py -- at FULL width."""
    g, state = cumatrix_golden.load_fullwidth()
    T, S, D, Cc, R, mmt, clip, lr = 60, 32, 512, 512, 256, 0.9, 5.0, 0.002
    rng = oracle.GoldenRng(state)
    names = [("w_x", "Wx"), ("w_r", "Wr"), ("bias", "bias"), ("peep_i", "pi"), ("peep_f", "pf"), ("peep_o", "po"), ("w_rm", "Wrm")]
    d, corr = oracle.LstmDir(D, Cc, R, False, zero=True), oracle.LstmDir(D, Cc, R, False, zero=True)
    d.w_x[...] = rng.fill((4 * Cc, D), -0.02, 0.02); d.w_r[...] = rng.fill((4 * Cc, R), -0.02, 0.02); d.w_rm[...] = rng.fill((R, Cc), -0.02, 0.02)
    d.bias[...] = rng.fill((4 * Cc,), -0.3, 0.3)
    d.peep_i[...] = rng.fill((Cc,), -0.3, 0.3); d.peep_f[...] = rng.fill((Cc,), -0.3, 0.3); d.peep_o[...] = rng.fill((Cc,), -0.3, 0.3)

    def digest_close(x, key, tol):
        pick, sums = cumatrix_golden.digest_of(x)
        return pick.shape == g[key].shape and sums[2] == g[key + "#"][2] and close(pick, g[key], tol) and \
            abs(sums[1] - g[key + "#"][1]) <= 4 * tol * g[key + "#"][1]

    clipped = 0
    for step in (0, 1):
        x, od = rng.fill((T * S, D), -1.5, 1.5), rng.fill((T * S, R), -1.0, 1.0)
        buf = d.forward(x, T, S)
        assert digest_close(d.out_of(buf, T, S), "out%d" % step, 1e-5), step
        dbuf, idf = d.backward(od, T, S, buf)
        assert digest_close(idf, "in_diff%d" % step, 1e-5), step
        d.grads(corr, x, T, S, buf, dbuf, mmt, clip)
        assert digest_close(corr.w_x, "cWx%d" % step, 5e-5) and digest_close(corr.w_rm, "cWrm%d" % step, 5e-5), step
        clipped += int((np.abs(g["cWrm%d" % step]) == np.float32(clip)).sum())
        for n, k in names:
            getattr(d, n)[...] = getattr(d, n) - np.float32(lr) * getattr(corr, n)
            key = "%s%d" % (k, step + 1)
            if key + "#" in g:
                assert digest_close(getattr(d, n), key, 2e-5), (step, n)
            else:
                assert close(getattr(d, n), g[key], 2e-5), (step, n)
    assert clipped > 0      # the clip is active in the fixture


def test_dnn_cfg2_full_size_two_training_steps_match_reference_library_digest(oracle):
    """tests/golden/dnn_cfg2_fullsize.bin: BASELINE cfg2 ITSELF (440 -> 5 x 2048 + BatchNormalization + Sigmoid -> 3000, minibatch 1024, learn rate
    0.008, momentum 0) through two training steps on fresh minibatches, every operation issued on the reference's library in the order of
    Nnet::Propagate / Backpropagate with Update behind each component (oracle/ref_dnn_bench.cpp golden mode) -- as a digest.  The oracle's DNN
    chain, the checker of tests/test_nnet_gpu.py::test_cfg2_full_size_matches_oracle and of bench.py's smoke, lands on the same posteriors,
    loss and parameters at FULL size."""
    import ctypes as C
    g, state, stride = cumatrix_golden.load_cfg2_fullsize()
    W, batches = cumatrix_golden.replay_cfg2(oracle.GoldenRng(state))
    d = oracle.lib.orc_dnn_create(440, 2048, 5, 3000, 1, 1024, 1)
    L = oracle.lib.orc_dnn_num_layers(d)
    assert L == 6

    def views(l):
        r, c = C.c_int(), C.c_int()
        w = np.ctypeslib.as_array(oracle.lib.orc_dnn_weight(d, l, C.byref(r), C.byref(c)), shape=(r.value, c.value))
        b = np.ctypeslib.as_array(oracle.lib.orc_dnn_bias(d, l), shape=(r.value,))
        return w, b
    for l in range(L):
        w, b = views(l)
        w[...] = W[l]
        b[...] = 0.0
        if l < L - 1:
            np.ctypeslib.as_array(oracle.lib.orc_dnn_bn_scale(d, l), shape=(2048,))[...] = 1.0
            np.ctypeslib.as_array(oracle.lib.orc_dnn_bn_shift(d, l), shape=(2048,))[...] = 0.0

    def digest_close(x, key, tol):
        pick, sums = cumatrix_golden.digest_of(x, stride)
        return pick.shape == g[key].shape and sums[2] == g[key + "#"][2] and close(pick, g[key], tol) and \
            abs(sums[1] - g[key + "#"][1]) <= 4 * tol * g[key + "#"][1]
    for s, (x, lab) in enumerate(batches):
        loss = oracle.lib.orc_dnn_train_step(d, x, lab, 0.008, 0.0)
        post = np.ctypeslib.as_array(oracle.lib.orc_dnn_output(d), shape=(1024, 3000))
        assert digest_close(post, "post%d" % s, 2e-5), s
        assert abs(loss - g["loss%d" % s]) <= 2e-5 * abs(g["loss%d" % s]), (s, loss, g["loss%d" % s])
        for l in range(L):
            w, b = views(l)
            assert digest_close(w, "W%d_%d" % (l, s + 1), 2e-5), (s, l)
            assert close(b, g["b%d_%d" % (l, s + 1)], 2e-5), (s, l)
            if l < L - 1:
                assert close(np.ctypeslib.as_array(oracle.lib.orc_dnn_bn_scale(d, l), shape=(2048,)), g["sc%d_%d" % (l, s + 1)], 2e-5), (s, l)
                assert close(np.ctypeslib.as_array(oracle.lib.orc_dnn_bn_shift(d, l), shape=(2048,)), g["sh%d_%d" % (l, s + 1)], 2e-5), (s, l)
    oracle.lib.orc_dnn_destroy(d)


def test_gru_chain_matches_reference_library(oracle):
    """nnet-gru-streams.h:238-450: forward buffer (z|r|m|g|h), backward buffer, input diff and the four gradients."""
    g = cumatrix_golden.load_blas()
    H, D = g["gru_Wg"].shape[0], g["gru_Wx"].shape[1]
    x, od = g["gru_in"], g["gru_od"]
    S = 3
    T = x.shape[0] // S
    u = oracle.Gru(D, H, zero=True)
    for n, k in (("w_zrm_x", "gru_Wx"), ("w_zr_h", "gru_Wh"), ("w_m_g", "gru_Wg"), ("bias", "gru_bias")):
        getattr(u, n)[...] = g[k]
    buf = u.forward(x, T, S)
    assert close(buf[S:(T + 1) * S], g["gru_fwd_buf"][S:(T + 1) * S], 2e-6)
    dbuf, idf = u.backward(od, T, S, buf)
    assert close(dbuf[S:(T + 1) * S], g["gru_bwd_buf"][S:(T + 1) * S], 5e-6)
    assert close(idf, g["gru_in_diff"], 5e-6)
    gr = oracle.Gru(D, H, zero=True)
    u.grads(gr, x, T, S, buf, dbuf, 0.0, 0.0)
    for n, k in (("w_zrm_x", "gru_gWx"), ("w_zr_h", "gru_gWh"), ("w_m_g", "gru_gWg"), ("bias", "gru_gb")):
        assert close(getattr(gr, n), g[k], 5e-6), n


def test_rowconvolution_chain_matches_reference_library(oracle):
    """nnet-row-convolution.cc:90-169 (the reference's D x D product per frame, diagonal taken), ragged sequence lengths."""
    g = cumatrix_golden.load_blas()
    w = g["rc_w"]
    D, K = w.shape[0], w.shape[1] - 1
    lens = g["rc_lens"]
    S = len(lens)
    T = g["rc_in"].shape[0] // S
    rc = oracle.RowConv(D, K, np.random.default_rng(0))
    rc.w[...] = w
    out = rc.propagate(g["rc_in"], T, S, lens)
    assert close(out, g["rc_out"], 2e-6)
    idf = rc.backpropagate(g["rc_od"], T, S, lens)
    assert close(idf, g["rc_in_diff"], 2e-6) and close(rc.w_diff, g["rc_w_diff"], 5e-6)


def test_compact_fsmn_chain_matches_reference_library(oracle):
    """nnet-cfsmn-component.h:169-264: output, input diff and the tap gradients (no clipping)."""
    g = cumatrix_golden.load_blas()
    coef = g["fsmn_coef"]
    f = oracle.Fsmn(coef.shape[1], 3, 2, np.random.default_rng(0))
    f.coef[...] = coef
    assert close(f.propagate(g["fsmn_in"]), g["fsmn_out"], 2e-6)
    idf = f.backpropagate(g["fsmn_in"], g["fsmn_od"], 0.0)
    assert close(idf, g["fsmn_in_diff"], 2e-6) and close(f.corr, g["fsmn_corr"], 5e-6)


def test_temporal_components_at_full_size_match_reference_library_digest(oracle):
    """tests/golden/temporal_fullsize.bin: RowConvolution (512, FutureContext 20, T = 800, S = 32 ragged streams) and CompactFsmn (512, 30 + 30 taps,
    T = 800) at the sizes BASELINE cfg5 swaps them in at, on the reference's library -- the D x D product per frame of
    nnet-row-convolution.cc:128-133 and the materialised T x 61 product of nnet-cfsmn-component.h:191-201 included -- as a digest: the oracle's
    direct filters (oracle/aslp_oracle_temporal.c), the checker of tests/test_temporal_gpu.py, land on the same outputs, input diffs and tap
    gradients."""
    rc, fs, t = cumatrix_golden.load_temporal_fullsize(oracle.GoldenRng)
    stride = cumatrix_golden.DIR_STRIDE

    def digest_close(a, g, key, tol):
        pick, sums = cumatrix_golden.digest_of(a, stride)
        return pick.shape == g[key].shape and sums[2] == g[key + "#"][2] and close(pick, g[key], tol) and \
            abs(sums[1] - g[key + "#"][1]) <= 4 * tol * g[key + "#"][1]
    m = oracle.RowConv(512, 20, np.random.default_rng(0))
    m.w[...] = t["rc_w"]
    assert digest_close(m.propagate(t["rc_in"], 800, 32, t["lens"]), rc, "out", 1e-5)
    idf = m.backpropagate(t["rc_od"], 800, 32, t["lens"])
    assert digest_close(idf, rc, "in_diff", 1e-5)
    assert digest_close(m.w_diff, rc, "w_diff", 1e-4)     # sums over ~19,000 frames of products up to 1.5: a few units in the seventh digit
    f = oracle.Fsmn(512, 30, 30, np.random.default_rng(0))
    f.coef[...] = t["fsmn_coef"]
    assert digest_close(f.propagate(t["fsmn_in"]), fs, "out", 1e-5)
    assert digest_close(f.backpropagate(t["fsmn_in"], t["fsmn_od"], 0.0), fs, "in_diff", 1e-5)
    assert digest_close(f.corr, fs, "corr", 5e-5)


def test_gru_at_full_width_matches_reference_library_digest(oracle):
    """`grufull` (tests/golden/temporal_fullsize.bin): GruStreams 512 -> 512 (BASELINE cfg5's swap), S = 32 streams, T = 60, on the reference's library
    (nnet-gru-streams.h:238-450): forward buffer (z|r|m|g|h of every frame), backward buffer, input diff and the four gradients, as a digest."""
    _, _, t = cumatrix_golden.load_temporal_fullsize(oracle.GoldenRng)
    g, stride = t["gru"], cumatrix_golden.DIR_STRIDE
    T, S, D, H = 60, 32, 512, 512

    def digest_close(a, key, tol):
        pick, sums = cumatrix_golden.digest_of(a, stride)
        return pick.shape == g[key].shape and sums[2] == g[key + "#"][2] and close(pick, g[key], tol) and \
            abs(sums[1] - g[key + "#"][1]) <= 4 * tol * g[key + "#"][1]
    u = oracle.Gru(D, H, zero=True)
    for n, k in (("w_zrm_x", "gru_Wx"), ("w_zr_h", "gru_Wh"), ("w_m_g", "gru_Wg"), ("bias", "gru_bias")):
        getattr(u, n)[...] = t[k]
    buf = u.forward(t["gru_in"], T, S)
    ref_like = np.zeros_like(buf)
    ref_like[S:(T + 1) * S] = buf[S:(T + 1) * S]       # (the boundary row blocks are zero in the generator's buffer)
    assert digest_close(ref_like, "fwd_buf", 1e-5)
    dbuf, idf = u.backward(t["gru_od"], T, S, buf)
    dref = np.zeros_like(dbuf)
    dref[S:(T + 1) * S] = dbuf[S:(T + 1) * S]
    assert digest_close(dref, "bwd_buf", 2e-5) and digest_close(idf, "in_diff", 2e-5)
    gr = oracle.Gru(D, H, zero=True)
    u.grads(gr, t["gru_in"], T, S, buf, dbuf, 0.0, 0.0)
    for n, k in (("w_zrm_x", "gWx"), ("w_zr_h", "gWh"), ("w_m_g", "gWg")):
        assert digest_close(getattr(gr, n), k, 5e-5), n
    assert close(gr.bias, g["gb"], 5e-5)


def test_xent_eval_chain_matches_reference_library(oracle):
    """Xent::Eval, nnet-loss.cc:63-156: a zero-weight frame, a frame without a target (masked), a soft posterior; the diff and the five
    sums {frames, correct, loss, entropy, likelihood}."""
    g = cumatrix_golden.load_blas()
    diff, st = oracle.xent_eval(g["xe_fw"], g["xe_y"], g["xe_tgt"])
    assert close(diff, g["xe_diff"], 1e-6)
    got = np.array([st["frames"], st["correct"], st["loss"], st["entropy"], st["likelyhood"]])
    assert np.allclose(got, g["xe_stats"], rtol=2e-6, atol=1e-6), (got, g["xe_stats"])
