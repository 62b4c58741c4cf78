"""GPU parity of the LSTM family and GruStreams (kaldi-aslp_amd/nnet/nnet-recurrent.*, fused gate
kernels in csrc/rnn_cells.hip) against oracle/aslp_oracle_rnn.c: Propagate output, input diff and
updated parameters over several consecutive batches (carried state, momentum, gradient clipping),
driven through the C ABI of include/aslp_nnet.h on model files in the reference's binary format."""
import ctypes as C

import numpy as np
import pytest
import torch

import nnet_io

pytestmark = pytest.mark.gpu
TOL = 1e-4

# marker: (bidir, proj, cifg, lc, cell_dim_token)
FAMILY = {
    "<Lstm>": (False, False, False, False, False),
    "<BLstm>": (True, False, False, False, False),
    "<LstmProjectedStreams>": (False, True, False, False, True),
    "<BLstmProjectedStreams>": (True, True, False, False, True),
    "<BLstmProjectedStreamsLC>": (True, True, False, True, True),
    "<LstmCifgProjectedStreams>": (False, True, True, False, True),
}


def build(oracle, tmp_path, marker, D, Cc, R, clip, seed, scale=0.3):
    bidir, proj, cifg, lc, celltok = FAMILY[marker]
    rng = np.random.default_rng(seed)
    Rr = R if proj else 0
    dirs = [oracle.LstmDir(D, Cc, Rr, cifg, rng, scale=scale) for _ in range(2 if bidir else 1)]
    grads = [oracle.LstmDir(D, Cc, Rr, cifg, zero=True) for _ in dirs]
    out_dim = dirs[0].rec * len(dirs)
    path = tmp_path / "rnn.nnet"
    nnet_io.write_simple_nnet(path, [(marker, D, out_dim, nnet_io.lstm(dirs, clip, Cc if celltok else None))])
    return dirs, grads, out_dim, path


def oracle_step(oracle, marker, dirs, grads, x, od, T, S, state, lens, chunk, lr, mmt, clip):
    """One Propagate + Backpropagate(+Update) of the component in the oracle; returns out, in_diff, new state."""
    bidir, proj, cifg, lc, _ = FAMILY[marker]
    carried = (not bidir) or lc
    f = dirs[0]
    fbuf = f.forward(x, T, S, reverse=False, init_state=state if carried else None)
    outs = [f.out_of(fbuf, T, S)]
    new_state = None
    if carried:
        rb = chunk if lc else T
        new_state = fbuf[rb * S:(rb + 1) * S].copy()
    if bidir:
        b = dirs[1]
        bbuf = b.forward(x, T, S, reverse=True, seq_len=None if lc else lens)
        outs.append(b.out_of(bbuf, T, S))
    out = np.concatenate(outs, axis=1)
    rec = f.rec
    fd, in_diff = f.backward(od[:, :rec], T, S, fbuf, reverse=False)
    if bidir:
        bd, in_diff = b.backward(od[:, rec:], T, S, bbuf, reverse=True, in_diff=in_diff, beta=1.0)
    f.grads(grads[0], x, T, S, fbuf, fd, mmt, clip, reverse=False)
    if bidir:
        b.grads(grads[1], x, T, S, bbuf, bd, mmt, clip, reverse=True)
    for p, g in zip(dirs, grads):
        p.update(g, lr)
    return out, in_diff, new_state


# (12, 16, 8, 5, 70): 70 streams are more than one persistent launch has chains for (32 bidirectional / 64 unidirectional): the engine
# runs the recurrence in stream windows (aslp_lstm_seq.s_begin / s_count), 3 resp. 2 launches per pass
@pytest.mark.parametrize("marker", list(FAMILY))
@pytest.mark.parametrize("dims", [(5, 8, 4, 6, 3), (40, 64, 32, 12, 4), (33, 48, 17, 9, 5), (12, 16, 8, 5, 70)])
def test_lstm_family_train_steps_match_oracle(aslp, oracle, dev, tmp_path, marker, dims):
    D, Cc, R, T, S = dims
    bidir, proj, cifg, lc, _ = FAMILY[marker]
    clip, lr, mmt = 0.5, 0.01, 0.9
    dirs, grads, out_dim, path = build(oracle, tmp_path, marker, D, Cc, R, clip, seed=1)
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt)
    flat = lambda: np.concatenate([d.flat() for d in dirs])
    assert net.NumParams() == flat().size
    assert oracle.rel_err(net.GetParams(), flat()) == 0.0
    rng = np.random.default_rng(7)
    carried = (not bidir) or lc
    chunk = T - 2 if lc else 0
    if lc:
        net.SetChunkSize(chunk)
    state = np.zeros((S, dirs[0].width), np.float32) if carried else None
    for step in range(3):
        x = rng.standard_normal((T * S, D)).astype(np.float32)
        od = rng.standard_normal((T * S, out_dim)).astype(np.float32)
        lens = None
        if carried:
            flags = [1] * S if step == 0 else [int(v) for v in rng.integers(0, 2, S)]
            net.ResetLstmStreams(flags)
            for s, fl in enumerate(flags):
                if fl:
                    state[s] = 0
        else:
            lens = rng.integers(1, T + 1, S).astype(np.int32)
            lens[0] = T
            net.SetSeqLengths(lens)
        out_ref, idf_ref, state = oracle_step(oracle, marker, dirs, grads, x, od, T, S, state, lens, chunk, lr, mmt, clip)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, out_ref) < TOL and oracle.max_err(out, out_ref) < 10 * TOL, ("out", step)
        before = net.GetParams()
        idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf, idf_ref) < TOL and oracle.max_err(idf, idf_ref) < 10 * TOL, ("in_diff", step)
        after = net.GetParams()
        assert oracle.rel_err(after, flat()) < TOL and oracle.max_err(after, flat()) < 10 * TOL, ("params", step)
        # the gradient the engine applied, tensor by tensor (W_x, W_r, bias, the peepholes, W_rm per direction), against the oracle's *_corr
        oracle.assert_applied_gradients(before, after, lr, [t for di, g in enumerate(grads) for t in g.named_tensors("dir%d." % di)], TOL, (marker, step))


@pytest.mark.parametrize("marker", list(FAMILY))
def test_lstm_family_at_cell_dim_512(aslp, oracle, dev, tmp_path, marker):
    """Every member of the family at the BASELINE cell size (C = 512, R = 256, S = 32 streams): the fused step kernels' full
    K-split / tile paths, not the ragged small shapes above.  Two batches, so the carried state and momentum are in play."""
    D, Cc, R, T, S = 64, 512, 256, 12, 32
    bidir, proj, cifg, lc, _ = FAMILY[marker]
    clip, lr, mmt = 5.0, 1e-3, 0.9
    dirs, grads, out_dim, path = build(oracle, tmp_path, marker, D, Cc, R, clip, seed=21, scale=0.05)
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt)
    flat = lambda: np.concatenate([d.flat() for d in dirs])
    rng = np.random.default_rng(17)
    carried = (not bidir) or lc
    chunk = T - 4 if lc else 0
    if lc:
        net.SetChunkSize(chunk)
    state = np.zeros((S, dirs[0].width), np.float32) if carried else None
    for step in range(2):
        x = rng.standard_normal((T * S, D)).astype(np.float32)
        od = (rng.standard_normal((T * S, out_dim)) * 0.1).astype(np.float32)
        lens = None
        if carried:
            net.ResetLstmStreams([1] * S if step == 0 else [0] * S)
        else:
            lens = rng.integers(1, T + 1, S).astype(np.int32)
            lens[0] = T
            net.SetSeqLengths(lens)
        out_ref, idf_ref, state = oracle_step(oracle, marker, dirs, grads, x, od, T, S, state, lens, chunk, lr, mmt, clip)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, out_ref) < TOL and oracle.max_err(out, out_ref) < 10 * TOL, ("out", step)
        before = net.GetParams()
        idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf, idf_ref) < TOL and oracle.max_err(idf, idf_ref) < 10 * TOL, ("in_diff", step)
        after = net.GetParams()
        assert oracle.rel_err(after, flat()) < TOL and oracle.max_err(after, flat()) < 10 * TOL, ("params", step)
        # the gradient the engine applied, tensor by tensor (W_x, W_r, bias, the peepholes, W_rm per direction), against the oracle's *_corr
        oracle.assert_applied_gradients(before, after, lr, [t for di, g in enumerate(grads) for t in g.named_tensors("dir%d." % di)], TOL, (marker, step))


def test_lstm_forward_without_reset_is_per_utterance(aslp, oracle, dev, tmp_path):
    """nnet-forward mode: no ResetLstmStreams call -> one stream, state zeroed every Propagate
    (nnet-lstm-projected-streams.h:316-323)."""
    D, Cc, R, T = 6, 8, 4, 7
    dirs, _, out_dim, path = build(oracle, tmp_path, "<LstmProjectedStreams>", D, Cc, R, 0.0, seed=3)
    net = aslp.Nnet.Read(path)
    rng = np.random.default_rng(1)
    for _ in range(2):
        x = rng.standard_normal((T, D)).astype(np.float32)
        ref = dirs[0].out_of(dirs[0].forward(x, T, 1), T, 1)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, ref) < TOL


def test_lc_blstm_baseline_shape(aslp, oracle, dev, tmp_path):
    """BASELINE.json config 3 geometry: C=512, R=256 per direction, chunk 40 + 20 right context,
    8 streams; two consecutive chunks with the forward state carried from row `chunk`."""
    D, Cc, R, S, chunk, right = 120, 512, 256, 8, 40, 20
    T = chunk + right
    marker = "<BLstmProjectedStreamsLC>"
    clip, lr, mmt = 5.0, 1e-3, 0.9   # (1e-4 would leave the applied gradient, read back as (W_before - W_after) / lr, under the weights' fp32 rounding)
    dirs, grads, out_dim, path = build(oracle, tmp_path, marker, D, Cc, R, clip, seed=9, scale=0.05)
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt)
    net.SetChunkSize(chunk)
    rng = np.random.default_rng(3)
    state = np.zeros((S, dirs[0].width), np.float32)
    for step in range(2):
        x = rng.standard_normal((T * S, D)).astype(np.float32)
        od = (rng.standard_normal((T * S, out_dim)) * 0.1).astype(np.float32)
        net.ResetLstmStreams([1] * S if step == 0 else [0] * S)
        out_ref, idf_ref, state = oracle_step(oracle, marker, dirs, grads, x, od, T, S, state, None, chunk, lr, mmt, clip)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, out_ref) < TOL and oracle.max_err(out, out_ref) < 10 * TOL, ("out", step)
        before = net.GetParams()
        idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf, idf_ref) < TOL and oracle.max_err(idf, idf_ref) < 10 * TOL, ("in_diff", step)
        after = net.GetParams()
        assert oracle.rel_err(after, np.concatenate([d.flat() for d in dirs])) < TOL, ("params", step)
        oracle.assert_applied_gradients(before, after, lr, [t for di, g in enumerate(grads) for t in g.named_tensors("dir%d." % di)], TOL, step)


# (512, 512, 60, 32): BASELINE cfg5's GruStreams swap at full size (H = 512, S = 32 streams, T = 60).
# Three execution paths: the persistent kernels (csrc/rnn_persistent.hip: H % 4 == 0, H <= 512, S <= 64 -- one launch per pass),
# in stream windows of 64 where there are more streams (here S = 70: two launches per pass), four fused launches per timestep
# (csrc/gru_fused.hip: H % 4 == 0 and H > 512, here 516), GEMM + cell kernels per timestep (the rest).
@pytest.mark.parametrize("dims", [(5, 6, 6, 3), (40, 64, 10, 4), (33, 50, 7, 5), (48, 128, 6, 40), (24, 36, 5, 33), (512, 512, 60, 32), (16, 64, 4, 70),
                                  (20, 132, 9, 11), (8, 516, 3, 4)])
def test_gru_train_steps_match_oracle(aslp, oracle, dev, tmp_path, dims):
    D, H, T, S = dims
    from kaldi_aslp_amd._lib import GruSeq
    q = GruSeq(None, None, None, None, 0, 0, (5 * H + 15) & ~15, T, S, H, 0, min(S, 64) if S > 64 else 0)
    for backward in (0, 1):   # the shapes meant for the persistent kernels do run on them (not on a silent fallback)
        assert bool(aslp.lib.aslp_gru_seq_supported(C.byref(q), backward)) == (H % 4 == 0 and H <= 512), (dims, backward)
    clip, lr, mmt = 0.5, 0.01, 0.9
    rng = np.random.default_rng(2)
    p = oracle.Gru(D, H, rng, scale=0.3 if H < 256 else 0.05)
    g = oracle.Gru(D, H, zero=True)
    path = tmp_path / "gru.nnet"
    nnet_io.write_simple_nnet(path, [("<GruStreams>", D, H, nnet_io.gru(p, clip))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt)
    assert oracle.rel_err(net.GetParams(), p.flat()) == 0.0
    state = np.zeros((S, 5 * H), np.float32)
    for step in range(3):
        x = rng.standard_normal((T * S, D)).astype(np.float32)
        od = rng.standard_normal((T * S, H)).astype(np.float32)
        flags = [1] * S if step == 0 else [int(v) for v in rng.integers(0, 2, S)]
        net.ResetLstmStreams(flags)
        for s, fl in enumerate(flags):
            if fl:
                state[s] = 0
        buf = p.forward(x, T, S, init_state=state)
        out_ref = p.out_of(buf, T, S)
        state = buf[T * S:(T + 1) * S].copy()
        dbuf, idf_ref = p.backward(od, T, S, buf)
        p.grads(g, x, T, S, buf, dbuf, mmt, clip)
        p.update(g, lr)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, out_ref) < TOL, ("out", step)
        idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf, idf_ref) < TOL, ("in_diff", step)
        assert oracle.rel_err(net.GetParams(), p.flat()) < TOL, ("params", step)


def test_recurrent_write_read_roundtrip(aslp, oracle, dev, tmp_path):
    for marker in FAMILY:
        dirs, _, out_dim, path = build(oracle, tmp_path, marker, 7, 6, 4, 1.5, seed=4)
        net = aslp.Nnet.Read(path)
        p2 = tmp_path / "copy.nnet"
        net.Write(p2, binary=True)
        assert open(path, "rb").read() == open(p2, "rb").read(), marker
        p3 = tmp_path / "copy.txt"
        net.Write(p3, binary=False)
        net3 = aslp.Nnet.Read(p3)
        assert oracle.rel_err(net3.GetParams(), net.GetParams()) < 1e-6


def test_recurrent_init_from_proto(aslp, dev):
    """<NnetProto> lines as aslp_scripts/aslp_nnet/make_*_proto.py emit them; checks dims, parameter
    counts and the uniform [-scale, scale] initialisation (e.g. lc.h:75-88, 96-150)."""
    proto = """<NnetProto>
<BLstmProjectedStreamsLC> <InputDim> 20 <OutputDim> 16 <CellDim> 12 <ParamScale> 0.05 <ClipGradient> 5.0
<LstmProjectedStreams> <InputDim> 16 <OutputDim> 8 <CellDim> 10 <ParamScale> 0.05 <ClipGradient> 5.0
<LstmCifgProjectedStreams> <InputDim> 8 <OutputDim> 8 <CellDim> 10 <ClipGradient> 5.0
<BLstmProjectedStreams> <InputDim> 8 <OutputDim> 12 <CellDim> 9
<Lstm> <InputDim> 12 <OutputDim> 7 <ParamScale> 0.05
<BLstm> <InputDim> 7 <OutputDim> 10 <ClipGradient> 1.0
<GruStreams> <InputDim> 10 <OutputDim> 6 <ParamScale> 0.05
<AffineTransform> <InputDim> 6 <OutputDim> 4 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04
<Softmax> <InputDim> 4 <OutputDim> 4
</NnetProto>
"""
    net = aslp.Nnet.Init(proto, seed=777)

    def n_lstm(D, Cc, R, G=4, ndir=1):
        rec = R if R else Cc
        return ndir * (G * Cc * D + G * Cc * rec + G * Cc + (G - 1) * Cc + R * Cc)

    want = (n_lstm(20, 12, 8, ndir=2) + n_lstm(16, 10, 8) + n_lstm(8, 10, 8, G=3) + n_lstm(8, 9, 6, ndir=2) +
            n_lstm(12, 7, 0) + n_lstm(7, 5, 0, ndir=2) + (3 * 6 * 10 + 2 * 6 * 6 + 6 * 6 + 3 * 6) + 6 * 4 + 4)
    assert net.NumParams() == want
    p = net.GetParams()
    first = p[:n_lstm(20, 12, 8, ndir=2)]
    assert np.abs(first).max() <= 0.05 and np.abs(first).max() > 0.04
    # one training pass runs end to end
    net.SetTrainOptions(learn_rate=0.01, momentum=0.5)
    S, T = 2, 5
    net.ResetLstmStreams([1] * S)
    net.SetSeqLengths([T] * S)
    net.SetChunkSize(3)
    x = torch.randn(T * S, 20, device=dev)
    out = net.Propagate(x)
    assert out.shape == (T * S, 4) and torch.isfinite(out).all()
    assert torch.allclose(out.sum(1), torch.ones(T * S, device=dev), atol=1e-5)
    net.Backpropagate(torch.randn(T * S, 4, device=dev) * 0.1)
    assert np.isfinite(net.GetParams()).all() and not np.array_equal(net.GetParams(), p)


def test_folded_update_equals_separate_update(aslp, dev):
    """With layer fusion on, the executor announces Update to the recurrent components and their SGD step rides in the gradient
    kernels (GEMM epilogue, rnn_vec_grads); with it off, Update is a launch sequence of its own (lc.h:1085-1110,
    nnet-gru-streams.h:457-466).  Same arithmetic either way: parameters agree to the last bit over several steps."""
    proto = """<NnetProto>
<BLstmProjectedStreamsLC> <InputDim> 20 <OutputDim> 16 <CellDim> 12 <ParamScale> 0.2 <ClipGradient> 0.05
<LstmProjectedStreams> <InputDim> 16 <OutputDim> 8 <CellDim> 12 <ParamScale> 0.2 <ClipGradient> 0.05
<LstmCifgProjectedStreams> <InputDim> 8 <OutputDim> 8 <CellDim> 8 <ParamScale> 0.2 <ClipGradient> 0.05
<GruStreams> <InputDim> 8 <OutputDim> 8 <ParamScale> 0.2 <ClipGradient> 0.05
<AffineTransform> <InputDim> 8 <OutputDim> 5 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.3
<Softmax> <InputDim> 5 <OutputDim> 5
</NnetProto>
"""
    nets = [aslp.Nnet.Init(proto, seed=5), aslp.Nnet.Init(proto, seed=5)]
    nets[1].SetLayerFusion(False)
    assert np.array_equal(nets[0].GetParams(), nets[1].GetParams())
    T, S, chunk = 9, 4, 6
    rng = np.random.default_rng(3)
    xents = [aslp.Xent(), aslp.Xent()]
    for step in range(4):
        x = torch.from_numpy(rng.standard_normal((T * S, 20)).astype(np.float32)).to(dev)
        lab = torch.from_numpy(rng.integers(0, 5, T * S).astype(np.int32)).to(dev)
        flags = [1] * S if step == 0 else [int(v) for v in rng.integers(0, 2, S)]
        for net, xe in zip(nets, xents):
            net.SetTrainOptions(learn_rate=0.05, momentum=0.9)
            net.SetChunkSize(chunk)
            net.ResetLstmStreams(flags)
            y = net.Propagate(x)
            diff = torch.empty_like(y)
            xe.Eval(torch.ones(T * S, device=dev), y, diff, labels=lab)
            net.Backpropagate(diff)
        a, b = nets[0].GetParams(), nets[1].GetParams()
        assert np.array_equal(a, b), (step, float(np.abs(a - b).max()))
    assert np.abs(nets[0].GetParams() - aslp.Nnet.Init(proto, seed=5).GetParams()).max() > 1e-3  # and they did move


def test_gru_persistent_long_sequence_and_reset_pattern(aslp, oracle, dev, tmp_path):
    """The persistent GRU kernels over a long sequence (T = 300: 600 hand-off rounds per pass, two per timestep) with H = 128,
    S = 16 (two chains, one of them with idle stream slots none) and a carried state: a missed or stale hand-off piece anywhere
    in the chain would show up in the last frames.  Also checks that a second call reuses the runtime state cleanly."""
    D, H, T, S = 24, 128, 300, 16
    clip, lr, mmt = 0.5, 0.005, 0.9
    rng = np.random.default_rng(9)
    p = oracle.Gru(D, H, rng, scale=0.08)
    g = oracle.Gru(D, H, zero=True)
    path = tmp_path / "gru.nnet"
    nnet_io.write_simple_nnet(path, [("<GruStreams>", D, H, nnet_io.gru(p, clip))])
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt)
    state = np.zeros((S, 5 * H), np.float32)
    for step in range(2):
        x = rng.standard_normal((T * S, D)).astype(np.float32)
        od = rng.standard_normal((T * S, H)).astype(np.float32) * 0.1
        net.ResetLstmStreams([1] * S if step == 0 else [0] * S)
        buf = p.forward(x, T, S, init_state=state)
        out_ref = p.out_of(buf, T, S)
        state = buf[T * S:(T + 1) * S].copy()
        dbuf, idf_ref = p.backward(od, T, S, buf)
        p.grads(g, x, T, S, buf, dbuf, mmt, clip)
        p.update(g, lr)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, out_ref) < TOL, ("out", step)
        assert oracle.rel_err(out[-S:], out_ref[-S:]) < TOL, ("last frame", step)
        idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf, idf_ref) < TOL, ("in_diff", step)
        assert oracle.rel_err(idf[:S], idf_ref[:S]) < 5 * TOL, ("first frame's in_diff", step)
        assert oracle.rel_err(net.GetParams(), p.flat()) < TOL, ("params", step)


def test_recurrent_products_against_float64(aslp, oracle, dev, tmp_path):
    """How far is each way of multiplying in the persistent recurrence from the truth?  One LstmProjectedStreams layer at cfg3's size (C = 512,
    R = 256, T = 60, S = 32) against the same recurrence in float64 (numpy; nnet-lstm-projected-streams.h:286-352 written out): the default kernels
    (v_mfma_f32_16x16x32_f16, every fp32 operand as two fp16 pieces), the fp32-instruction kernels (aslp_lstm_split16(0)) and the oracle's fp32 C
    code.  All three sit at fp32 rounding level; the two-piece products must not be further from float64 than the fp32 instruction's."""
    D, Cc, R, T, S = 40, 512, 256, 60, 32
    marker = "<LstmProjectedStreams>"
    dirs, _, out_dim, path = build(oracle, tmp_path, marker, D, Cc, R, 5.0, seed=21, scale=0.05)
    d = dirs[0]
    rng = np.random.default_rng(8)
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    f8 = np.float64
    wx, wr, b, wrm = d.w_x.astype(f8), d.w_r.astype(f8), d.bias.astype(f8), d.w_rm.astype(f8)
    pi, pf, po = d.peep_i.astype(f8), d.peep_f.astype(f8), d.peep_o.astype(f8)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    r, c, ref = np.zeros((S, R)), np.zeros((S, Cc)), []
    for t in range(T):
        pre = x[t * S:(t + 1) * S].astype(f8) @ wx.T + b + r @ wr.T
        g, i, f = np.tanh(pre[:, :Cc]), sig(pre[:, Cc:2 * Cc] + c * pi), sig(pre[:, 2 * Cc:3 * Cc] + c * pf)
        c = np.clip(g * i + c * f, -50.0, 50.0)
        o = sig(pre[:, 3 * Cc:] + c * po)
        r = (o * np.tanh(c)) @ wrm.T
        ref.append(r)
    ref = np.concatenate(ref, axis=0)
    errs = {}
    try:
        for name, on in (("fp32_instruction", 0), ("two_piece_fp16", 1)):
            aslp.lib.aslp_lstm_split16(on)
            net = aslp.Nnet.Read(path)
            net.ResetLstmStreams([1] * S)
            out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy().astype(f8)
            errs[name] = float(np.linalg.norm(out - ref) / np.linalg.norm(ref))
    finally:
        aslp.lib.aslp_lstm_split16(-1)
    obuf = d.forward(x, T, S, reverse=False, init_state=np.zeros((S, d.width), np.float32))
    errs["oracle_fp32_c"] = float(np.linalg.norm(d.out_of(obuf, T, S).astype(f8) - ref) / np.linalg.norm(ref))
    print("relative error against float64:", errs)
    assert max(errs.values()) < 5e-6, errs
    assert errs["two_piece_fp16"] <= 1.5 * errs["fp32_instruction"] + 2e-7, errs


def test_persistent_recurrences_from_two_threads_and_streams(aslp, dev):
    """Two host threads, each with its own stream and its own LC-BLSTM net, train at the same time.  A persistent recurrence needs every CU, so
    the launches of a process are chained: a launch that goes to another stream than its predecessor waits for it, through an event recorded
    on the predecessor's stream at that moment (rnn_persistent.hip chain_behind_last_launch; no event behind every launch).  Each thread
    must get, bit for bit, what the same net gives when it trains alone."""
    import threading
    S, T, D, steps = 16, 8, 64, 6
    proto = "<NnetProto>\n<BLstmProjectedStreamsLC> <InputDim> %d <OutputDim> 128 <CellDim> 128 <ParamScale> 0.05 <ClipGradient> 5.0\n</NnetProto>\n" % D

    def make(seed):   # (nets are initialised from a process-wide generator: one at a time, on the calling thread)
        net = aslp.Nnet.Init(proto, seed=seed)
        net.SetTrainOptions(learn_rate=1e-3, momentum=0.9)
        net.SetChunkSize(6)
        g = torch.Generator(device="cpu").manual_seed(seed)
        data = [(torch.randn(T * S, D, generator=g).to(dev), (torch.randn(T * S, 128, generator=g) * 0.1).to(dev)) for _ in range(steps)]
        torch.cuda.synchronize()
        return net, data

    def train(net, data, stream):
        with torch.cuda.stream(stream):
            aslp.ops.use_torch_stream()
            outs = []
            for step, (x, od) in enumerate(data):
                net.ResetLstmStreams([1] * S if step == 0 else [0] * S)
                outs.append(net.Propagate(x).cpu().numpy())
                outs.append(net.Backpropagate(od, want_in_diff=True).cpu().numpy())
            outs.append(np.asarray(net.GetParams(), np.float32))
            stream.synchronize()
        return np.concatenate([o.ravel() for o in outs])

    try:
        alone = [train(*make(11 + i), torch.cuda.Stream()) for i in range(2)]
        jobs = [make(11 + i) for i in range(2)]
        got, errs = [None, None], []

        def worker(i):
            try:
                got[i] = train(jobs[i][0], jobs[i][1], torch.cuda.Stream())
            except Exception as e:   # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        for i in range(2):
            assert np.isfinite(got[i]).all() and np.array_equal(got[i], alone[i]), i
    finally:
        aslp.ops.use_torch_stream()
