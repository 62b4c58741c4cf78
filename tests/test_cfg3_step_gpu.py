"""BASELINE.json cfg3 at its full layer sizes: 4 x BLstmProjectedStreamsLC (C = 512, R = 256, in 40) + AffineTransform
512 -> 128 + WarpCtc, S = 32 whole-utterance streams: the loop body of aslp-nnet-train-warp-ctc-streams.cc:158-223
(SetSeqLengths -> Propagate -> WarpCtc::Eval -> Backpropagate) preceded by ResetLstmStreams(all 1), two steps with momentum
0.9, against the oracle chain
(LSTM oracle per layer and direction + affine oracle + the reference-pinned CTC restatement + wrapper logic).  T is kept at
60 frames (ragged 30..60) so the CPU chain finishes in seconds; every kernel runs at the BASELINE widths.

Why the ResetLstmStreams call: Nnet::SetSeqLengths does not reach BLstmProjectedStreamsLC in the reference (nnet-nnet.cc:498-530
lists seven component types, the LC one is not among them), so under the Warp-CTC tool alone the component never learns the
number of streams and runs its "nnet-forward" branch -- ONE stream of T*S frames (lc.h:505-512).  The engine reproduces that
(it is what the log line "Running nnet-forward with per-utterance LSTM-state reset" says); a meaningful multi-stream CTC
step tells the component its streams the way aslp-nnet-train-blstm-streams-lc.cc does, by ResetLstmStreams."""
import ctypes as C

import numpy as np
import pytest
import torch

import nnet_io
from test_warpctc_gpu import FilterState, oracle_wrapper

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_cfg3_lcblstm_warpctc_two_steps_match_oracle_chain(aslp, oracle, dev, tmp_path):
    lcblstm_warpctc_steps_against_oracle(aslp, oracle, dev, tmp_path, T=60, NL=4, steps=2, seed=33)


def test_cfg3_whole_utterances_of_400_frames_match_oracle_chain(aslp, oracle, dev, tmp_path):
    """the whole-utterance Warp-CTC step (bench.py `whole_utterance_warpctc`, T <= 800) was checked against the oracle
    at T = 60 only.  Here T = 400 (ragged 200 ... 400, L = T / 4 labels), S = 32 streams at the BASELINE widths, on ONE
    BLstmProjectedStreamsLC layer + AffineTransform + WarpCtc so that the CPU chain stays at seconds: 400 dependent timesteps in either
    direction of time through the persistent recurrences, the 12800-row layer products from planes, the lattice kernels at T = 400."""
    lcblstm_warpctc_steps_against_oracle(aslp, oracle, dev, tmp_path, T=400, NL=1, steps=1, seed=47)


def lcblstm_warpctc_steps_against_oracle(aslp, oracle, dev, tmp_path, T, NL, steps, seed):
    D, Cc, R, A, S = 40, 512, 256, 128, 32
    clip, lr, mmt = 5.0, 1e-3, 0.9   # (at 1e-4 the applied gradient, read back as (W_before - W_after) / lr, drowns in the weights' fp32 rounding)
    rng = np.random.default_rng(seed)
    layers, params, grads = [], [], []
    d = D
    for l in range(NL):
        dirs = [oracle.LstmDir(d, Cc, R, False, rng, scale=0.05) for _ in range(2)]
        params.append(dirs)
        grads.append([oracle.LstmDir(d, Cc, R, False, zero=True) for _ in range(2)])
        layers.append(("<BLstmProjectedStreamsLC>", d, 2 * R, nnet_io.lstm(dirs, clip, Cc)))
        d = 2 * R
    W = (rng.standard_normal((A, d)) * 0.04).astype(np.float32)
    b = np.zeros(A, np.float32)
    Wc, bc = np.zeros_like(W), np.zeros_like(b)
    layers.append(("<AffineTransform>", d, A, nnet_io.affine(W, b)))
    path = tmp_path / "cfg3.nnet"
    nnet_io.write_simple_nnet(path, layers)
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt)
    ctc = aslp.WarpCtc()
    st = FilterState(0, 0, 0, 0, 0, 0, 500, 0, 0)
    o = oracle.AffineOpts(lr, mmt, 0.0, 0.0, 1.0, 1.0, 0.0)

    def flat():
        return np.concatenate([p.flat() for dirs in params for p in dirs] + [W.ravel(), b])

    assert oracle.rel_err(net.GetParams(), flat()) == 0.0
    for step in range(steps):
        in_len = rng.integers(T // 2, T + 1, S).astype(np.int32)
        in_len[0] = T
        labels = [[int(v) for v in rng.integers(1, A, max(1, int(t) // 4))] for t in in_len]
        x = rng.standard_normal((T * S, D)).astype(np.float32)
        # ---- oracle chain: forward (whole-utterance batches: the carried state is zeroed by SetSeqLengths, lc.h:498-501)
        h, bufs, ins = x, [], []
        for dirs in params:
            f, bk = dirs
            fbuf = f.forward(h, T, S, reverse=False, init_state=np.zeros((S, f.width), np.float32))
            bbuf = bk.forward(h, T, S, reverse=True, seq_len=None)
            ins.append(h)
            bufs.append((fbuf, bbuf))
            h = np.concatenate([f.out_of(fbuf, T, S), bk.out_of(bbuf, T, S)], axis=1)
        y = np.empty((T * S, A), np.float32)
        oracle.lib.orc_affine_propagate(y, A, h, d, T * S, W, d, b, d, A)
        costs_ref, diff, _ = oracle_wrapper(oracle, st, y, labels, in_len, A, S, T)
        diff = np.ascontiguousarray(diff, np.float32)
        dh = np.empty((T * S, d), np.float32)
        oracle.lib.orc_affine_backpropagate(dh, d, diff, A, T * S, W, d, d, A)
        oracle.lib.orc_affine_update(W, d, b, Wc, d, bc, h, d, diff, A, T * S, d, A, C.byref(o))
        for l in range(NL - 1, -1, -1):
            f, bk = params[l]
            fbuf, bbuf = bufs[l]
            fd, idf = f.backward(np.ascontiguousarray(dh[:, :R]), T, S, fbuf, reverse=False)
            bd, idf = bk.backward(np.ascontiguousarray(dh[:, R:]), T, S, bbuf, reverse=True, in_diff=idf, beta=1.0)
            f.grads(grads[l][0], ins[l], T, S, fbuf, fd, mmt, clip, reverse=False)
            bk.grads(grads[l][1], ins[l], T, S, bbuf, bd, mmt, clip, reverse=True)
            f.update(grads[l][0], lr)
            bk.update(grads[l][1], lr)
            dh = idf
        # ---- engine
        net.ResetLstmStreams([1] * S)
        before = net.GetParams()
        net.TrainStepWarpCtc(ctc, torch.from_numpy(x).to(dev), in_len, labels)
        out = net.ComponentOutput(net.NumComponents() - 1, T * S, A)
        assert oracle.rel_err(out, y) < TOL and oracle.max_err(out, y) < 10 * TOL, ("activations", step)
        got, want = net.GetParams(), flat()
        assert oracle.rel_err(got, want) < TOL and oracle.max_err(got, want) < 10 * TOL, ("params", step)
        # the gradients the engine applied, tensor by tensor (4 layers x 2 directions x {W_x, W_r, bias, 3 peepholes, W_rm}, W and b of the
        # output layer), against the oracle's momentum-carrying clipped *_corr tensors
        ref = [t for li in range(NL) for di in range(2) for t in grads[li][di].named_tensors("layer%d.dir%d." % (li, di))] + [("W", Wc), ("b", bc)]
        oracle.assert_applied_gradients(before, got, lr, ref, TOL, step)
        # per tensor group too: a whole-vector norm would hide a small tensor (peepholes, biases) that is wrong
        off = 0
        for li, dirs in enumerate(params):
            for di, p in enumerate(dirs):
                n = p.flat().size
                assert oracle.rel_err(got[off:off + n], want[off:off + n]) < TOL, ("layer", li, "dir", di, step)
                off += n
        stt = ctc.GetStats()
        assert abs(stt["obj"] - st.obj) <= 1e-4 * abs(st.obj), step
    assert ctc.GetStats()["sequences"] == steps * S


def test_cfg3_chunked_three_chunks_with_mixed_stream_resets(aslp, oracle, dev, tmp_path):
    """The chunked (latency-controlled) use of the same 4-layer net as aslp-nnet-train-blstm-streams-lc.cc drives it: chunk 40 + 20 frames
    of right context, S = 32 streams, THREE consecutive chunks with mixed ResetLstmStreams flags -- streams that end are restarted from
    zero while their neighbours carry the forward direction's state from row chunk * S of the previous chunk (lc.h:477-483, 629); the
    backward direction always starts from zero at the end of the right context.  Output, parameters and the applied gradients of every
    tensor against the oracle chain."""
    D, Cc, R, A, S, NL, chunk, right = 40, 512, 256, 128, 32, 4, 40, 20
    T = chunk + right
    clip, lr, mmt = 5.0, 1e-3, 0.9
    rng = np.random.default_rng(71)
    layers, params, grads = [], [], []
    d = D
    for l in range(NL):
        dirs = [oracle.LstmDir(d, Cc, R, False, rng, scale=0.05) for _ in range(2)]
        params.append(dirs)
        grads.append([oracle.LstmDir(d, Cc, R, False, zero=True) for _ in range(2)])
        layers.append(("<BLstmProjectedStreamsLC>", d, 2 * R, nnet_io.lstm(dirs, clip, Cc)))
        d = 2 * R
    W = (rng.standard_normal((A, d)) * 0.04).astype(np.float32)
    b = np.zeros(A, np.float32)
    Wc, bc = np.zeros_like(W), np.zeros_like(b)
    layers.append(("<AffineTransform>", d, A, nnet_io.affine(W, b)))
    path = tmp_path / "cfg3_chunked.nnet"
    nnet_io.write_simple_nnet(path, layers)
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=mmt)
    net.SetChunkSize(chunk)
    o = oracle.AffineOpts(lr, mmt, 0.0, 0.0, 1.0, 1.0, 0.0)
    state = [np.zeros((S, params[l][0].width), np.float32) for l in range(NL)]

    def flat():
        return np.concatenate([p.flat() for dirs in params for p in dirs] + [W.ravel(), b])

    for step in range(3):
        flags = [1] * S if step == 0 else [int(v) for v in rng.integers(0, 2, S)]
        if step > 0:
            flags[0], flags[1] = 0, 1   # both kinds present whatever the draw
        x = rng.standard_normal((T * S, D)).astype(np.float32)
        od = (rng.standard_normal((T * S, A)) * 0.05).astype(np.float32)
        od[chunk * S:] = 0.0            # right-context frames carry no loss (the tool's frame mask)
        # ---- oracle chain
        h, bufs, ins = x, [], []
        for l, dirs in enumerate(params):
            f, bk = dirs
            for s_, fl in enumerate(flags):
                if fl:
                    state[l][s_] = 0
            fbuf = f.forward(h, T, S, reverse=False, init_state=state[l])
            state[l] = fbuf[chunk * S:(chunk + 1) * S].copy()
            bbuf = bk.forward(h, T, S, reverse=True, seq_len=None)
            ins.append(h)
            bufs.append((fbuf, bbuf))
            h = np.concatenate([f.out_of(fbuf, T, S), bk.out_of(bbuf, T, S)], axis=1)
        y = np.empty((T * S, A), np.float32)
        oracle.lib.orc_affine_propagate(y, A, h, d, T * S, W, d, b, d, A)
        dh = np.empty((T * S, d), np.float32)
        oracle.lib.orc_affine_backpropagate(dh, d, od, A, T * S, W, d, d, A)
        oracle.lib.orc_affine_update(W, d, b, Wc, d, bc, h, d, od, A, T * S, d, A, C.byref(o))
        for l in range(NL - 1, -1, -1):
            f, bk = params[l]
            fbuf, bbuf = bufs[l]
            fd, idf = f.backward(np.ascontiguousarray(dh[:, :R]), T, S, fbuf, reverse=False)
            bd, idf = bk.backward(np.ascontiguousarray(dh[:, R:]), T, S, bbuf, reverse=True, in_diff=idf, beta=1.0)
            f.grads(grads[l][0], ins[l], T, S, fbuf, fd, mmt, clip, reverse=False)
            bk.grads(grads[l][1], ins[l], T, S, bbuf, bd, mmt, clip, reverse=True)
            f.update(grads[l][0], lr)
            bk.update(grads[l][1], lr)
            dh = idf
        # ---- engine
        net.ResetLstmStreams(flags)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, y) < TOL and oracle.max_err(out, y) < 10 * TOL, ("output", step)
        before = net.GetParams()
        idf_e = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf_e, dh) < TOL and oracle.max_err(idf_e, dh) < 10 * TOL, ("in_diff", step)
        got = net.GetParams()
        assert oracle.rel_err(got, flat()) < TOL and oracle.max_err(got, flat()) < 10 * TOL, ("params", step)
        ref = [t for li in range(NL) for di in range(2) for t in grads[li][di].named_tensors("layer%d.dir%d." % (li, di))] + [("W", Wc), ("b", bc)]
        oracle.assert_applied_gradients(before, got, lr, ref, TOL, step)
