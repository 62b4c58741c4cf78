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
    D, Cc, R, A, T, S, NL = 40, 512, 256, 128, 60, 32, 4
    clip, lr, mmt = 5.0, 1e-4, 0.9
    rng = np.random.default_rng(33)
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
    for step in range(2):
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
        net.TrainStepWarpCtc(ctc, torch.from_numpy(x).to(dev), in_len, labels)
        out = net.ComponentOutput(net.NumComponents() - 1, T * S, A)
        assert oracle.rel_err(out, y) < TOL, ("activations", step)
        got, want = net.GetParams(), flat()
        assert oracle.rel_err(got, want) < TOL, ("params", step)
        # per tensor group too: a whole-vector norm would hide a small tensor (peepholes, biases) that is wrong
        off = 0
        for li, dirs in enumerate(params):
            for di, p in enumerate(dirs):
                n = p.flat().size
                assert oracle.rel_err(got[off:off + n], want[off:off + n]) < TOL, ("layer", li, "dir", di, step)
                off += n
        stt = ctc.GetStats()
        assert abs(stt["obj"] - st.obj) <= 1e-4 * abs(st.obj), step
    assert ctc.GetStats()["sequences"] == 2 * S
