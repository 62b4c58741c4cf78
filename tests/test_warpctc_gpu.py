"""GPU parity of the WarpCtc loss wrapper (kaldi-aslp_amd/nnet/warp-ctc.*, reference
src/aslp-nnet/warp-ctc.cc) against the oracle: CTC cost/grad (pinned to the reference's own CPU
code, see test_oracle_ctc_cpu.py) + the restated wrapper logic (abnormal-loss filter, +-1 clip,
zero diff past an utterance's end, token error rate, Report string), on alphabets that are NOT a
multiple of the matrix row padding (so the strided path is exercised)."""
import ctypes as C
import re

import numpy as np
import pytest
import torch

import nnet_io
from test_oracle_ctc_cpu import orc_ctc

pytestmark = pytest.mark.gpu
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


class FilterState(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("loss_sum", "loss_square_sum", "loss_sum_bak", "loss_square_sum_bak", "obj")] + \
               [(n, C.c_int) for n in ("normal_num", "stat_period", "frames", "sequences")]


def make_batch(rng, A, mb, maxT):
    in_len = rng.integers(max(2, maxT // 2), maxT + 1, mb).astype(np.int32)
    in_len[0] = maxT
    labels = [[int(v) for v in rng.integers(1, A, int(rng.integers(1, max(2, t // 3))))] for t in in_len]
    acts = (rng.standard_normal((maxT * mb, A)) * 2).astype(np.float32)
    return in_len, labels, acts


def oracle_wrapper(oracle, st, acts, labels, in_len, A, mb, maxT):
    """costs/grads from the oracle CTC, then the wrapper logic of warp-ctc.cc:139-173, 288-365"""
    flat = np.array([v for l in labels for v in l], np.int32)
    lab_len = np.array([len(l) for l in labels], np.int32)
    costs, grads = orc_ctc(oracle, acts.reshape(-1).copy(), flat, lab_len, in_len, A, mb)
    grads = grads.reshape(maxT, mb, A)
    for s in range(mb):  # only rows of valid frames are copied back
        grads[in_len[s]:, s] = 0
    keep = np.zeros(mb, np.int32)
    fn = oracle.lib.orc_ctc_loss_filter
    fn.restype = None
    fn.argtypes = [f32p, i32p, C.c_int, C.POINTER(FilterState), i32p]
    fn(costs, in_len, mb, C.byref(st), keep)
    for s in range(mb):
        if not keep[s]:
            grads[:in_len[s], s] = 0
    if not np.isfinite(grads.astype(np.float64).sum()):
        grads[...] = 0
    return costs, np.clip(grads, -1.0, 1.0).reshape(maxT * mb, A), keep


@pytest.mark.parametrize("A,mb,maxT", [(29, 6, 40), (128, 8, 90), (45, 3, 17)])
def test_warpctc_eval_matches_oracle(aslp, oracle, dev, A, mb, maxT):
    rng = np.random.default_rng(A)
    ctc = aslp.WarpCtc()
    st = FilterState(0, 0, 0, 0, 0, 0, 500, 0, 0)
    tot_ref, tot_hyp_err = 0, 0
    for it in range(3):
        in_len, labels, acts = make_batch(rng, A, mb, maxT)
        costs_ref, diff_ref, keep = oracle_wrapper(oracle, st, acts, labels, in_len, A, mb, maxT)
        x = torch.from_numpy(acts).to(dev)
        diff, costs = ctc.Eval(in_len, x, labels)
        assert np.allclose(costs, costs_ref, rtol=1e-4, atol=1e-5)
        d = diff.cpu().numpy()
        assert oracle.rel_err(d, diff_ref) < 1e-4
        assert np.abs(d).max() <= 1.0
        for s in range(mb):
            assert np.all(d.reshape(maxT, mb, A)[in_len[s]:, s] == 0)
        ctc.ErrorRate(in_len, x, labels)
        fn = oracle.lib.orc_ctc_token_errors
        fn.restype = C.c_int
        fn.argtypes = [f32p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.POINTER(C.c_int)]
        for s in range(mb):
            seq = np.ascontiguousarray(acts.reshape(maxT, mb, A)[:in_len[s], s])
            hl = C.c_int()
            tot_hyp_err += fn(seq, A, int(in_len[s]), A, np.array(labels[s], np.int32), len(labels[s]), C.byref(hl))
            tot_ref += len(labels[s])
    stt = ctc.GetStats()
    assert stt["sequences"] == 3 * mb and stt["frames"] == st.frames
    assert abs(stt["obj"] - st.obj) <= 1e-4 * abs(st.obj)
    assert stt["error_tokens"] == tot_hyp_err and stt["ref_tokens"] == tot_ref
    rep = ctc.Report()
    m = re.search(r"Obj\(log\[Pzx\]\) = (\S+) Obj\(frame\) = (\S+) TOKEN_ACCURACY >> (\S+) % <<", rep)
    assert m, rep
    assert abs(float(m.group(1)) - st.obj / (3 * mb)) < 1e-3 * abs(st.obj / (3 * mb))
    assert abs(float(m.group(3)) - 100.0 * (1.0 - tot_hyp_err / tot_ref)) < 1e-2


def test_warpctc_abnormal_loss_is_dropped(aslp, oracle, dev):
    """After the 250-sequence warm-up an utterance whose per-frame loss is > 6 RMS from the mean (or
    whose cost is outside (0, 3000)) contributes a zero diff and no statistics (warp-ctc.cc:309-343)."""
    A, mb, maxT = 11, 50, 12
    rng = np.random.default_rng(0)
    ctc = aslp.WarpCtc()
    st = FilterState(0, 0, 0, 0, 0, 0, 500, 0, 0)
    dropped = 0
    for it in range(7):
        in_len, labels, acts = make_batch(rng, A, mb, maxT)
        if it >= 5:  # make some utterances wildly improbable: strong evidence for blank everywhere
            a = acts.reshape(maxT, mb, A)
            for s in (1, 7):  # finite but > 6 RMS per frame: long label string against all-blank evidence
                a[:, s, :] = -45.0
                a[:, s, 0] = 45.0
                labels[s] = [1 + (i % 2) for i in range(int(in_len[s]) // 2)]
            a[:, 3, :] = -60.0  # probability underflows to 0: infinite cost
            a[:, 3, 0] = 60.0
        costs_ref, diff_ref, keep = oracle_wrapper(oracle, st, acts, labels, in_len, A, mb, maxT)
        diff, costs = ctc.Eval(in_len, torch.from_numpy(acts).to(dev), labels)
        d = diff.cpu().numpy()
        assert oracle.rel_err(d, diff_ref) < 1e-4, it
        if it >= 5:
            assert keep[1] == 0 and keep[7] == 0 and keep[3] == 0
            assert np.isinf(costs[3]) and np.isfinite(costs[1])
            assert np.all(d.reshape(maxT, mb, A)[:, 1] == 0) and np.all(d.reshape(maxT, mb, A)[:, 3] == 0)
            dropped += 3
    assert dropped == 6
    assert abs(ctc.GetStats()["obj"] - st.obj) <= 1e-4 * abs(st.obj)


def test_warpctc_cpu_mode_fails_loudly(aslp, dev):
    import kaldi_aslp_amd.nnet as N
    ctc = aslp.WarpCtc()
    # there is no host path: asking for it is an error, not a silent fallback
    assert not hasattr(N.lib, "aslp_warpctc_set_use_gpu")


def test_train_step_warpctc_lstm(aslp, oracle, dev, tmp_path):
    """One end-to-end CTC training step of a small LSTM + affine net: Propagate -> WarpCtc ->
    Backpropagate(+Update) reproduces the oracle chain (LSTM oracle + affine + CTC oracle)."""
    D, Cc, R, A, T, S = 9, 12, 6, 13, 10, 3
    rng = np.random.default_rng(12)
    p = oracle.LstmDir(D, Cc, R, False, rng, scale=0.3)
    g = oracle.LstmDir(D, Cc, R, False, zero=True)
    W = (rng.standard_normal((A, R)) * 0.3).astype(np.float32)
    b = np.zeros(A, np.float32)
    path = tmp_path / "ctc.nnet"
    nnet_io.write_simple_nnet(path, [("<LstmProjectedStreams>", D, R, nnet_io.lstm([p], 5.0, Cc)),
                                     ("<AffineTransform>", R, A, nnet_io.affine(W, b))])
    net = aslp.Nnet.Read(path)
    lr = 0.01
    net.SetTrainOptions(learn_rate=lr, momentum=0.0)
    ctc = aslp.WarpCtc()
    st = FilterState(0, 0, 0, 0, 0, 0, 500, 0, 0)
    in_len = np.array([T, T - 3, T - 1], np.int32)
    labels = [[1, 2, 2, 3], [4, 5], [6]]
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    # oracle chain
    buf = p.forward(x, T, S, init_state=np.zeros((S, p.width), np.float32))
    h = p.out_of(buf, T, S)
    y = oracle.add_mat_mat(np.tile(b, (T * S, 1)), 1.0, h, False, W, True, 1.0)
    _, diff, _ = oracle_wrapper(oracle, st, y, labels, in_len, A, S, T)
    dW = oracle.add_mat_mat(np.zeros_like(W), 1.0, diff, True, h, False, 0.0)
    db = diff.sum(0)
    dh = oracle.add_mat_mat(np.zeros_like(h), 1.0, diff, False, W, False, 0.0)
    dbuf, _ = p.backward(dh, T, S, buf)
    p.grads(g, x, T, S, buf, dbuf, 0.0, 5.0)
    p.update(g, lr)
    W -= lr * dW
    b -= lr * db
    # engine
    net.TrainStepWarpCtc(ctc, torch.from_numpy(x).to(dev), in_len, labels)
    want = np.concatenate([p.flat(), W.ravel(), b])
    assert oracle.rel_err(net.GetParams(), want) < 1e-4
    assert ctc.GetStats()["sequences"] == S
