"""Self-consistency of the recurrent oracle (oracle/aslp_oracle_rnn.c).  The reference ships no
tests or golden vectors for its LSTM/GRU components; the gate blocks are pinned against the reference's
own CuMatrix library in tests/test_oracle_ref_blas_cpu.py, and what is checked here is that the hand-written
BPTT is the gradient of its forward pass (directional central differences), that the direction /
state / masking plumbing does what the reference's comments say, and the update rule."""
import numpy as np
import pytest

import oracle_lib as O

VARIANTS = {  # name: (R, cifg)
    "lstm": (0, False),
    "projected": (6, False),
    "cifg": (6, True),
}


def loss_lstm(p, x, T, S, Wt, reverse, init):
    buf = p.forward(x, T, S, reverse=reverse, init_state=init)
    return float((p.out_of(buf, T, S).astype(np.float64) * Wt).sum())


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("reverse", [False, True])
def test_lstm_backward_is_gradient_of_forward(variant, reverse):
    R, cifg = VARIANTS[variant]
    D, Cc, T, S = 5, 7, 6, 3
    rng = np.random.default_rng(11)
    p = O.LstmDir(D, Cc, R, cifg, rng, scale=0.5)
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    init = None if reverse else (rng.standard_normal((S, p.width)) * 0.3).astype(np.float32)
    Wt = rng.standard_normal((T * S, p.rec))
    buf = p.forward(x, T, S, reverse=reverse, init_state=init)
    dbuf, in_diff = p.backward(Wt.astype(np.float32), T, S, buf, reverse=reverse)
    g = O.LstmDir(D, Cc, R, cifg, zero=True)
    p.grads(g, x, T, S, buf, dbuf, 0.0, 0.0, reverse=reverse)
    eps = 2e-2
    # input direction
    for trial in range(3):
        v = rng.standard_normal(x.shape).astype(np.float32)
        fd = (loss_lstm(p, x + eps * v, T, S, Wt, reverse, init) - loss_lstm(p, x - eps * v, T, S, Wt, reverse, init)) / (2 * eps)
        an = float((in_diff.astype(np.float64) * v).sum())
        assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), ("in_diff", fd, an)
    # each parameter tensor
    for name in ("w_x", "w_r", "bias", "peep_i", "peep_f", "peep_o", "w_rm"):
        if (name == "peep_i" and cifg) or (name == "w_rm" and R == 0):
            continue
        w = getattr(p, name)
        v = rng.standard_normal(w.shape).astype(np.float32)
        w0 = w.copy()
        w[...] = w0 + eps * v
        lp = loss_lstm(p, x, T, S, Wt, reverse, init)
        w[...] = w0 - eps * v
        lm = loss_lstm(p, x, T, S, Wt, reverse, init)
        w[...] = w0
        fd = (lp - lm) / (2 * eps)
        an = float((getattr(g, name).astype(np.float64) * v).sum())
        assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), (name, fd, an)


def test_lstm_momentum_clip_update():
    D, Cc, R, T, S = 4, 5, 3, 5, 2
    rng = np.random.default_rng(5)
    p = O.LstmDir(D, Cc, R, False, rng, scale=0.5)
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    od = rng.standard_normal((T * S, R)).astype(np.float32)
    buf = p.forward(x, T, S)
    dbuf, _ = p.backward(od, T, S, buf)
    g0 = O.LstmDir(D, Cc, R, False, zero=True)
    p.grads(g0, x, T, S, buf, dbuf, 0.0, 0.0)
    # momentum: second accumulation = g + 0.5 g; then clipping bounds every element
    g1 = O.LstmDir(D, Cc, R, False, zero=True)
    p.grads(g1, x, T, S, buf, dbuf, 0.0, 0.0)
    p.grads(g1, x, T, S, buf, dbuf, 0.5, 0.0)
    np.testing.assert_allclose(g1.flat(), 1.5 * g0.flat(), rtol=1e-5, atol=1e-6)
    clip = float(np.abs(g0.w_x).max()) * 0.25
    g2 = O.LstmDir(D, Cc, R, False, zero=True)
    p.grads(g2, x, T, S, buf, dbuf, 0.0, clip)
    np.testing.assert_allclose(g2.flat(), np.clip(g0.flat(), -clip, clip), rtol=0, atol=0)
    before = p.flat().copy()
    p.update(g2, 0.1)
    np.testing.assert_allclose(p.flat(), before - np.float32(0.1) * g2.flat(), rtol=1e-6, atol=1e-7)


def test_lstm_state_carry_equals_one_long_batch():
    """Forward direction: running [0,T1) then [T1,T) with the saved last row block as history
    equals one pass over T frames (nnet-lstm-projected-streams.h:332,432)."""
    D, Cc, R, S, T1, T2 = 4, 6, 3, 2, 4, 3
    rng = np.random.default_rng(2)
    p = O.LstmDir(D, Cc, R, False, rng, scale=0.5)
    x = rng.standard_normal(((T1 + T2) * S, D)).astype(np.float32)
    full = p.out_of(p.forward(x, T1 + T2, S), T1 + T2, S)
    b1 = p.forward(x[:T1 * S], T1, S)
    b2 = p.forward(x[T1 * S:], T2, S, init_state=b1[T1 * S:(T1 + 1) * S])
    got = np.concatenate([p.out_of(b1, T1, S), p.out_of(b2, T2, S)])
    np.testing.assert_allclose(got, full, rtol=1e-6, atol=1e-7)


def test_lstm_reverse_masking():
    """Backward direction with per-stream lengths: frames past the length are zero rows and the
    valid prefix equals the same stream run alone at its own length
    (nnet-blstm-projected-streams.h:654-657)."""
    D, Cc, R, S, T = 3, 5, 4, 3, 6
    lens = np.array([6, 4, 2], np.int32)
    rng = np.random.default_rng(4)
    p = O.LstmDir(D, Cc, R, False, rng, scale=0.5)
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    buf = p.forward(x, T, S, reverse=True, seq_len=lens)
    out = p.out_of(buf, T, S).reshape(T, S, R)
    for s, L in enumerate(lens):
        assert np.all(buf.reshape(T + 2, S, -1)[L + 1:T + 1, s] == 0)
        xs = np.ascontiguousarray(x.reshape(T, S, D)[:L, s])
        alone = p.out_of(p.forward(xs, int(L), 1, reverse=True), int(L), 1)
        np.testing.assert_allclose(out[:L, s], alone, rtol=1e-6, atol=1e-7)


def test_lstm_cell_clip():
    D, Cc, T, S = 2, 3, 40, 1
    rng = np.random.default_rng(8)
    p = O.LstmDir(D, Cc, 0, False, rng, scale=0.1)
    p.bias[:] = 0
    p.bias[0:Cc] = 20.0        # g -> 1
    p.bias[Cc:2 * Cc] = 20.0   # i -> 1
    p.bias[2 * Cc:3 * Cc] = 30.0  # f -> 1: c grows by ~1 each frame... scale input so it saturates at 50
    x = np.zeros((T * S, D), np.float32)
    init = np.zeros((S, p.width), np.float32)
    init[:, 4 * Cc:5 * Cc] = 49.5
    buf = p.forward(x, T, S, init_state=init)
    c = buf[S:(T + 1) * S, 4 * Cc:5 * Cc]
    assert c.max() == 50.0


def loss_gru(p, x, T, S, Wt, init):
    return float((p.out_of(p.forward(x, T, S, init_state=init), T, S).astype(np.float64) * Wt).sum())


def test_gru_backward_is_gradient_of_forward():
    D, H, T, S = 5, 6, 6, 3
    rng = np.random.default_rng(21)
    p = O.Gru(D, H, rng, scale=0.5)
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    init = (rng.standard_normal((S, 5 * H)) * 0.3).astype(np.float32)
    Wt = rng.standard_normal((T * S, H))
    buf = p.forward(x, T, S, init_state=init)
    dbuf, in_diff = p.backward(Wt.astype(np.float32), T, S, buf)
    g = O.Gru(D, H, zero=True)
    p.grads(g, x, T, S, buf, dbuf, 0.0, 0.0)
    eps = 2e-2
    v = rng.standard_normal(x.shape).astype(np.float32)
    fd = (loss_gru(p, x + eps * v, T, S, Wt, init) - loss_gru(p, x - eps * v, T, S, Wt, init)) / (2 * eps)
    an = float((in_diff.astype(np.float64) * v).sum())
    assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), ("in_diff", fd, an)
    for name in O.Gru.NAMES:
        w = getattr(p, name)
        v = rng.standard_normal(w.shape).astype(np.float32)
        w0 = w.copy()
        w[...] = w0 + eps * v
        lp = loss_gru(p, x, T, S, Wt, init)
        w[...] = w0 - eps * v
        lm = loss_gru(p, x, T, S, Wt, init)
        w[...] = w0
        fd = (lp - lm) / (2 * eps)
        an = float((getattr(g, name).astype(np.float64) * v).sum())
        assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), (name, fd, an)
