"""Self-consistency of oracle/aslp_oracle_temporal.c (RowConvolution, CompactFsmn; pinned against the reference's own
CuMatrix library in tests/test_oracle_ref_blas_cpu.py -- no reference tests / golden vectors exist for these components): the CPU suite checks the
hand-written backward passes against central differences of the forward pass and pins down the
one place where the reference is knowingly not the exact gradient."""
import numpy as np

import oracle_lib as O


def test_fsmn_forward_definition():
    rng = np.random.default_rng(0)
    D, P, F, T = 5, 3, 2, 11
    m = O.Fsmn(D, P, F, rng)
    x = rng.standard_normal((T, D)).astype(np.float32)
    out = m.propagate(x)
    ref = x.astype(np.float64).copy()
    for t in range(T):
        for j in range(P + F + 1):
            r = t + j - P
            if 0 <= r < T:
                ref[t] += m.coef[j].astype(np.float64) * x[r]
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-6)


def test_fsmn_backward_is_gradient():
    rng = np.random.default_rng(1)
    D, P, F, T = 4, 3, 2, 9
    m = O.Fsmn(D, P, F, rng)
    x = rng.standard_normal((T, D)).astype(np.float32)
    Wt = rng.standard_normal((T, D))
    loss = lambda xx: float((m.propagate(xx).astype(np.float64) * Wt).sum())
    in_diff = m.backpropagate(x, Wt.astype(np.float32), 0.0)
    eps = 1e-2
    v = rng.standard_normal(x.shape).astype(np.float32)
    fd = (loss(x + eps * v) - loss(x - eps * v)) / (2 * eps)
    an = float((in_diff.astype(np.float64) * v).sum())
    assert abs(fd - an) < 1e-2 * max(1, abs(an))
    corr = m.corr.copy()
    v = rng.standard_normal(m.coef.shape).astype(np.float32)
    c0 = m.coef.copy()
    m.coef[...] = c0 + eps * v; lp = loss(x)
    m.coef[...] = c0 - eps * v; lm = loss(x)
    m.coef[...] = c0
    assert abs((lp - lm) / (2 * eps) - float((corr.astype(np.float64) * v).sum())) < 1e-2 * max(1, abs(lp))
    # clipping and the (momentum-free) update
    clip = float(np.abs(corr).max()) / 3
    m.backpropagate(x, Wt.astype(np.float32), clip)
    np.testing.assert_array_equal(m.corr, np.clip(corr, -clip, clip))
    m.update(0.1)
    np.testing.assert_allclose(m.coef, c0 - np.float32(0.1) * m.corr, rtol=1e-6, atol=1e-7)


def test_rowconv_forward_definition_and_tail_replication():
    rng = np.random.default_rng(2)
    D, K, T, S = 4, 2, 6, 3
    lens = np.array([6, 4, 1], np.int32)
    m = O.RowConv(D, K, rng)
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    out = m.propagate(x, T, S, lens).reshape(T, S, D)
    xs = x.reshape(T, S, D)
    for s, L in enumerate(lens):
        for t in range(T):
            if t >= L:
                assert np.all(out[t, s] == 0)
                continue
            ref = sum(m.w[:, k].astype(np.float64) * xs[min(t + k, L - 1), s] for k in range(K + 1))
            np.testing.assert_allclose(out[t, s], ref, rtol=1e-5, atol=1e-6)


def test_rowconv_backward_vs_finite_differences():
    """w_diff is the exact gradient.  in_diff is exact for frames whose window does not touch the
    replicated tail; the share of the diff that lands on the replicas is dropped instead of being
    added to the last real frame (nnet-row-convolution.cc:168-174) -- checked explicitly."""
    rng = np.random.default_rng(3)
    D, K, T, S = 3, 2, 7, 2
    lens = np.array([7, 5], np.int32)
    m = O.RowConv(D, K, rng)
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    Wt = rng.standard_normal((T * S, D))
    loss = lambda xx: float((m.propagate(xx, T, S, lens).astype(np.float64) * Wt).sum())
    m.propagate(x, T, S, lens)
    in_diff = m.backpropagate(Wt.astype(np.float32), T, S, lens)
    eps = 1e-2
    # weights
    v = rng.standard_normal(m.w.shape).astype(np.float32)
    w0 = m.w.copy()
    m.w[...] = w0 + eps * v; lp = loss(x)
    m.w[...] = w0 - eps * v; lm = loss(x)
    m.w[...] = w0
    an = float((m.w_diff.astype(np.float64) * v).sum())
    assert abs((lp - lm) / (2 * eps) - an) < 1e-2 * max(1, abs(an))
    # inputs: exact gradient = reported in_diff + the dropped tail share on frame L-1
    exact = in_diff.astype(np.float64).reshape(T, S, D).copy()
    od = Wt.reshape(T, S, D)
    for s, L in enumerate(lens):
        for t in range(L):
            for k in range(K + 1):
                if t + k >= L:
                    exact[L - 1, s] += m.w[:, k] * od[t, s]
    v = rng.standard_normal(x.shape).astype(np.float32)
    fd = (loss(x + eps * v) - loss(x - eps * v)) / (2 * eps)
    an = float((exact.reshape(T * S, D) * v).sum())
    assert abs(fd - an) < 1e-2 * max(1, abs(an))
    # frames past the length get no diff
    idf = in_diff.reshape(T, S, D)
    assert np.all(idf[5:, 1] == 0)
    # momentum update
    m.update(0.1, 0.5)
    np.testing.assert_allclose(m.w, w0 - np.float32(0.1) * m.w_diff, rtol=1e-6, atol=1e-7)
    m.update(0.1, 0.5)
    np.testing.assert_allclose(m.w_corr, 1.5 * m.w_diff, rtol=1e-6, atol=1e-7)
