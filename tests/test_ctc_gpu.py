"""GPU parity of the HIP CTC forward-backward (called through the Warp-CTC C ABI of
include/aslp_ctc.h) against (a) the fixtures produced by the reference's own CPU code
(tests/golden/ctc_*.bin) and (b) the oracle restatement on seeded random ragged minibatches.
Tolerance (BASELINE.json north_star): CTC loss fp32 within 1e-4 relative; gradients 1e-4
relative Frobenius."""
import numpy as np
import pytest
import torch

import ctc_golden
from test_oracle_ctc_cpu import orc_ctc

pytestmark = pytest.mark.gpu


def split_labels(flat, lens):
    out, o = [], 0
    for l in lens:
        out.append([int(v) for v in flat[o:o + l]])
        o += l
    return out


@pytest.mark.parametrize("name", ctc_golden.CASES)
def test_ctc_matches_reference_fixtures(aslp, oracle, dev, name):
    g = ctc_golden.load(name)
    A, mb, maxT = g["A"], g["mb"], g["maxT"]
    acts = torch.from_numpy(g["acts"].reshape(maxT * mb, A)).to(dev)
    costs, grads = aslp.ops.ctc_loss(acts, split_labels(g["flat_labels"], g["label_lengths"]), g["input_lengths"])
    assert np.array_equal(np.isinf(costs), np.isinf(g["costs"]))
    fin = np.isfinite(g["costs"])
    rel = np.abs(costs[fin] - g["costs"][fin]) / np.maximum(np.abs(g["costs"][fin]), 1e-6)
    assert rel.max() < 1e-4 if fin.any() else True, rel
    gr = grads.cpu().numpy().reshape(-1)
    assert not np.isnan(gr).any()
    assert oracle.rel_err(gr, g["grads"]) < 1e-4
    # untouched rows stay exactly zero (padding frames, infeasible utterances)
    assert np.array_equal(gr == 0, g["grads"] == 0) or (np.abs(gr[g["grads"] == 0]).max() < 1e-30)


def test_reference_small_test_known_answer(aslp, dev):
    """small_test of src/warp-ctc/tests/test_cpu.cpp:12-67: exp(-cost) == p[1] * p[7], scores only."""
    acts = torch.tensor([[0.1, 0.6, 0.1, 0.1, 0.1], [0.1, 0.1, 0.6, 0.1, 0.1]], device=dev)
    costs, grads = aslp.ops.ctc_loss(acts, [[1, 2]], [2], want_grad=False)
    p = torch.softmax(acts.double(), 1)
    assert grads is None
    assert abs(np.exp(-costs[0]) - (p[0, 1] * p[1, 2]).item()) < 1e-6


@pytest.mark.parametrize("A,mb,maxT,seed", [(4, 1, 3, 0), (30, 7, 45, 1), (128, 32, 200, 2), (3000, 4, 60, 3), (50, 300, 20, 4), (128, 8, 800, 5)])
def test_ctc_random_ragged_vs_oracle(aslp, oracle, dev, A, mb, maxT, seed):
    rng = np.random.default_rng(seed)
    in_len = rng.integers(1, maxT + 1, mb).astype(np.int32)
    in_len[0] = maxT
    labels = []
    for t in in_len:
        L = int(rng.integers(0, max(1, t // 2) + 1))
        lab = rng.integers(1, A, L)
        if L >= 3:  # force repeats like genLabels (tests/test.h:48-53)
            lab[L // 2] = lab[L // 2 + 1] if L // 2 + 1 < L else lab[L // 2]
            lab[L // 2 - 1] = lab[L // 2]
        labels.append([int(v) for v in lab])
    acts = (rng.standard_normal((maxT * mb, A)) * 2).astype(np.float32)
    flat = np.array([v for l in labels for v in l], np.int32)
    lab_len = np.array([len(l) for l in labels], np.int32)
    rcost, rgrad = orc_ctc(oracle, acts.reshape(-1).copy(), flat, lab_len, in_len, A, mb)
    costs, grads = aslp.ops.ctc_loss(torch.from_numpy(acts).to(dev), labels, in_len)
    assert np.allclose(costs, rcost, rtol=1e-4, atol=1e-5)
    assert oracle.rel_err(grads.cpu().numpy().reshape(-1), rgrad) < 1e-4


def test_ctc_gradient_sums_to_zero_per_frame(aslp, dev):
    """Size-independent property at the BASELINE shape (S = 32 utterances, T = 800, A = 128, L = T/4):
    d cost / d activations of a softmax-normalised loss sums to 0 over the alphabet for every valid frame."""
    rng = np.random.default_rng(5)
    mb, T, A = 32, 800, 128
    in_len = rng.integers(200, T + 1, mb)
    in_len[0] = T
    labels = [[int(v) for v in rng.integers(1, A, int(t) // 4)] for t in in_len]
    acts = torch.from_numpy(rng.standard_normal((T * mb, A)).astype(np.float32)).to(dev)
    costs, grads = aslp.ops.ctc_loss(acts, labels, in_len)
    assert np.isfinite(costs).all() and (costs > 0).all()
    g = grads.view(T, mb, A)
    # log-likelihoods are ~ -2500 here: one fp32 ulp there is 2.4e-4, which bounds how well
    # p - exp(out - log p - logZ) can cancel (the reference's float CPU code has the same limit)
    assert g.sum(-1).abs().max().item() < 1e-2
    for n in range(mb):
        assert (g[int(in_len[n]):, n, :] == 0).all()


def test_ctc_errors(aslp, dev):
    acts = torch.zeros(4, 5, device=dev)
    with pytest.raises(RuntimeError, match="invalid value"):
        aslp.ops.ctc_loss(acts, [[7]], [4])  # label outside the alphabet
