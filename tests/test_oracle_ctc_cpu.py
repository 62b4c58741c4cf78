"""CPU-only: pins the oracle's CTC restatement (oracle/aslp_oracle_ctc.c) against (a) the
fixtures the REFERENCE produced (tests/golden/ctc_*.bin, generator oracle/gen_ctc_golden.cpp),
including the reference's own known-answer tests (small_test / inf_test / grad_check of
src/warp-ctc/tests/test_cpu.cpp), and (b) the reference library itself when it was built here
(oracle/_ref/libwarpctc_ref.so)."""
import ctypes as C
import os

import numpy as np
import pytest

import ctc_golden

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def orc_ctc(oracle, acts, flat, lab_len, in_len, A, mb, want_grad=True):
    fn = oracle.lib.orc_ctc_cost_and_grad
    fn.restype = C.c_int
    fn.argtypes = [f32p, C.c_void_p, i32p, i32p, i32p, C.c_int, C.c_int, f32p]
    costs = np.zeros(mb, np.float32)
    grads = np.zeros_like(acts)
    flat = np.ascontiguousarray(flat if len(flat) else np.zeros(1, np.int32), np.int32)
    fn(acts, grads.ctypes.data if want_grad else None, flat, np.ascontiguousarray(lab_len, np.int32), np.ascontiguousarray(in_len, np.int32), A, mb, costs)
    return costs, grads


@pytest.mark.parametrize("name", ctc_golden.CASES)
def test_oracle_matches_reference_fixtures(oracle, name):
    g = ctc_golden.load(name)
    costs, grads = orc_ctc(oracle, g["acts"], g["flat_labels"], g["label_lengths"], g["input_lengths"], g["A"], g["mb"])
    # identical algorithm and operation order: expect (near) bit equality
    assert np.array_equal(np.isinf(costs), np.isinf(g["costs"]))
    fin = np.isfinite(g["costs"])
    assert np.allclose(costs[fin], g["costs"][fin], rtol=1e-6, atol=1e-6)
    assert not np.isnan(grads).any()
    assert np.allclose(grads, g["grads"], rtol=1e-5, atol=1e-7)


def test_reference_known_answers():
    """The reference's own assertions, evaluated on the reference-produced fixtures."""
    g = ctc_golden.load("small")  # test_cpu.cpp:12-67: exp(-cost) == p[1] * p[7]
    a = g["acts"].reshape(2, 5).astype(np.float64)
    p = np.exp(a) / np.exp(a).sum(1, keepdims=True)
    assert abs(np.exp(-g["costs"][0]) - p[0, 1] * p[1, 2]) < 1e-6
    g = ctc_golden.load("inf")    # test_cpu.cpp:69-122
    assert np.isinf(g["costs"][0]) and not np.isnan(g["grads"]).any()
    g = ctc_golden.load("ragged")
    assert g["costs"][2] == 0.0   # L + repeats > T: cost 0 (cpu_ctc.h:196-198) ...
    A, mb = g["A"], g["mb"]
    gr = g["grads"].reshape(g["maxT"], mb, A)
    assert (gr[:, 2, :] == 0).all()                       # ... and an untouched (zero) gradient
    for n in range(mb):
        assert (gr[g["input_lengths"][n]:, n, :] == 0).all()  # padding frames untouched


def test_grad_check_central_differences(oracle):
    """grad_check of test_cpu.cpp:124-211 (eps 1e-2, rel err < 1e-5) run on the oracle."""
    g = ctc_golden.load("grad_a20_t50_l15")
    A, T = g["A"], g["maxT"]
    _, grads = orc_ctc(oracle, g["acts"], g["flat_labels"], g["label_lengths"], g["input_lengths"], A, 1)
    eps = 1e-2
    num = np.zeros_like(grads)
    acts = g["acts"].copy()
    for i in range(T * A):
        acts[i] += eps
        c1, _ = orc_ctc(oracle, acts, g["flat_labels"], g["label_lengths"], g["input_lengths"], A, 1, want_grad=False)
        acts[i] -= 2 * eps
        c2, _ = orc_ctc(oracle, acts, g["flat_labels"], g["label_lengths"], g["input_lengths"], A, 1, want_grad=False)
        acts[i] += eps
        num[i] = (c1[0] - c2[0]) / (2 * eps)
    diff = ((grads - num) ** 2).sum() / (grads ** 2).sum()
    assert diff < 1e-5


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libwarpctc_ref.so")),
                    reason="reference library only exists in the development container")
def test_oracle_vs_reference_library_random(oracle):
    ref = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libwarpctc_ref.so"))

    class Info(C.Structure):
        _fields_ = [("loc", C.c_int), ("num_threads", C.c_uint), ("pad", C.c_uint)]
    ref.get_workspace_size.argtypes = [i32p, i32p, C.c_int, C.c_int, Info, C.POINTER(C.c_size_t)]
    ref.compute_ctc_loss.argtypes = [f32p, f32p, i32p, i32p, i32p, C.c_int, C.c_int, f32p, C.c_void_p, Info]
    rng = np.random.default_rng(0)
    for trial in range(8):
        A = int(rng.integers(3, 40)); mb = int(rng.integers(1, 9)); maxT = int(rng.integers(5, 60))
        in_len = rng.integers(1, maxT + 1, mb).astype(np.int32); in_len[0] = maxT
        lab_len = np.array([rng.integers(0, max(1, t // 2) + 1) for t in in_len], np.int32)
        flat = np.concatenate([rng.integers(1, A, l) for l in lab_len] + [np.zeros(0, np.int64)]).astype(np.int32)
        if len(flat) == 0:
            flat = np.zeros(1, np.int32)
        acts = (rng.standard_normal(maxT * mb * A) * 2).astype(np.float32)
        info = Info(0, 1, 0)
        sz = C.c_size_t()
        assert ref.get_workspace_size(lab_len, in_len, A, mb, info, C.byref(sz)) == 0
        ws = C.create_string_buffer(sz.value)
        rcost = np.zeros(mb, np.float32); rgrad = np.zeros_like(acts)
        assert ref.compute_ctc_loss(acts, rgrad, flat, lab_len, in_len, A, mb, rcost, ws, info) == 0
        costs, grads = orc_ctc(oracle, acts, flat, lab_len, in_len, A, mb)
        assert np.allclose(costs, rcost, rtol=1e-6, atol=1e-6), trial
        assert np.allclose(grads, rgrad, rtol=1e-5, atol=1e-7), trial


def test_token_error_and_loss_filter(oracle):
    fn = oracle.lib.orc_ctc_token_errors
    fn.restype = C.c_int
    fn.argtypes = [f32p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.POINTER(C.c_int)]
    # frames argmax: 0 1 1 0 2 2 2 0 0 3 -> collapse + drop blanks -> 1 2 3
    ids = [0, 1, 1, 0, 2, 2, 2, 0, 0, 3]
    out = np.zeros((10, 4), np.float32)
    out[np.arange(10), ids] = 1.0
    hl = C.c_int()
    assert fn(out, 4, 10, 4, np.array([1, 2, 3], np.int32), 3, C.byref(hl)) == 0 and hl.value == 3
    assert fn(out, 4, 10, 4, np.array([1, 3], np.int32), 2, C.byref(hl)) == 1
    assert fn(out, 4, 10, 4, np.array([2, 2, 2, 2], np.int32), 4, C.byref(hl)) == 3


def orc_eesen(oracle, probs, labels, in_len, T, S, A):
    fn = oracle.lib.orc_eesen_ctc_mseq
    fn.restype = None
    fn.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, f32p, C.c_int, f32p]
    flat = np.array([v for l in labels for v in l] or [0], np.int32)
    ll = np.array([len(l) for l in labels], np.int32)
    diff = np.zeros((T * S, A), np.float32)
    pzx = np.zeros(S, np.float32)
    fn(np.ascontiguousarray(probs, np.float32), A, T, S, A, flat, ll, np.ascontiguousarray(in_len, np.int32), diff, A, pzx)
    return diff, pzx


@pytest.mark.parametrize("A,S,T,seed", [(5, 1, 6, 0), (12, 4, 15, 1), (40, 6, 30, 2)])
def test_eesen_restatement_agrees_with_pinned_warpctc(oracle, A, S, T, seed):
    """Ctc::EvalParallel (GPU-only in the reference) is the same objective as Warp-CTC on post-softmax
    outputs: its unclipped diff is Warp-CTC's gradient w.r.t. the activations and -pzx its cost
    (SURVEY.md 8c).  The Warp-CTC restatement is pinned to the reference's own code above."""
    rng = np.random.default_rng(seed)
    in_len = rng.integers(max(2, T // 2), T + 1, S).astype(np.int32)
    in_len[0] = T
    labels = []
    for t in in_len:
        L = int(rng.integers(1, max(2, t // 2)))
        lab = rng.integers(1, A, L)
        if L >= 3:
            lab[1] = lab[2]  # a repeat
        labels.append([int(v) for v in lab])
    acts = (rng.standard_normal((T * S, A)) * 1.5).astype(np.float32)
    e = np.exp(acts.astype(np.float64) - acts.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    flat = np.array([v for l in labels for v in l], np.int32)
    ll = np.array([len(l) for l in labels], np.int32)
    wcost, wgrad = orc_ctc(oracle, acts.reshape(-1).copy(), flat, ll, in_len, A, S)
    diff, pzx = orc_eesen(oracle, probs, labels, in_len, T, S, A)
    feas = np.array([len(l) + sum(a == b for a, b in zip(l, l[1:])) <= t for l, t in zip(labels, in_len)])
    assert np.allclose(-pzx[feas], wcost[feas], rtol=1e-4, atol=1e-4)
    wg = wgrad.reshape(T, S, A)
    dg = diff.reshape(T, S, A)
    for s in range(S):
        if feas[s]:
            assert np.abs(dg[:in_len[s], s] - wg[:in_len[s], s]).max() < 2e-4
        assert np.all(dg[in_len[s]:, s] == 0)


@pytest.mark.parametrize("name", ["small", "grad_a20_t50_l15", "grad_a5_t10_l5_mb65", "ragged", "a128_t200"])
def test_eesen_restatement_against_reference_output(oracle, name):
    """The Eesen objective (ctc-loss.cc:115-227, GPU-only in the reference, no output of its own obtainable here) on the softmax of the
    committed fixtures' activations against what the REFERENCE's Warp-CTC CPU code produced for them (tests/golden/ctc_*.bin: costs and
    gradients w.r.t. the activations): Eesen's unclipped diff (ctc-loss.cc:180-189, before the +-1 clip) is that gradient and -pzx that
    cost.  This leans the restatement of row a15 on reference output instead of on another restatement."""
    g = ctc_golden.load(name)
    A, S, T = g["A"], g["mb"], g["maxT"]
    acts = g["acts"].reshape(T * S, A)
    e = np.exp(acts.astype(np.float64) - acts.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    labels, o = [], 0
    for l in g["label_lengths"]:
        labels.append([int(v) for v in g["flat_labels"][o:o + l]])
        o += l
    diff, pzx = orc_eesen(oracle, probs, labels, g["input_lengths"], T, S, A)
    # utterances both formulations define: at least one frame, and enough frames for the labels with their repeats (an empty input is
    # cost 0 for Warp-CTC and "no path" for Eesen's lattice)
    feas = np.array([t > 0 and len(l) + sum(a == b for a, b in zip(l, l[1:])) <= t for l, t in zip(labels, g["input_lengths"])])
    fin = np.isfinite(g["costs"]) & (g["costs"] < 1e30) & feas
    assert fin.any()
    assert np.allclose(-pzx[fin], g["costs"][fin], rtol=1e-4, atol=1e-4)
    ref = g["grads"].reshape(T, S, A)
    dg = diff.reshape(T, S, A)
    for s_ in range(S):
        n = int(g["input_lengths"][s_])
        if fin[s_]:
            assert np.abs(dg[:n, s_] - ref[:n, s_]).max() < 2e-4, s_
