"""GPU parity of the Eesen-style Ctc loss (kaldi-aslp_amd/nnet/ctc-loss.*, fused lattice in
csrc/ctc.hip) and of the per-row kernel ABI cudaF_compute_ctc_*_multiple_sequence
(csrc/ctc_eesen.hip) against oracle/aslp_oracle_ctc.c::orc_eesen_ctc_mseq, which restates the
reference's device kernels (cu-kernels.cu:3276-3534) and agrees with the pinned Warp-CTC
restatement (tests/test_oracle_ctc_cpu.py)."""
import ctypes as C
import re

import numpy as np
import pytest
import torch

from test_oracle_ctc_cpu import orc_eesen
from test_warpctc_gpu import FilterState, f32p, i32p

pytestmark = pytest.mark.gpu


def make_batch(rng, A, S, T):
    in_len = rng.integers(max(2, T // 2), T + 1, S).astype(np.int32)
    in_len[0] = T
    labels = []
    for t in in_len:
        L = int(rng.integers(1, max(2, t // 3)))
        lab = rng.integers(1, A, L)
        if L >= 3:
            lab[1] = lab[2]
        labels.append([int(v) for v in lab])
    acts = (rng.standard_normal((T * S, A)) * 1.5).astype(np.float32)
    e = np.exp(acts.astype(np.float64) - acts.max(1, keepdims=True))
    return in_len, labels, (e / e.sum(1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("A,S,T", [(29, 5, 30), (128, 8, 60), (45, 1, 17)])
def test_eval_parallel_matches_oracle(aslp, oracle, dev, A, S, T):
    rng = np.random.default_rng(A)
    ctc = aslp.Ctc()
    st = FilterState(0, 0, 0, 0, 0, 0, 100, 0, 0)
    filt = oracle.lib.orc_eesen_ctc_loss_filter
    filt.restype = None
    filt.argtypes = [f32p, i32p, C.c_int, C.POINTER(FilterState), i32p]
    for it in range(3):
        in_len, labels, probs = make_batch(rng, A, S, T)
        diff_ref, pzx = orc_eesen(oracle, probs, labels, in_len, T, S, A)
        costs_ref = -pzx
        keep = np.zeros(S, np.int32)
        filt(costs_ref, in_len, S, C.byref(st), keep)
        d3 = diff_ref.reshape(T, S, A)
        for s in range(S):
            if not keep[s]:
                d3[:in_len[s], s] = 0
        diff_ref = np.clip(diff_ref, -1, 1)
        x = torch.from_numpy(probs).to(dev)
        diff, costs = ctc.EvalParallel(in_len, x, labels)
        assert np.allclose(costs, costs_ref, rtol=1e-4, atol=1e-4)
        assert oracle.rel_err(diff.cpu().numpy(), diff_ref) < 2e-4
        ctc.ErrorRateMSeq(in_len, x, labels)
    stt = ctc.GetStats()
    assert stt["sequences"] == 3 * S and stt["frames"] == st.frames
    assert abs(stt["obj"] - st.obj) <= 1e-4 * abs(st.obj)
    assert re.search(r"Obj\(log\[Pzx\]\) = \S+ Obj\(frame\) = \S+ TOKEN_ACCURACY >> \S+ % <<", ctc.Report())


def test_eval_single_sequence(aslp, oracle, dev):
    A, T = 20, 25
    rng = np.random.default_rng(3)
    in_len, labels, probs = make_batch(rng, A, 1, T)
    diff_ref, pzx = orc_eesen(oracle, probs, labels, in_len, T, 1, A)
    ctc = aslp.Ctc()
    x = torch.from_numpy(probs).to(dev)
    diff, costs = ctc.Eval(x, labels[0])
    assert abs(costs[0] + pzx[0]) < 1e-4 * abs(pzx[0])
    assert oracle.rel_err(diff.cpu().numpy(), np.clip(diff_ref, -1, 1)) < 2e-4
    ctc.ErrorRate(x, labels[0])
    st = ctc.GetStats()
    assert st["sequences"] == 1 and st["frames"] == T and st["ref_tokens"] == len(labels[0])


def test_per_row_kernel_abi_reproduces_the_loss(aslp, oracle, dev):
    """Drive cudaF_compute_ctc_{alpha,beta,error}_multiple_sequence exactly as Ctc::EvalParallel does in
    the reference (ctc-loss.cc:155-191): log, T alpha rows, T beta rows, pzx on the host, error, then
    diff = err.*y - y*rowsum(err.*y)."""
    A, S, T = 17, 4, 14
    rng = np.random.default_rng(8)
    in_len, labels, probs = make_batch(rng, A, S, T)
    diff_ref, pzx_ref = orc_eesen(oracle, probs, labels, in_len, T, S, A)
    maxL = max(len(l) for l in labels)
    E = 2 * maxL + 1
    lab = -np.ones((S, E), np.int32)
    for s, l in enumerate(labels):
        lab[s, 0:2 * len(l) + 1:2] = 0
        lab[s, 1:2 * len(l):2] = l
    lab_d = torch.from_numpy(lab).to(dev)
    len_d = torch.from_numpy(in_len).to(dev)
    explen_d = torch.tensor([2 * len(l) + 1 for l in labels], dtype=torch.int32, device=dev)
    y = torch.from_numpy(probs).to(dev)
    logy = torch.log(y)
    alpha = torch.full((T * S, E), -1e30, device=dev)
    beta = torch.full((T * S, E), -1e30, device=dev)
    lib, MD, D3 = aslp.lib, aslp.MatrixDim, aslp.Dim3
    vp, ci = C.c_void_p, C.c_int
    lib.cudaF_compute_ctc_alpha_multiple_sequence.argtypes = [D3, D3, vp, ci, ci, MD, vp, MD, vp, ci, vp]
    lib.cudaF_compute_ctc_beta_multiple_sequence.argtypes = [D3, D3, vp, ci, ci, MD, vp, MD, vp, ci, vp, vp]
    lib.cudaF_compute_ctc_error_multiple_sequence.argtypes = [D3, D3, vp, ci, MD, vp, vp, MD, vp, vp, ci, vp, vp]
    for fn in (lib.cudaF_compute_ctc_alpha_multiple_sequence, lib.cudaF_compute_ctc_beta_multiple_sequence,
               lib.cudaF_compute_ctc_error_multiple_sequence):
        fn.restype = None
    dl, dp = MD(T * S, E, E), MD(T * S, A, A)
    for t in range(T):
        lib.cudaF_compute_ctc_alpha_multiple_sequence(D3(), D3(), alpha.data_ptr(), S, t, dl, logy.data_ptr(), dp, lab_d.data_ptr(), E,
                                                      len_d.data_ptr())
    for t in range(T - 1, -1, -1):
        lib.cudaF_compute_ctc_beta_multiple_sequence(D3(), D3(), beta.data_ptr(), S, t, dl, logy.data_ptr(), dp, lab_d.data_ptr(), E,
                                                     len_d.data_ptr(), explen_d.data_ptr())
    aslp.check_error()
    al = alpha.cpu().numpy().reshape(T, S, E).astype(np.float64)
    pzx = np.array([np.logaddexp(al[in_len[s] - 1, s, 2 * len(labels[s])], al[in_len[s] - 1, s, 2 * len(labels[s]) - 1])
                    for s in range(S)], np.float32)
    assert np.allclose(pzx, pzx_ref, rtol=1e-4, atol=1e-4)
    err = torch.zeros(T * S, A, device=dev)
    pzx_d = torch.from_numpy(pzx).to(dev)
    lib.cudaF_compute_ctc_error_multiple_sequence(D3(), D3(), err.data_ptr(), S, dp, alpha.data_ptr(), beta.data_ptr(), dl, y.data_ptr(),
                                                  lab_d.data_ptr(), E, len_d.data_ptr(), pzx_d.data_ptr())
    aslp.check_error()
    err = err * y
    diff = err - y * err.sum(1, keepdim=True)
    assert oracle.rel_err(diff.cpu().numpy(), diff_ref) < 2e-4
