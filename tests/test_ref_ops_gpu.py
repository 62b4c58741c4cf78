"""The HIP kernels behind the B1 entry points against outputs of the REFERENCE's own CuMatrix CPU branch
(tests/golden/cumatrix_ops.bin, generator oracle/gen_cumatrix_golden.cpp): index operations bit-exact, floating point
to 1e-6 relative (well inside the 1e-4 bar)."""
import ctypes as C

import numpy as np
import pytest
import torch

import cumatrix_golden

pytestmark = pytest.mark.gpu


def close(a, b, tol=1e-6):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol


def test_kernels_match_reference_cumatrix_cpu(aslp, dev):
    g = cumatrix_golden.load()
    from kaldi_aslp_amd._lib import D3, check_error
    ops, lib = aslp.ops, aslp.lib
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    x, d = t(g["x"]), t(g["d"])
    y = torch.empty_like(x)
    ops.sigmoid(y, x); assert close(y.cpu().numpy(), g["sigmoid"])
    e = torch.empty_like(x)
    ops.diff_sigmoid(e, t(g["sigmoid"]), d); assert close(e.cpu().numpy(), g["diff_sigmoid"])
    ops.tanh(y, x); assert close(y.cpu().numpy(), g["tanh"])
    ops.diff_tanh(e, t(g["tanh"]), d); assert close(e.cpu().numpy(), g["diff_tanh"])

    def inplace(fn, src, *args):
        m = t(src)
        fn(D3, D3, ops.ptr(m), *args, ops.dim(m))
        check_error()
        return m.cpu().numpy()
    assert np.array_equal(inplace(lib.cudaF_apply_floor, g["x"], 0.0), g["relu"])
    assert np.array_equal(inplace(lib.cudaF_apply_heaviside, g["x"]), g["heaviside"])
    m = t(g["x"]); lib.cudaF_apply_floor(D3, D3, ops.ptr(m), -1.5, ops.dim(m)); lib.cudaF_apply_ceiling(D3, D3, ops.ptr(m), 2.5, ops.dim(m))
    assert np.array_equal(m.cpu().numpy(), g["floor_ceil"])
    m = t(g["x"]); lib.aslp_apply_clamp(ops.ptr(m), ops.dim(m), -1.5, 2.5)
    assert np.array_equal(m.cpu().numpy(), g["floor_ceil"])
    m = t(g["x"]); lib.cudaF_mul_elements(D3, D3, ops.ptr(m), ops.ptr(d), ops.dim(m), ops.dim(d).stride)
    assert np.array_equal(m.cpu().numpy(), g["mul_elements"])
    assert close(inplace(lib.cudaF_apply_pow, g["x"], 2.0), g["pow2"])
    assert close(inplace(lib.cudaF_apply_log, g["pos"]), g["log"])
    assert close(inplace(lib.cudaF_apply_exp, g["d"]), g["exp"])
    assert close(inplace(lib.cudaF_invert_elements, g["pos"]), g["invert"])
    assert np.array_equal(ops.find_row_max_id(t(g["argmax_in"])).cpu().numpy(), g["argmax"])
    # DiffXent
    p = t(g["xent_in"]); tgt = t(g["xent_tgt"], torch.int32); lp = torch.empty(p.shape[0], device=dev)
    lib.cudaF_diff_xent(D3, D3, ops.ptr(tgt), ops.ptr(p), ops.ptr(lp), ops.dim(p))
    assert close(p.cpu().numpy(), g["xent_diff"]) and close(lp.cpu().numpy(), g["xent_logpost"][0])
    # index operations: bit-exact
    f = t(g["splice_in"]); off = t(g["splice_off"], torch.int32)
    o = torch.empty(f.shape[0], f.shape[1] * len(g["splice_off"]), device=dev)
    ops.splice(o, f, off); assert np.array_equal(o.cpu().numpy(), g["splice_out"])
    cols = t(g["copy_cols"], torch.int32)
    oc = torch.empty(f.shape[0], len(g["copy_cols"]), device=dev)
    lib.cudaF_copy(D3, D3, ops.ptr(oc), ops.ptr(f), ops.ptr(cols), ops.dim(oc), ops.dim(f))
    assert np.array_equal(oc.cpu().numpy(), g["copy_out"])
    mask = t(g["rand_mask"], torch.int32)
    orr = torch.empty_like(f)
    ops.randomize(orr, f, mask); assert np.array_equal(orr.cpu().numpy(), g["randomize_out"])


def test_composite_kernels_match_reference_cumatrix_cpu(aslp, dev):
    """Round 2 pins (same records as tests/test_oracle_ref_ops_cpu.py): the HIP kernels behind CompactFsmn's product
    (cudaF_add_conv_mat_mat_elements), cudaF_add_mat_mat_elements, the peephole term (cudaF_add_mat_diag_vec, both
    orientations), the broadcasts (cudaF_add_vec_to_rows / _cols, cudaF_mul_cols_vec / _rows_vec), cudaF_regularize_l1 and
    cudaF_copy_cols / cudaF_add_cols against what the reference's own CuMatrix CPU branch produced."""
    g = cumatrix_golden.load()
    from kaldi_aslp_amd._lib import D3, check_error
    ops, lib = aslp.ops, aslp.lib
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    A, B = t(g["conv_A"]), t(g["conv_B"])
    d = t(g["conv_dst_in"])
    # the reference launches this kernel with blockDim.y = rows of B (cu-matrix.cc:3054-3058): the tap count travels in Bl.y
    from kaldi_aslp_amd._lib import Dim3
    BL = Dim3(2, B.shape[0], 1)
    lib.cudaF_add_conv_mat_mat_elements(D3, BL, ops.ptr(d), ops.ptr(A), ops.ptr(B), ops.dim(d), ops.dim(A).stride, ops.dim(B).stride, 0.7, 0.3)
    check_error(); assert close(d.cpu().numpy(), g["conv_dst_out"])
    d = torch.zeros_like(d)
    lib.cudaF_add_conv_mat_mat_elements(D3, BL, ops.ptr(d), ops.ptr(A), ops.ptr(B), ops.dim(d), ops.dim(A).stride, ops.dim(B).stride, 1.0, 0.0)
    check_error(); assert close(d.cpu().numpy(), g["conv_dst_beta0"])
    d, a, b = t(g["mme_dst_in"]), t(g["mme_A"]), t(g["mme_B"])
    lib.cudaF_add_mat_mat_elements(D3, D3, ops.ptr(d), ops.ptr(a), ops.ptr(b), ops.dim(d), ops.dim(a).stride, ops.dim(b).stride, -1.25, 0.5)
    check_error(); assert close(d.cpu().numpy(), g["mme_dst_out"])
    v = t(g["mdv_vec"][0])
    d, m = t(g["mdv_dst_in"]), t(g["mdv_M"])
    lib.cudaF_add_mat_diag_vec(D3, D3, 0.75, ops.ptr(d), ops.dim(d), ops.ptr(m), ops.dim(m).stride, 1, ops.ptr(v), 1.0)
    check_error(); assert close(d.cpu().numpy(), g["mdv_dst_out"])
    d, m = t(g["mdv_dst_in_t"]), t(g["mdv_Mt"])
    lib.cudaF_add_mat_diag_vec(D3, D3, -0.5, ops.ptr(d), ops.dim(d), ops.ptr(m), 1, ops.dim(m).stride, ops.ptr(v), 1.0)   # kTrans (cu-matrix.cc:1166-1168)
    check_error(); assert close(d.cpu().numpy(), g["mdv_dst_out_t"])
    row, col = t(g["bc_row"][0]), t(g["bc_col"][0])
    d = t(g["bc_in"]); lib.cudaF_add_vec_to_rows(D3, D3, 0.5, ops.ptr(row), 1.0, ops.ptr(d), ops.dim(d)); check_error()
    assert close(d.cpu().numpy(), g["add_vec_to_rows"])
    d = t(g["bc_in"]); lib.cudaF_add_vec_to_cols(D3, D3, -1.5, ops.ptr(col), 1.0, ops.ptr(d), ops.dim(d)); check_error()
    assert close(d.cpu().numpy(), g["add_vec_to_cols"])
    d = t(g["bc_in"]); lib.cudaF_mul_cols_vec(D3, D3, ops.ptr(d), ops.ptr(row), ops.dim(d)); check_error()
    assert np.array_equal(d.cpu().numpy(), g["mul_cols_vec"])      # one rounding per element: exact
    d = t(g["bc_in"]); lib.cudaF_mul_rows_vec(D3, D3, ops.ptr(d), ops.ptr(col), ops.dim(d)); check_error()
    assert np.array_equal(d.cpu().numpy(), g["mul_rows_vec"])
    w, gr = t(g["l1_w_in"]), t(g["l1_g_in"])
    lib.cudaF_regularize_l1(D3, D3, ops.ptr(w), ops.ptr(gr), 0.002, 0.01, ops.dim(w), ops.dim(gr).stride); check_error()
    wn, gn = w.cpu().numpy(), gr.cpu().numpy()
    assert close(wn, g["l1_w_out"]) and close(gn, g["l1_g_out"])
    # which elements were clamped to zero: identical unless the sign test sits within an fma rounding of zero
    assert ((wn == 0) != (g["l1_w_out"] == 0)).sum() <= 1 and ((gn == 0) != (g["l1_g_out"] == 0)).sum() <= 1
    src, idx = t(g["cols_in"]), t(g["cols_idx"], torch.int32)
    d = t(g["cols_dst_in"]); lib.cudaF_copy_cols(D3, D3, ops.ptr(d), ops.ptr(src), ops.ptr(idx), ops.dim(d), ops.dim(src).stride); check_error()
    assert np.array_equal(d.cpu().numpy(), g["copy_cols_out"])
    d = t(g["cols_dst_in"]); lib.cudaF_add_cols(D3, D3, ops.ptr(d), ops.ptr(src), ops.ptr(idx), ops.dim(d), ops.dim(src).stride); check_error()
    assert np.array_equal(d.cpu().numpy(), g["add_cols_out"])
