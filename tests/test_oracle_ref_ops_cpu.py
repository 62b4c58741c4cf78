"""The C oracle's element-wise, arg-max and index operations against the REFERENCE's own CuMatrix CPU branch
(tests/golden/cumatrix_ops.bin).  This pins oracle rows a3 (Sigmoid / Tanh / ReLU and derivatives), a6 (Splice), a17
(Randomize) and the gather / arg-max helpers: index operations bit-exact, floating point to 1e-6 (same formulas, libm on
both sides)."""
import numpy as np

import cumatrix_golden


def close(a, b, tol=1e-6):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol


def test_oracle_matches_reference_cumatrix_cpu(oracle):
    g = cumatrix_golden.load()
    x, d = g["x"], g["d"]
    assert close(oracle.unary("orc_sigmoid", x), g["sigmoid"])
    assert close(oracle.unary("orc_tanh", x), g["tanh"])
    assert close(oracle.unary("orc_relu", x), g["relu"]) and np.array_equal(oracle.unary("orc_relu", x), g["relu"])
    assert close(oracle.binary("orc_diff_sigmoid", g["sigmoid"], d), g["diff_sigmoid"])
    assert close(oracle.binary("orc_diff_tanh", g["tanh"], d), g["diff_tanh"])
    assert np.array_equal(oracle.find_row_max_id(g["argmax_in"]), g["argmax"])
    assert np.array_equal(oracle.splice(g["splice_in"], g["splice_off"]), g["splice_out"])
    T, D = g["splice_in"].shape
    out = np.zeros((T, len(g["copy_cols"])), np.float32)
    oracle.lib.orc_copy_cols(out, out.shape[1], oracle.c32(g["splice_in"]), D, T, g["copy_cols"].astype(np.int32), len(g["copy_cols"]))
    assert np.array_equal(out, g["copy_out"])
    out = np.zeros((T, D), np.float32)
    oracle.lib.orc_randomize(out, D, oracle.c32(g["splice_in"]), D, D, g["rand_mask"].astype(np.int32), T)
    assert np.array_equal(out, g["randomize_out"])
    # the piecewise sigmoid / tanh at the extremes written into row 0 (kaldi-vector.cc:885-936): exact limits
    assert g["sigmoid"][0, 0] == 0.5 and g["sigmoid"][0, 6] == 1.0 and g["sigmoid"][0, 7] < 1e-37
    assert g["tanh"][0, 4] == 1.0 and g["tanh"][0, 5] == -1.0


def test_oracle_composite_arithmetic_matches_reference_cumatrix_cpu(oracle):
    """Round 2 pins: the BLAS-free arithmetic the components are composed from, as the reference's own CuMatrix CPU branch
    computes it -- CompactFsmn's product (AddConvMatMatElements, cu-matrix.cc:3037-3073), AddMatMatElements, the peephole term
    (AddMatDiagVec, beta = 1), the bias / scale broadcasts, cu::RegularizeL1 (the affine L1 step) and CopyCols / AddCols."""
    g = cumatrix_golden.load()
    L, c32 = oracle.lib, oracle.c32
    cp = lambda a: np.array(a, dtype=np.float32, order="C", copy=True)   # destinations are updated in place: never the fixture itself
    A, B = g["conv_A"], g["conv_B"]
    d = cp(g["conv_dst_in"])
    L.orc_add_conv_mat_mat_elements(d, d.shape[1], d.shape[1], c32(A), A.shape[1], A.shape[0], c32(B), B.shape[1], B.shape[0], 0.7, 0.3)
    assert close(d, g["conv_dst_out"])
    d = np.full_like(g["conv_dst_in"], np.nan)   # beta = 0 must not read the destination (0 * NaN)... the reference multiplies: keep it finite
    d[...] = 0
    L.orc_add_conv_mat_mat_elements(d, d.shape[1], d.shape[1], c32(A), A.shape[1], A.shape[0], c32(B), B.shape[1], B.shape[0], 1.0, 0.0)
    assert close(d, g["conv_dst_beta0"])
    d = cp(g["mme_dst_in"])
    R, Cc = d.shape
    L.orc_add_mat_mat_elements(d, Cc, c32(g["mme_A"]), Cc, c32(g["mme_B"]), Cc, R, Cc, -1.25, 0.5)
    assert close(d, g["mme_dst_out"])
    d = cp(g["mdv_dst_in"])
    L.orc_add_mat_diag_vec(d, Cc, c32(g["mdv_M"]), Cc, 1, c32(g["mdv_vec"][0]), R, Cc, 0.75)
    assert close(d, g["mdv_dst_out"])
    d = cp(g["mdv_dst_in_t"])
    L.orc_add_mat_diag_vec(d, Cc, c32(g["mdv_Mt"]), 1, R, c32(g["mdv_vec"][0]), R, Cc, -0.5)   # kTrans: strides swapped
    assert close(d, g["mdv_dst_out_t"])
    m = g["bc_in"]
    r2, c2 = m.shape
    d = cp(m); L.orc_add_vec_to_rows(d, c2, c32(g["bc_row"][0]), r2, c2, 0.5); assert close(d, g["add_vec_to_rows"])
    d = cp(m); L.orc_add_vec_to_cols(d, c2, c32(g["bc_col"][0]), r2, c2, -1.5); assert close(d, g["add_vec_to_cols"])
    d = cp(m); L.orc_mul_cols_vec(d, c2, c32(g["bc_row"][0]), r2, c2); assert close(d, g["mul_cols_vec"])
    d = cp(m); L.orc_mul_rows_vec(d, c2, c32(g["bc_col"][0]), r2, c2); assert close(d, g["mul_rows_vec"])
    w, gr = cp(g["l1_w_in"]), cp(g["l1_g_in"])
    L.orc_regularize_l1(w, Cc, gr, Cc, R, Cc, 0.002, 0.01)
    assert close(w, g["l1_w_out"]) and close(gr, g["l1_g_out"])
    assert np.array_equal(w == 0, g["l1_w_out"] == 0) and np.array_equal(gr == 0, g["l1_g_out"] == 0)
    assert (g["l1_w_out"] == 0).sum() > (g["l1_w_in"] == 0).sum() > 0      # the fixture does exercise both branches
    src, idx = g["cols_in"], g["cols_idx"].astype(np.int32)
    d = cp(g["cols_dst_in"]); L.orc_copy_cols_idx(d, len(idx), c32(src), src.shape[1], src.shape[0], idx, len(idx))
    assert np.array_equal(d, g["copy_cols_out"])
    d = cp(g["cols_dst_in"]); L.orc_add_cols_idx(d, len(idx), c32(src), src.shape[1], src.shape[0], idx, len(idx))
    assert close(d, g["add_cols_out"])
