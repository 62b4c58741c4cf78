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
