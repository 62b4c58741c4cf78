"""The oracle's restatement of the front-end components (oracle/aslp_oracle_conv.c) against (1) the same op sequences issued on the
REFERENCE's own CuMatrix library (tests/golden/component_ops.bin, generator oracle/gen_component_golden.cpp) and (2) the known answers
the reference's own unit test holds (src/aslp-nnet/nnet-component-test.cc:53-206, tests/golden/component_known_answers.json).
Tolerances: 2e-6 element-wise of max(1, |ref|) where no BLAS sum is involved (bit-exact for the index / mask ops), 2e-5 behind products."""
import re

import numpy as np

import cumatrix_golden
import oracle_lib as oracle


def close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol


def test_linear_transform_matches_reference_library():
    g = cumatrix_golden.load_components()
    lr, mmt, l2, l1, coef = [float(v) for v in g["lin_opts"]]
    lin = oracle.Linear(g["lin_W0"])
    for step in (0, 1):
        x, od = g["lin_in%d" % step], g["lin_od%d" % step]
        assert close(lin.propagate(x), g["lin_out%d" % step], 2e-6)
        assert close(lin.backpropagate(od), g["lin_id%d" % step], 2e-6)
        lin.update(x, od, lr, mmt, l2, l1, coef)
        assert close(lin.corr, g["lin_corr%d" % step], 5e-6)
        assert close(lin.W, g["lin_W%d" % (step + 1)], 2e-6)


def test_convolutional_component_matches_reference_library():
    g = cumatrix_golden.load_components()
    in_dim, F, pd, ps, pst, lr, coef, bcoef, max_norm = [float(v) for v in g["conv_geom"]]
    conv = oracle.Conv(g["conv_filters0"], g["conv_bias0"], int(in_dim), int(pd), int(ps), int(pst))
    assert conv.F == int(F)
    for step in (0, 1):
        x, od = g["conv_in%d" % step], g["conv_od%d" % step]
        assert close(conv.propagate(x), g["conv_out%d" % step], 5e-6)
        assert close(conv.backpropagate(od), g["conv_id%d" % step], 5e-6)
        conv.update(od, lr, coef, bcoef, max_norm)
        assert close(conv.fgrad, g["conv_fgrad%d" % step], 5e-6) and close(conv.bgrad, g["conv_bgrad%d" % step], 5e-6)
        assert close(conv.filters, g["conv_filters%d" % (step + 1)], 5e-6) and close(conv.bias, g["conv_bias%d" % (step + 1)], 5e-6)
    # the max-norm did act on this fixture (some filter rows were longer than 1.5)
    assert np.any(np.abs(np.linalg.norm(g["conv_filters2"], axis=1) - max_norm) < 1e-4)


def test_max_pooling_matches_reference_library():
    g = cumatrix_golden.load_components()
    _, size, step, stride = [int(v) for v in g["pool_geom"]]
    out = oracle.max_pool(g["pool_in"], size, step, stride)
    assert np.array_equal(out, g["pool_out"])
    idf = oracle.max_pool_backprop(g["pool_in"], out, g["pool_od"], size, step, stride)
    assert np.array_equal(idf, g["pool_id"])          # masks, sums in pool order and one scale: the reference's bits
    assert (np.sum(np.isclose(g["pool_in"][:, None, :], g["pool_in"][:, :, None]).sum()) > 0)   # (the fixture has ties)


def test_length_norm_matches_reference_library():
    g = cumatrix_golden.load_components()
    for w in (0, 1):
        out, sc = oracle.length_norm(g["ln%d_in" % w])
        assert close(sc, g["ln%d_scales" % w], 1e-6) and close(out, g["ln%d_out" % w], 1e-6)
        assert close(oracle.length_norm_backprop(g["ln%d_od" % w], sc), g["ln%d_id" % w], 1e-6)


def test_group_pnorm_and_max_match_reference_library():
    g = cumatrix_golden.load_components()
    x, od = g["grp_in"], g["grp_od"]
    for i, p in enumerate((2.0, 1.0, 3.0)):
        y = oracle.group_pnorm(x, od.shape[1], p)
        assert close(y, g["pnorm%d_out" % i], 1e-6), p
        d = oracle.group_pnorm_deriv(x, y, p)
        assert close(d, g["pnorm%d_deriv" % i], 2e-6), p
        assert close(oracle.mul_rows_group_mat(d, od), g["pnorm%d_id" % i], 2e-6), p
    y = oracle.group_max(x, od.shape[1])
    assert np.array_equal(y, g["gmax_out"])
    d = oracle.group_max_deriv(x, y)
    assert np.array_equal(d, g["gmax_deriv"]) and np.array_equal(oracle.mul_rows_group_mat(d, od), g["gmax_id"])


def parse_component(text):
    """'<Marker> dim_out dim_in payload...' of the reference's (pre-ASLP) nnet text format -> marker, dims, {token: value}"""
    toks = text.split()
    marker, dout, din = toks[0], int(toks[1]), int(toks[2])
    rest = " ".join(toks[3:])
    fields = {}
    for m in re.finditer(r"(<\w+>)\s*(\[[^\]]*\]|[-+.\deE]+)", rest):
        v = m.group(2)
        if v.startswith("["):
            rows = [[float(t) for t in r.split()] for r in v[1:-1].split(";")]
            fields[m.group(1)] = np.asarray([r for r in rows if r], np.float32)
        else:
            fields[m.group(1)] = float(v)
    return marker, dout, din, fields


def test_reference_known_answers():
    ka = cumatrix_golden.load_known_answers()
    # UnitTestLengthNorm (:53-75): rows of the output have unit length
    out, _ = oracle.length_norm(ka["UnitTestLengthNorm"]["matrices"]["mat_in"])
    assert np.allclose(np.sqrt((out.astype(np.float64) ** 2).sum(1)), 1.0, atol=1e-6)
    # UnitTestConvolutionalComponentUnity (:77-103): identity filter, out == in, in_diff == out_diff
    t = ka["UnitTestConvolutionalComponentUnity"]
    _, dout, din, f = parse_component(t["component"])
    conv = oracle.Conv(f["<Filters>"], f["<Bias>"].ravel(), din, int(f["<PatchDim>"]), int(f["<PatchStep>"]), int(f["<PatchStride>"]))
    x = t["matrices"]["mat_in"]
    assert np.array_equal(conv.propagate(x), x) and np.array_equal(conv.backpropagate(x), x)
    # UnitTestConvolutionalComponent3x3 (:105-139): zero output, hand-computed in-diff
    t = ka["UnitTestConvolutionalComponent3x3"]
    _, dout, din, f = parse_component(t["component"])
    conv = oracle.Conv(f["<Filters>"], f["<Bias>"].ravel(), din, int(f["<PatchDim>"]), int(f["<PatchStep>"]), int(f["<PatchStride>"]))
    m = t["matrices"]
    assert conv.F * conv.P == dout
    assert np.array_equal(conv.propagate(m["mat_in"]), m["mat_out_ref"])
    assert np.array_equal(conv.backpropagate(m["mat_out_diff"]), m["mat_in_diff_ref"])
    # UnitTestMaxPoolingComponent (:143-206)
    t = ka["UnitTestMaxPoolingComponent"]
    opts = dict(re.findall(r"<(\w+)> (\d+)", t["component"]))
    size, step, stride = int(opts["PoolSize"]), int(opts["PoolStep"]), int(opts["PoolStride"])
    m = t["matrices"]
    out = oracle.max_pool(m["mat_in"], size, step, stride)
    assert np.array_equal(out, m["mat_out_ref"])
    assert np.array_equal(oracle.max_pool_backprop(m["mat_in"], out, np.ones_like(out), size, step, stride), m["mat_in_diff_ref"])
