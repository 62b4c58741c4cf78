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
