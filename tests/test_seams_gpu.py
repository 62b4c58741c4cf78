"""Seams B2 and B3 of SURVEY.md §8b through ctypes: the cuBLAS-signature, column-major BLAS entry points (include/aslp_blas.h,
reference src/aslp-cudamatrix/cublas-wrappers.h:28-133) and the C view of CuDevice (include/aslp_device.h, reference
cu-device.h:43-151).  The gemm is called exactly the way CuMatrixBase::AddMatMat calls cuBLAS (cu-matrix.cc:1049-1053:
operands swapped, row-major data handed over as column-major) and compared with a float64 product."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N, T = 0, 1


def sigs(lib):
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    lib.aslp_blas_create.argtypes = [C.POINTER(vp)]
    lib.aslp_blas_destroy.argtypes = [vp]
    lib.aslp_blas_set_stream.argtypes = [vp, vp]
    lib.aslp_blas_sgemm.argtypes = [vp, i, i, i, i, i, f, vp, i, vp, i, f, vp, i]
    lib.aslp_blas_sger.argtypes = [vp, i, i, f, vp, i, vp, i, vp, i]
    lib.aslp_blas_sgemv.argtypes = [vp, i, i, i, f, vp, i, vp, i, f, vp, i]
    lib.aslp_blas_sdot.argtypes = [vp, i, vp, i, vp, i, C.POINTER(f)]
    lib.aslp_blas_saxpy.argtypes = [vp, i, f, vp, i, vp, i]
    lib.aslp_blas_sscal.argtypes = [vp, i, f, vp, i]
    lib.aslp_blas_scopy.argtypes = [vp, i, vp, i, vp, i]
    for n in ("create", "destroy", "set_stream", "sgemm", "sger", "sgemv", "sdot", "saxpy", "sscal", "scopy"):
        getattr(lib, "aslp_blas_" + n).restype = i


@pytest.fixture()
def blas(aslp, dev):
    sigs(aslp.lib)
    h = C.c_void_p()
    assert aslp.lib.aslp_blas_create(C.byref(h)) == 0
    yield aslp.lib, h
    aslp.lib.aslp_blas_destroy(h)


def dptr(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("ta,tb", [(N, N), (N, T), (T, N), (T, T)])
@pytest.mark.parametrize("shape", [(37, 53, 29), (256, 2048, 440), (64, 8, 300)])
def test_sgemm_as_cumatrix_addmatmat_calls_it(blas, dev, ta, tb, shape):
    """AddMatMat(alpha, A, transA, B, transB, beta) on ROW-major data: cublas_gemm(handle, transb, transa, m = C.cols, n = C.rows,
    k, alpha, B, B.stride, A, A.stride, beta, C, C.stride)  (cu-matrix.cc:1040-1053)."""
    lib, h = blas
    rows, cols, k = shape
    rng = np.random.default_rng(rows + cols)
    A = rng.standard_normal((k, rows) if ta else (rows, k)).astype(np.float32)
    B = rng.standard_normal((cols, k) if tb else (k, cols)).astype(np.float32)
    Cm = rng.standard_normal((rows, cols)).astype(np.float32)
    alpha, beta = 0.75, -0.5
    ref = alpha * ((A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)) + beta * Cm
    # padded strides, as CuMatrix allocates them
    def pad(x):
        st = (x.shape[1] + 15) // 16 * 16
        t = torch.zeros(x.shape[0], st, device=dev)
        t[:, :x.shape[1]] = torch.from_numpy(x).to(dev)
        return t, st
    At, lda = pad(A)
    Bt, ldb = pad(B)
    Ct, ldc = pad(Cm)
    rc = lib.aslp_blas_sgemm(h, tb, ta, cols, rows, k, alpha, dptr(Bt), ldb, dptr(At), lda, beta, dptr(Ct), ldc)
    assert rc == 0
    got = Ct[:, :cols].cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


def test_column_major_sgemm_plain_blas_semantics(blas, dev):
    """the same entry point read as textbook BLAS: column-major C[m x n] = op(A) op(B)"""
    lib, h = blas
    m, n, k = 19, 23, 31
    rng = np.random.default_rng(3)
    for ta in (N, T):
        for tb in (N, T):
            A = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)   # logical (as stored, before op)
            B = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
            # column-major storage of a logical [r x c] matrix = the bytes of its transpose in C order
            At = torch.from_numpy(np.ascontiguousarray(A.T)).to(dev)
            Bt = torch.from_numpy(np.ascontiguousarray(B.T)).to(dev)
            Ct = torch.zeros(n, m, device=dev)
            assert lib.aslp_blas_sgemm(h, ta, tb, m, n, k, 1.0, dptr(At), A.shape[0], dptr(Bt), B.shape[0], 0.0, dptr(Ct), m) == 0
            ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
            assert np.abs(Ct.cpu().numpy().T - ref).max() <= 1e-4 * np.abs(ref).max()


def test_level1_and_level2(blas, dev):
    lib, h = blas
    rng = np.random.default_rng(9)
    n = 1000
    x = rng.standard_normal(3 * n).astype(np.float32)
    y = rng.standard_normal(2 * n).astype(np.float32)
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    assert lib.aslp_blas_saxpy(h, n, 0.5, dptr(xt), 3, dptr(yt), 2) == 0           # strided, as CuVector rows / columns need it
    yr = y.copy(); yr[::2] += np.float32(0.5) * x[::3]
    assert np.allclose(yt.cpu().numpy(), yr, rtol=1e-6, atol=1e-6)
    assert lib.aslp_blas_sscal(h, n, -2.0, dptr(yt), 2) == 0
    yr[::2] *= np.float32(-2.0)
    assert np.array_equal(yt.cpu().numpy(), yr)
    zt = torch.zeros(n, device=dev)
    assert lib.aslp_blas_scopy(h, n, dptr(xt), 3, dptr(zt), 1) == 0
    assert np.array_equal(zt.cpu().numpy(), x[::3])
    res = C.c_float()
    assert lib.aslp_blas_sdot(h, n, dptr(xt), 3, dptr(yt), 2, C.byref(res)) == 0
    assert abs(res.value - float(x[::3].astype(np.float64) @ yr[::2].astype(np.float64))) < 1e-3 * n ** 0.5
    # ger / gemv on a column-major [m x n] matrix with lda > m
    m, nn, lda = 70, 45, 80
    A = rng.standard_normal((nn, lda)).astype(np.float32)      # C-order bytes of the column-major matrix (column j = row j here)
    At = torch.from_numpy(A).to(dev)
    u, v = rng.standard_normal(m).astype(np.float32), rng.standard_normal(nn).astype(np.float32)
    ut, vt = torch.from_numpy(u).to(dev), torch.from_numpy(v).to(dev)
    assert lib.aslp_blas_sger(h, m, nn, 0.3, dptr(ut), 1, dptr(vt), 1, dptr(At), lda) == 0
    Ar = A.copy(); Ar[:, :m] += np.float32(0.3) * np.outer(v, u)
    assert np.allclose(At.cpu().numpy(), Ar, rtol=1e-5, atol=1e-6)
    Acm = Ar[:, :m].T.astype(np.float64)                        # the logical m x n matrix
    out = torch.from_numpy(rng.standard_normal(m).astype(np.float32)).to(dev)
    o0 = out.cpu().numpy().copy()
    assert lib.aslp_blas_sgemv(h, N, m, nn, 2.0, dptr(At), lda, dptr(vt), 1, 0.5, dptr(out), 1) == 0
    assert np.allclose(out.cpu().numpy(), 2.0 * Acm @ v + 0.5 * o0, rtol=1e-4, atol=1e-4)
    out2 = torch.zeros(nn, device=dev)
    assert lib.aslp_blas_sgemv(h, T, m, nn, 1.0, dptr(At), lda, dptr(ut), 1, 0.0, dptr(out2), 1) == 0
    assert np.allclose(out2.cpu().numpy(), Acm.T @ u, rtol=1e-4, atol=1e-4)


def test_cu_device_c_view(aslp, dev):
    lib = aslp.lib
    lib.aslp_device_last_error.restype = C.c_char_p
    lib.aslp_device_malloc.restype = C.c_void_p
    lib.aslp_device_malloc.argtypes = [C.c_size_t]
    lib.aslp_device_malloc_pitch.restype = C.c_void_p
    lib.aslp_device_malloc_pitch.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.aslp_device_free.argtypes = [C.c_void_p]
    lib.aslp_device_accu_profile.argtypes = [C.c_char_p, C.c_double]
    lib.aslp_device_get_free_memory.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    # no CPU engine behind this library: "no" is an error with a message, so is a bad flag value (KALDI_ERR in the reference)
    assert lib.aslp_device_select_gpu_id(b"no") != 0 and b"no CPU compute path" in lib.aslp_device_last_error()
    assert lib.aslp_device_select_gpu_id(b"maybe") != 0 and b"yes|no|optional" in lib.aslp_device_last_error()
    assert lib.aslp_device_set_gpu_id(9999) != 0 and b"Invalid gpu id" in lib.aslp_device_last_error()
    if not lib.aslp_device_enabled():
        assert lib.aslp_device_select_gpu_id(b"yes") == 0, lib.aslp_device_last_error()
    assert lib.aslp_device_enabled() == 1 and lib.aslp_device_active_gpu_id() >= 0
    assert lib.aslp_device_set_gpu_id(0) != 0 and b"already an active GPU" in lib.aslp_device_last_error()   # cu-device.cc:205-208
    assert lib.aslp_device_check_gpu_health() == 0, lib.aslp_device_last_error()
    pitch = C.c_size_t()
    p = lib.aslp_device_malloc_pitch(440 * 4, 1024, C.byref(pitch))
    assert p and pitch.value >= 440 * 4 and pitch.value % 64 == 0
    t = torch.zeros(8, device=dev)
    odd = (1 << 20) + 77 * 4096               # a size nothing else in this process has cached
    q = lib.aslp_device_malloc(odd)
    assert q and q != p
    lib.aslp_device_free(q)
    q2 = lib.aslp_device_malloc(odd)           # the caching allocator hands the block back (cu-allocator.h:67-70)
    assert q2 == q
    lib.aslp_device_free(q2)
    lib.aslp_device_free(p)
    lib.aslp_device_accu_profile(b"AddMatMat", 1.5)
    lib.aslp_device_print_profile()
    buf = C.create_string_buffer(256)
    fr, tot = C.c_longlong(), C.c_longlong()
    assert lib.aslp_device_get_free_memory(buf, 256, C.byref(fr), C.byref(tot)) == 0
    assert b"free:" in buf.value and 0 < fr.value <= tot.value
    del t
