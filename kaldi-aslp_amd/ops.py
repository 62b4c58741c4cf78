"""torch.Tensor wrappers over the kernel C ABI (B1/B2).  Tensors must live on the GPU; the
wrappers only extract (pointer, rows, cols, stride) -- all arithmetic happens in
libaslp_hip.so.  Used by tests/, bench.py and the Python mirror of the sync workers."""
import ctypes as C

import torch

from ._lib import lib, MatrixDim, Dim3, GemmEpilogue, CtcComputeInfo, D3, check_error


def _chk(t, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("aslp ops need device tensors (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))
    if t.dim() == 2 and t.stride(1) != 1:
        raise ValueError("matrix must be row-major with unit column stride")
    return t


def dim(t):
    """MatrixDim of a 2-D row-major tensor (or view with stride >= cols)."""
    if t.dim() == 1:
        return MatrixDim(1, t.shape[0], t.shape[0])
    return MatrixDim(t.shape[0], t.shape[1], t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def use_torch_stream():
    """Enqueue library kernels on torch's current stream (so torch.cuda.Event sees them)."""
    lib.aslp_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))


def sgemm(transA, transB, alpha, A, B, beta, Cm, epilogue=None):
    """C = alpha*op(A)*op(B) + beta*C (cu-matrix.cc:1027-1061 semantics)."""
    _chk(A), _chk(B), _chk(Cm)
    M, N = Cm.shape
    K = A.shape[0] if transA else A.shape[1]
    ep = C.byref(epilogue) if epilogue is not None else None
    rc = lib.aslp_sgemm_ex(int(transA), int(transB), M, N, K, alpha, ptr(A), dim(A).stride, ptr(B), dim(B).stride,
                           beta, ptr(Cm), dim(Cm).stride, ep)
    if rc != 0:
        raise ValueError("aslp_sgemm argument error %d" % rc)
    check_error()


class Planes:
    """the two fp16 planes of an fp32 matrix (aslp_planes_*): made once, read by every product the matrix takes part in"""

    def __init__(self, t):
        self.h = C.c_void_p(lib.aslp_planes_new())
        self.t = _chk(t)
        rc = lib.aslp_planes_convert(self.h, ptr(t), dim(t))
        if rc != 0:
            raise ValueError("aslp_planes_convert error %d" % rc)
        check_error()

    def __del__(self):
        if getattr(self, "h", None):
            lib.aslp_planes_free(self.h)
            self.h = None


def sgemm_planes(transA, transB, alpha, A, pa, B, pb, beta, Cm, epilogue=None):
    """sgemm with prepared planes of A and / or B (Planes or None)"""
    _chk(A), _chk(B), _chk(Cm)
    M, N = Cm.shape
    K = A.shape[0] if transA else A.shape[1]
    ep = C.byref(epilogue) if epilogue is not None else None
    rc = lib.aslp_sgemm_planes_ex(int(transA), int(transB), M, N, K, alpha, ptr(A), dim(A).stride, pa.h if pa else None, ptr(B), dim(B).stride,
                                  pb.h if pb else None, beta, ptr(Cm), dim(Cm).stride, ep)
    if rc != 0:
        raise ValueError("aslp_sgemm_planes argument error %d" % rc)
    check_error()


def sgemm_pair(transA, transB, alpha, A0, A1, B0, B1, beta, C0, C1, ep0=None, ep1=None):
    """two products of one shape in one launch (aslp_sgemm_pair_ex)"""
    for t in (A0, A1, B0, B1, C0, C1):
        _chk(t)
    assert A0.shape == A1.shape and B0.shape == B1.shape and C0.shape == C1.shape
    assert dim(A0).stride == dim(A1).stride and dim(B0).stride == dim(B1).stride and dim(C0).stride == dim(C1).stride
    M, N = C0.shape
    K = A0.shape[0] if transA else A0.shape[1]
    rc = lib.aslp_sgemm_pair_ex(int(transA), int(transB), M, N, K, alpha, ptr(A0), ptr(A1), dim(A0).stride, ptr(B0), ptr(B1), dim(B0).stride,
                                beta, ptr(C0), ptr(C1), dim(C0).stride, C.byref(ep0) if ep0 is not None else None,
                                C.byref(ep1) if ep1 is not None else None)
    if rc != 0:
        raise ValueError("aslp_sgemm_pair argument error %d" % rc)
    check_error()


def sigmoid(y, x):
    lib.cudaF_sigmoid(D3, D3, ptr(_chk(y)), ptr(_chk(x)), dim(y), dim(x).stride); check_error()


def tanh(y, x):
    lib.cudaF_tanh(D3, D3, ptr(_chk(y)), ptr(_chk(x)), dim(y), dim(x).stride); check_error()


def diff_sigmoid(eout, y, e):
    lib.cudaF_diff_sigmoid(D3, D3, ptr(_chk(eout)), ptr(_chk(e)), ptr(_chk(y)), dim(eout), dim(e).stride, dim(y).stride); check_error()


def diff_tanh(eout, y, e):
    lib.cudaF_diff_tanh(D3, D3, ptr(_chk(eout)), ptr(_chk(e)), ptr(_chk(y)), dim(eout), dim(e).stride, dim(y).stride); check_error()


def softmax(y, x):
    lib.cudaF_softmax_reduce(0, 0, ptr(_chk(y)), ptr(_chk(x)), dim(y), dim(x).stride); check_error()


def splice(y, x, offsets):
    lib.cudaF_splice(D3, D3, ptr(_chk(y)), ptr(_chk(x)), ptr(_chk(offsets, torch.int32)), dim(y), dim(x)); check_error()


def splice_backward(in_diff, out_diff, offsets):
    lib.aslp_splice_backward(ptr(_chk(in_diff)), dim(in_diff), ptr(_chk(out_diff)), dim(out_diff).stride,
                             ptr(_chk(offsets, torch.int32)), offsets.numel()); check_error()


def randomize(y, x, copy_from):
    d_out = dim(y); d_out.rows = copy_from.numel()
    d_in = dim(x); d_in.rows = copy_from.numel()
    lib.cudaF_randomize(D3, D3, ptr(_chk(y)), ptr(_chk(x)), ptr(_chk(copy_from, torch.int32)), d_out, d_in); check_error()


def find_row_max_id(m):
    out = torch.empty(m.shape[0], dtype=torch.int32, device=m.device)
    lib.aslp_find_row_max_id(ptr(_chk(m)), dim(m), ptr(out)); check_error()
    return out


def add_row_sum_mat_vec(alpha, M, beta, v):
    lib.aslp_add_row_sum_mat_vec(alpha, ptr(_chk(M)), dim(M), beta, ptr(_chk(v))); check_error()


def rnn_vec_grads(jobs, d_stride, rows, mmt, clip, neg_lr):
    """jobs: list of (d_view, x_view or None, corr, param); d_view / x_view are [rows x n] column blocks of pitched buffers."""
    from ._lib import RnnVecGrad
    arr = (RnnVecGrad * len(jobs))()
    for k, (d, x, corr, param) in enumerate(jobs):
        arr[k].d = ptr(d); arr[k].x = ptr(x) if x is not None else None
        arr[k].ldx = x.stride(0) if x is not None else 0
        arr[k].n = d.shape[1]; arr[k].corr = ptr(_chk(corr)); arr[k].param = ptr(_chk(param))
    lib.aslp_rnn_vec_grads(arr, len(jobs), d_stride, rows, mmt, clip, neg_lr); check_error()


def bn_forward(x, out, xhat, scale, shift, mean, inv_std, acc_means=None, acc_vars=None, var_floor=1e-7):
    lib.aslp_bn_forward(ptr(_chk(x)), dim(x), ptr(_chk(out)), dim(out).stride, ptr(_chk(xhat)), dim(xhat).stride,
                        ptr(scale), ptr(shift), ptr(mean), ptr(inv_std), ptr(acc_means), ptr(acc_vars), var_floor)
    check_error()


def bn_backward(x, dy, xhat, scale, mean, inv_std, dscale, dshift, momentum, in_diff):
    lib.aslp_bn_backward(ptr(_chk(x)), dim(x), ptr(_chk(dy)), dim(dy).stride, ptr(_chk(xhat)), dim(xhat).stride,
                         ptr(scale), ptr(mean), ptr(inv_std), ptr(dscale), ptr(dshift), momentum,
                         ptr(in_diff), dim(in_diff).stride if in_diff is not None else 0)
    check_error()


def xent_eval(net_out, frame_weights, diff, stats, targets=None, labels=None):
    lib.aslp_xent_eval(ptr(_chk(net_out)), dim(net_out), ptr(targets), dim(targets).stride if targets is not None else 0,
                       ptr(labels), ptr(_chk(frame_weights)), ptr(_chk(diff)), dim(diff).stride,
                       ptr(_chk(stats, torch.float64)))
    check_error()


def xent_eval_rows(net_out, frame_weights, diff, rowstats, labels, softmax=False):
    """aslp_xent_eval_rows: the loss diff and the batch's per-row statistics (rowstats [rows x 5] float64); the sums are taken later"""
    lib.aslp_xent_eval_rows(ptr(_chk(net_out)), dim(net_out), ptr(labels), ptr(_chk(frame_weights)), ptr(_chk(diff)), dim(diff).stride,
                            ptr(_chk(rowstats, torch.float64)), int(bool(softmax)), None)
    check_error()


def xent_sum_rowstats(rowstats, rows, batches, stats):
    """aslp_xent_sum_rowstats: `batches` blocks of `rows` per-row statistics added to the five accumulators, in order"""
    lib.aslp_xent_sum_rowstats(ptr(_chk(rowstats, torch.float64)), rows, batches, ptr(_chk(stats, torch.float64)))
    check_error()


def ctc_loss(acts, labels, input_lengths, want_grad=True):
    """compute_ctc_loss (warp-ctc/include/ctc.h:88-97) on device activations [(maxT*mb) x A] laid out (t, n, p).
    labels: list of int lists; returns (costs numpy[mb], grads tensor or None)."""
    import numpy as np
    _chk(acts)
    mb = len(labels)
    A = acts.shape[-1]
    flat = [int(v) for l in labels for v in l] or [0]
    flat_c = (C.c_int * len(flat))(*flat)
    lab_len = (C.c_int * mb)(*[len(l) for l in labels])
    in_len = (C.c_int * mb)(*[int(t) for t in input_lengths])
    info = CtcComputeInfo(1, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    sz = C.c_size_t()
    st = lib.get_workspace_size(lab_len, in_len, A, mb, info, C.byref(sz))
    if st != 0:
        raise RuntimeError("get_workspace_size: " + lib.ctcGetStatusString(st).decode())
    ws = torch.empty(sz.value, dtype=torch.uint8, device=acts.device)
    grads = torch.zeros_like(acts) if want_grad else None
    costs = (C.c_float * mb)()
    st = lib.compute_ctc_loss(ptr(acts), ptr(grads), flat_c, lab_len, in_len, A, mb, costs, ptr(ws), info)
    if st != 0:
        raise RuntimeError("compute_ctc_loss: " + lib.ctcGetStatusString(st).decode())
    check_error()
    return np.array(list(costs), np.float32), grads


# ---- depthwise temporal filters (csrc/temporal.hip; nnet-cfsmn-component.h:170-262, nnet-row-convolution.cc:105-186) ----
def fsmn_forward(out, x, coef, past, future):
    """out[t] = x[t] + sum_j coef[j] .* x[t + j - past]; coef [(past + future + 1) x D]"""
    lib.aslp_fsmn_filter(ptr(_chk(out)), dim(out).stride, ptr(_chk(x)), dim(x).stride, ptr(_chk(coef)), dim(coef).stride, x.shape[1], past, future,
                         x.shape[0], 0)
    check_error()


def fsmn_backward(in_diff, coef_corr, coef, x, out_diff, past, future, clip=0.0, lr=0.0):
    """one launch: in_diff (reversed filter over out_diff), coef_corr (tap gradients, clipped), and with lr != 0 coef += -lr coef_corr"""
    lib.aslp_fsmn_backward(ptr(_chk(in_diff)), dim(in_diff).stride, ptr(_chk(coef_corr)), dim(coef_corr).stride, ptr(_chk(coef)), dim(coef).stride,
                           ptr(_chk(x)), dim(x).stride, ptr(_chk(out_diff)), dim(out_diff).stride, x.shape[1], past, future, x.shape[0], clip, lr)
    check_error()


def rowconv_forward(out, x, w, seq_len, K):
    """rows t * S + s; w dense [D x (K + 1)]; seq_len int32 [S] on the device"""
    S = seq_len.numel()
    lib.aslp_rowconv_forward(ptr(_chk(out)), dim(out).stride, ptr(_chk(x)), dim(x).stride, ptr(w), x.shape[1], K, x.shape[0] // S, S, ptr(seq_len))
    check_error()


def rowconv_backward(in_diff, w_diff, x, out_diff, w, seq_len, K, w_corr=None, momentum=0.0, lr=0.0):
    """in_diff and the tap gradients from one pass over x / out_diff; with w_corr given also the momentum + SGD step"""
    S = seq_len.numel()
    lib.aslp_rowconv_backward_fused(ptr(_chk(in_diff)), dim(in_diff).stride, ptr(w_diff), ptr(_chk(x)), dim(x).stride, ptr(_chk(out_diff)),
                                    dim(out_diff).stride, ptr(w), x.shape[1], K, x.shape[0] // S, S, ptr(seq_len), ptr(w_corr), momentum, lr,
                                    1 if w_corr is not None else 0)
    check_error()

