"""ctypes/numpy binding of oracle/_build/libaslp_oracle.so (the CPU restatement of the
reference) and oracle/_ref/libwarpctc_ref.so (the reference's own Warp-CTC CPU code).
TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline use it."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "_build", "libaslp_oracle.so")
REF_CTC_SO = os.path.join(ODIR, "_ref", "libwarpctc_ref.so")


def build(force=False):
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(SO)) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ODIR], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/warp-ctc") and not os.path.exists(REF_CTC_SO):
        subprocess.check_call(["make", "-C", ODIR, "ref"], stdout=subprocess.DEVNULL)


build()
lib = C.CDLL(SO)
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i, _f = C.c_int, C.c_float


def _sig(name, restype, *args):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(args)
    return fn


_sig("orc_add_mat_mat", None, f32p, _i, _i, _i, _f, f32p, _i, _i, f32p, _i, _i, _i, _f)
_sig("orc_set_num_threads", None, _i)
_sig("orc_get_num_threads", _i)
for n in ("orc_sigmoid", "orc_tanh", "orc_softmax_rows", "orc_relu"):
    _sig(n, None, f32p, _i, f32p, _i, _i, _i)
for n in ("orc_diff_sigmoid", "orc_diff_tanh", "orc_diff_relu"):
    _sig(n, None, f32p, _i, f32p, _i, f32p, _i, _i, _i)
_sig("orc_find_row_max_id", None, f32p, _i, _i, _i, i32p)
_sig("orc_splice", None, f32p, _i, f32p, _i, _i, _i, i32p, _i)
_sig("orc_splice_backpropagate", None, f32p, _i, f32p, _i, _i, _i, i32p, _i)
_sig("orc_copy_cols", None, f32p, _i, f32p, _i, _i, i32p, _i)
_sig("orc_randomize", None, f32p, _i, f32p, _i, _i, i32p, _i)
_sig("orc_add_row_sum_mat", None, f32p, _i, _i, _i, f32p, _i, _i, _f, _f)
_sig("orc_add_conv_mat_mat_elements", None, f32p, _i, _i, f32p, _i, _i, f32p, _i, _i, _f, _f)
_sig("orc_regularize_l1", None, f32p, _i, f32p, _i, _i, _i, _f, _f)
_sig("orc_add_mat_mat_elements", None, f32p, _i, f32p, _i, f32p, _i, _i, _i, _f, _f)
_sig("orc_add_mat_diag_vec", None, f32p, _i, f32p, _i, _i, f32p, _i, _i, _f)
_sig("orc_add_vec_to_rows", None, f32p, _i, f32p, _i, _i, _f)
_sig("orc_add_vec_to_cols", None, f32p, _i, f32p, _i, _i, _f)
_sig("orc_mul_cols_vec", None, f32p, _i, f32p, _i, _i)
_sig("orc_mul_rows_vec", None, f32p, _i, f32p, _i, _i)
_sig("orc_copy_cols_idx", None, f32p, _i, f32p, _i, _i, i32p, _i)
_sig("orc_add_cols_idx", None, f32p, _i, f32p, _i, _i, i32p, _i)
_sig("orc_affine_propagate", None, f32p, _i, f32p, _i, _i, f32p, _i, f32p, _i, _i)
_sig("orc_affine_backpropagate", None, f32p, _i, f32p, _i, _i, f32p, _i, _i, _i)


class AffineOpts(C.Structure):
    _fields_ = [(n, _f) for n in ("learn_rate", "momentum", "l2_penalty", "l1_penalty", "learn_rate_coef",
                                  "bias_learn_rate_coef", "max_norm")]


_sig("orc_affine_update", None, f32p, _i, f32p, f32p, _i, f32p, f32p, _i, f32p, _i, _i, _i, _i, C.POINTER(AffineOpts))


class BnState(C.Structure):
    _fields_ = [("dim", _i), ("scale", C.c_void_p), ("shift", C.c_void_p), ("dscale", C.c_void_p),
                ("dshift", C.c_void_p), ("mean_vec", C.c_void_p), ("var_vec", C.c_void_p),
                ("acc_means", C.c_void_p), ("acc_vars", C.c_void_p), ("num_acc_frames", C.c_double),
                ("acc_cleaned", _i)]


_sig("orc_bn_propagate", None, C.POINTER(BnState), f32p, _i, f32p, _i, _i, f32p)
_sig("orc_bn_backpropagate", None, C.POINTER(BnState), f32p, _i, f32p, _i, f32p, _i, _i, _f, f32p)
_sig("orc_bn_update", None, C.POINTER(BnState), _f)
_sig("orc_bn_feedforward", None, C.POINTER(BnState), f32p, _i, f32p, _i, _i)
_sig("orc_bn_global_stats_from_acc", None, C.POINTER(BnState))


class XentStats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("frames", "correct", "loss", "entropy", "likelyhood")]


_sig("orc_xent_eval", None, f32p, f32p, _i, f32p, _i, _i, _i, f32p, _i, C.POINTER(XentStats))
_sig("orc_mse_eval", None, f32p, f32p, _i, f32p, _i, _i, _i, f32p, _i, C.POINTER(C.c_double), C.POINTER(C.c_double))
_sig("orc_multitask_eval", None, _i, i32p, i32p, f32p, f32p, f32p, _i, f32p, _i, _i, f32p, _i, C.POINTER(XentStats), C.POINTER(C.c_double), C.POINTER(C.c_double))
_sig("orc_dnn_create", C.c_void_p, _i, _i, _i, _i, _i, _i, C.c_uint)
_sig("orc_dnn_destroy", None, C.c_void_p)
_sig("orc_dnn_train_step", C.c_double, C.c_void_p, f32p, i32p, _f, _f)
_sig("orc_dnn_num_layers", _i, C.c_void_p)
_sig("orc_dnn_weight", C.POINTER(_f), C.c_void_p, _i, C.POINTER(_i), C.POINTER(_i))
_sig("orc_dnn_bias", C.POINTER(_f), C.c_void_p, _i)
_sig("orc_dnn_bn_scale", C.POINTER(_f), C.c_void_p, _i)
_sig("orc_dnn_bn_shift", C.POINTER(_f), C.c_void_p, _i)
_sig("orc_dnn_output", C.POINTER(_f), C.c_void_p)
_sig("orc_golden_uniform_fill", None, C.POINTER(C.c_ulonglong), f32p, C.c_long, _f, _f)


class GoldenRng:
    """the golden generators' Uniform() / Fill() (oracle/gen_cumatrix_blas_golden.cpp), seeded with a recorded state"""

    def __init__(self, state):
        self.state = C.c_ulonglong(int(state))

    def fill(self, shape, lo, hi):
        out = np.empty(shape, np.float32)
        lib.orc_golden_uniform_fill(C.byref(self.state), out.reshape(-1), out.size, lo, hi)
        return out


# front-end components of the CNN / cFSMN recipes (oracle/aslp_oracle_conv.c)
_sig("orc_linear_propagate", None, f32p, _i, f32p, _i, _i, f32p, _i, _i, _i)
_sig("orc_linear_backpropagate", None, f32p, _i, f32p, _i, _i, f32p, _i, _i, _i)
_sig("orc_linear_update", None, f32p, _i, f32p, _i, f32p, _i, f32p, _i, _i, _i, _i, C.POINTER(AffineOpts))
_sig("orc_conv_propagate", None, f32p, _i, f32p, f32p, _i, _i, _i, f32p, _i, f32p, _i, _i, _i, _i)
_sig("orc_conv_backpropagate", None, f32p, _i, f32p, f32p, _i, _i, _i, f32p, _i, _i, _i, _i, _i)
_sig("orc_conv_update", None, f32p, _i, f32p, f32p, f32p, f32p, f32p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f)
_sig("orc_max_pool_propagate", None, f32p, _i, f32p, _i, _i, _i, _i, _i, _i)
_sig("orc_max_pool_backpropagate", None, f32p, _i, f32p, _i, f32p, _i, f32p, _i, _i, _i, _i, _i, _i)
_sig("orc_length_norm_propagate", None, f32p, _i, f32p, f32p, _i, _i, _i)
_sig("orc_length_norm_backpropagate", None, f32p, _i, f32p, _i, f32p, _i, _i)
_sig("orc_group_pnorm", None, f32p, _i, f32p, _i, _i, _i, _i, _f)
_sig("orc_group_pnorm_deriv", None, f32p, _i, f32p, _i, f32p, _i, _i, _i, _i, _f)
_sig("orc_group_max", None, f32p, _i, f32p, _i, _i, _i, _i)
_sig("orc_group_max_deriv", None, f32p, _i, f32p, _i, f32p, _i, _i, _i, _i)
_sig("orc_mul_rows_group_mat", None, f32p, _i, f32p, _i, _i, _i, _i)


def c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- numpy-level helpers (always contiguous, stride = cols) -------------------------------
def add_mat_mat(Cm, alpha, A, transA, B, transB, beta):
    Cm = c32(Cm).copy(); A = c32(A); B = c32(B)
    M, N = Cm.shape
    K = A.shape[0] if transA else A.shape[1]
    lib.orc_add_mat_mat(Cm, M, N, N, alpha, A, A.shape[1], int(transA), B, B.shape[1], int(transB), K, beta)
    return Cm


def unary(name, x):
    x = c32(x); y = np.empty_like(x)
    getattr(lib, name)(y, x.shape[1], x, x.shape[1], x.shape[0], x.shape[1])
    return y


def binary(name, a, b):
    """orc_diff_*(eout, y|in, e|out_diff)"""
    a = c32(a); b = c32(b); o = np.empty_like(a)
    getattr(lib, name)(o, a.shape[1], a, a.shape[1], b, b.shape[1], a.shape[0], a.shape[1])
    return o


def splice(x, offsets):
    x = c32(x); off = np.ascontiguousarray(offsets, dtype=np.int32)
    y = np.empty((x.shape[0], x.shape[1] * len(off)), np.float32)
    lib.orc_splice(y, y.shape[1], x, x.shape[1], x.shape[0], x.shape[1], off, len(off))
    return y


def splice_backprop(out_diff, in_cols, offsets):
    od = c32(out_diff); off = np.ascontiguousarray(offsets, dtype=np.int32)
    idf = np.empty((od.shape[0], in_cols), np.float32)
    lib.orc_splice_backpropagate(idf, in_cols, od, od.shape[1], od.shape[0], in_cols, off, len(off))
    return idf


def find_row_max_id(m):
    m = c32(m); out = np.empty(m.shape[0], np.int32)
    lib.orc_find_row_max_id(m, m.shape[1], m.shape[0], m.shape[1], out)
    return out


class Bn:
    """Holds the state arrays of one BatchNormalization component for the oracle."""

    def __init__(self, dim):
        self.dim = dim
        self.scale = np.ones(dim, np.float32); self.shift = np.zeros(dim, np.float32)
        self.dscale = np.zeros(dim, np.float32); self.dshift = np.zeros(dim, np.float32)
        self.mean = np.zeros(dim, np.float32); self.var = np.zeros(dim, np.float32)
        self.acc_means = np.zeros(dim, np.float64); self.acc_vars = np.zeros(dim, np.float64)
        self.st = BnState(dim, self.scale.ctypes.data, self.shift.ctypes.data, self.dscale.ctypes.data,
                          self.dshift.ctypes.data, self.mean.ctypes.data, self.var.ctypes.data,
                          self.acc_means.ctypes.data, self.acc_vars.ctypes.data, 0.0, 0)
        self.xs = None

    def propagate(self, x):
        x = c32(x); out = np.zeros_like(x)  # Component::Propagate zeroes `out` first (nnet-component.h:311)
        self.xs = np.empty_like(x)
        lib.orc_bn_propagate(C.byref(self.st), out, x.shape[1], x, x.shape[1], x.shape[0], self.xs)
        return out

    def backpropagate(self, x, dy, momentum):
        x = c32(x); dy = c32(dy); idf = np.empty_like(x)
        lib.orc_bn_backpropagate(C.byref(self.st), idf, x.shape[1], x, x.shape[1], dy, dy.shape[1], x.shape[0], momentum, self.xs)
        return idf

    def update(self, lr):
        lib.orc_bn_update(C.byref(self.st), lr)


def xent_eval(fw, net_out, tgt):
    fw = c32(fw); y = c32(net_out); t = c32(tgt); diff = np.empty_like(y); st = XentStats()
    lib.orc_xent_eval(fw, y, y.shape[1], t, t.shape[1], y.shape[0], y.shape[1], diff, y.shape[1], C.byref(st))
    return diff, dict(frames=st.frames, correct=st.correct, loss=st.loss, entropy=st.entropy, likelyhood=st.likelyhood)


def multitask_eval(spec, fw, net_out, tgt):
    """MultiTaskLoss::Eval (nnet-loss.cc:341-368) for spec = [(kind, dim, weight), ...], kind 'xent' | 'mse'.  Returns the diff and, per
    task, the call's increments: xent -> dict(frames, correct, loss, entropy, likelyhood); mse -> dict(loss, frames)."""
    n = len(spec)
    kinds = np.array([0 if k == "xent" else 1 for k, _, _ in spec], np.int32)
    dims = np.array([d for _, d, _ in spec], np.int32)
    wts = np.array([w for _, _, w in spec], np.float32)
    fw = c32(fw); y = c32(net_out); t = c32(tgt); diff = np.zeros_like(y)
    xs = (XentStats * n)()
    ml = (C.c_double * n)(); mf = (C.c_double * n)()
    lib.orc_multitask_eval(n, kinds, dims, wts, fw, y, y.shape[1], t, t.shape[1], y.shape[0], diff, y.shape[1], xs, ml, mf)
    out = []
    for i, (k, _, _) in enumerate(spec):
        if k == "xent":
            out.append(dict(frames=xs[i].frames, correct=xs[i].correct, loss=xs[i].loss, entropy=xs[i].entropy, likelyhood=xs[i].likelyhood))
        else:
            out.append(dict(loss=ml[i], frames=mf[i]))
    return diff, out


def rel_err(a, b):
    """relative Frobenius error, the reference's AssertEqual metric (cu-matrix.h:803-811)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))


def max_err(a, b):
    """largest element-wise deviation, relative to max(1, largest |reference|): a whole-tensor norm lets a few wrong elements through"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b))))) if a.size else 0.0


def applied_gradient_errors(before, after, lr, grad_ref):
    """What the engine APPLIED in one step, read back from the parameters -- g = (W_before - W_after) / lr -- against the gradient the
    oracle applied (its momentum-carrying, clipped `*_corr` tensor).  Comparing updated parameters instead hides a wrong gradient:
    with |lr g| << |W| a 10 % error in g moves rel_err(W) by 1e-5.  Returns (relative Frobenius error, max-element error, and the
    read-back floor of both): W_after is an fp32 number, so every element of g carries up to 2^-24 |W| / lr of rounding."""
    before = np.asarray(before, np.float64).ravel(); after = np.asarray(after, np.float64).ravel()
    ref = np.asarray(grad_ref, np.float64).ravel()
    g = (before - after) / lr
    nref = np.linalg.norm(ref)
    floor_el = 2.0 ** -24 * np.maximum(np.abs(before), np.abs(after)) / abs(lr)
    rel = np.linalg.norm(g - ref) / nref if nref > 0 else np.linalg.norm(g - ref)
    rel_floor = np.linalg.norm(floor_el) / nref if nref > 0 else 0.0
    den = max(1.0, float(np.max(np.abs(ref)))) if ref.size else 1.0
    return float(rel), float(np.max(np.abs(g - ref)) / den) if ref.size else 0.0, float(rel_floor), float(np.max(floor_el) / den) if ref.size else 0.0


def assert_applied_gradients(before, after, lr, tensors_ref, tol=1e-4, what=""):
    """tensors_ref: list of (name, reference gradient) in GetParams order; every tensor on its own (W_x, W_r, bias, each peephole, W_rm ...)"""
    off = 0
    for name, ref in tensors_ref:
        n = int(np.asarray(ref).size)
        rel, mx, rel_floor, mx_floor = applied_gradient_errors(before[off:off + n], after[off:off + n], lr, ref)
        assert rel <= tol + 2.0 * rel_floor, (what, name, "rel", rel, "read-back floor", rel_floor)
        assert mx <= 10.0 * tol + 2.0 * mx_floor, (what, name, "max element", mx, "read-back floor", mx_floor)
        off += n
    assert off == len(before), (off, len(before))


# ---- recurrent oracle (oracle/aslp_oracle_rnn.c) -------------------------------------------------
class LstmDirC(C.Structure):
    _fields_ = [("D", _i), ("C", _i), ("R", _i), ("cifg", _i)] + [(n, C.c_void_p) for n in
                ("w_x", "w_r", "bias", "peep_i", "peep_f", "peep_o", "w_rm")]


class GruC(C.Structure):
    _fields_ = [("D", _i), ("H", _i)] + [(n, C.c_void_p) for n in ("w_zrm_x", "w_zr_h", "w_m_g", "bias")]


_LP, _GP = C.POINTER(LstmDirC), C.POINTER(GruC)
_sig("orc_lstm_width", _i, _LP)
_sig("orc_lstm_forward", None, _LP, f32p, _i, _i, _i, _i, C.c_void_p, C.c_void_p, f32p)
_sig("orc_lstm_backward", None, _LP, f32p, _i, _i, _i, _i, f32p, f32p, f32p, _i, _f)
_sig("orc_lstm_grads", None, _LP, _LP, f32p, _i, _i, _i, _i, f32p, f32p, _f, _f)
_sig("orc_lstm_update", None, _LP, _LP, _f)
_sig("orc_gru_forward", None, _GP, f32p, _i, _i, _i, C.c_void_p, f32p)
_sig("orc_gru_backward", None, _GP, f32p, _i, _i, _i, f32p, f32p, f32p, _i)
_sig("orc_gru_grads", None, _GP, _GP, f32p, _i, _i, _i, f32p, f32p, _f, _f)
_sig("orc_gru_update", None, _GP, _GP, _f)


class LstmDir:
    """One direction of an LSTM-family component as numpy arrays + the C view of them."""
    NAMES = ("w_x", "w_r", "bias", "peep_i", "peep_f", "peep_o", "w_rm")

    def __init__(self, D, Cc, R, cifg, rng=None, scale=0.1, zero=False):
        self.D, self.C, self.R, self.cifg = D, Cc, R, int(cifg)
        G = 3 if cifg else 4
        rec = R if R > 0 else Cc
        shapes = dict(w_x=(G * Cc, D), w_r=(G * Cc, rec), bias=(G * Cc,), peep_i=(Cc,), peep_f=(Cc,), peep_o=(Cc,),
                      w_rm=(max(R, 1), Cc))
        for n in self.NAMES:
            a = np.zeros(shapes[n], np.float32) if zero else ((rng.random(shapes[n]) - 0.5) * 2 * scale).astype(np.float32)
            setattr(self, n, a)
        self.c = LstmDirC(D, Cc, R, int(cifg), *[getattr(self, n).ctypes.data for n in self.NAMES])
        if R == 0:
            self.c.w_rm = None
        self.width = lib.orc_lstm_width(C.byref(self.c))
        self.rec = rec
        self.off_rec = (G + 3) * Cc if R > 0 else (G + 2) * Cc

    def named_tensors(self, prefix=""):
        """(name, tensor) in file / GetParams order"""
        names = ["w_x", "w_r", "bias"] + ([] if self.cifg else ["peep_i"]) + ["peep_f", "peep_o"] + (["w_rm"] if self.R > 0 else [])
        return [(prefix + n, getattr(self, n)) for n in names]

    def tensors(self):
        """file / GetParams order"""
        out = [self.w_x, self.w_r, self.bias]
        if not self.cifg:
            out.append(self.peep_i)
        out += [self.peep_f, self.peep_o]
        if self.R > 0:
            out.append(self.w_rm)
        return out

    def flat(self):
        return np.concatenate([t.ravel() for t in self.tensors()])

    def forward(self, x, T, S, reverse=False, init_state=None, seq_len=None):
        x = c32(x)
        buf = np.zeros(((T + 2) * S, self.width), np.float32)
        ist = c32(init_state) if init_state is not None else None
        sl = np.ascontiguousarray(seq_len, np.int32) if seq_len is not None else None
        lib.orc_lstm_forward(C.byref(self.c), x, x.shape[1], T, S, int(reverse), ist.ctypes.data if ist is not None else None,
                             sl.ctypes.data if sl is not None else None, buf)
        return buf

    def out_of(self, buf, T, S):
        return buf[S:(T + 1) * S, self.off_rec:self.off_rec + self.rec].copy()

    def backward(self, out_diff, T, S, buf, reverse=False, in_diff=None, beta=0.0):
        od = c32(out_diff)
        dbuf = np.zeros_like(buf)
        idf = np.zeros((T * S, self.D), np.float32) if in_diff is None else in_diff
        lib.orc_lstm_backward(C.byref(self.c), od, od.shape[1], T, S, int(reverse), buf, dbuf, idf, self.D, beta)
        return dbuf, idf

    def grads(self, g, x, T, S, buf, dbuf, mmt, clip, reverse=False):
        x = c32(x)
        lib.orc_lstm_grads(C.byref(self.c), C.byref(g.c), x, x.shape[1], T, S, int(reverse), buf, dbuf, mmt, clip)

    def update(self, g, lr):
        lib.orc_lstm_update(C.byref(self.c), C.byref(g.c), lr)


class Gru:
    NAMES = ("w_zrm_x", "w_zr_h", "w_m_g", "bias")

    def __init__(self, D, H, rng=None, scale=0.1, zero=False):
        self.D, self.H = D, H
        shapes = dict(w_zrm_x=(3 * H, D), w_zr_h=(2 * H, H), w_m_g=(H, H), bias=(3 * H,))
        for n in self.NAMES:
            a = np.zeros(shapes[n], np.float32) if zero else ((rng.random(shapes[n]) - 0.5) * 2 * scale).astype(np.float32)
            setattr(self, n, a)
        self.c = GruC(D, H, *[getattr(self, n).ctypes.data for n in self.NAMES])

    def tensors(self):
        return [self.w_zrm_x, self.w_zr_h, self.w_m_g, self.bias]

    def flat(self):
        return np.concatenate([t.ravel() for t in self.tensors()])

    def forward(self, x, T, S, init_state=None):
        x = c32(x)
        buf = np.zeros(((T + 2) * S, 5 * self.H), np.float32)
        ist = c32(init_state) if init_state is not None else None
        lib.orc_gru_forward(C.byref(self.c), x, x.shape[1], T, S, ist.ctypes.data if ist is not None else None, buf)
        return buf

    def out_of(self, buf, T, S):
        return buf[S:(T + 1) * S, 4 * self.H:].copy()

    def backward(self, out_diff, T, S, buf):
        od = c32(out_diff)
        dbuf = np.zeros_like(buf)
        idf = np.zeros((T * S, self.D), np.float32)
        lib.orc_gru_backward(C.byref(self.c), od, od.shape[1], T, S, buf, dbuf, idf, self.D)
        return dbuf, idf

    def grads(self, g, x, T, S, buf, dbuf, mmt, clip):
        x = c32(x)
        lib.orc_gru_grads(C.byref(self.c), C.byref(g.c), x, x.shape[1], T, S, buf, dbuf, mmt, clip)

    def update(self, g, lr):
        lib.orc_gru_update(C.byref(self.c), C.byref(g.c), lr)


# ---- depthwise temporal oracle (oracle/aslp_oracle_temporal.c) ------------------------------------
_sig("orc_rowconv_propagate", None, f32p, _i, _i, f32p, _i, _i, _i, i32p, f32p, f32p, _i)
_sig("orc_rowconv_backpropagate", None, f32p, _i, _i, f32p, _i, _i, _i, i32p, f32p, f32p, f32p, f32p, _i)
_sig("orc_rowconv_update", None, f32p, f32p, f32p, _i, _i, _f, _f)
_sig("orc_fsmn_propagate", None, f32p, _i, _i, _i, f32p, _i, _i, f32p, _i)
_sig("orc_fsmn_backpropagate", None, f32p, _i, _i, _i, f32p, _i, f32p, _i, _i, _f, f32p, f32p, _i)
_sig("orc_fsmn_update", None, f32p, f32p, _i, _i, _i, _f)


class RowConv:
    def __init__(self, D, K, rng):
        self.D, self.K = D, K
        self.w = rng.standard_normal((D, K + 1)).astype(np.float32)
        self.w_diff = np.zeros_like(self.w); self.w_corr = np.zeros_like(self.w)

    def propagate(self, x, T, S, lens):
        x = c32(x); lens = np.ascontiguousarray(lens, np.int32)
        self.in_buf = np.zeros((S * (T + self.K), self.D), np.float32)
        out = np.zeros((T * S, self.D), np.float32)   # Component::Propagate zeroes out first
        lib.orc_rowconv_propagate(self.w, self.D, self.K, x, x.shape[1], T, S, lens, self.in_buf, out, self.D)
        return out

    def backpropagate(self, od, T, S, lens):
        od = c32(od); lens = np.ascontiguousarray(lens, np.int32)
        idb = np.zeros_like(self.in_buf)
        in_diff = np.zeros((T * S, self.D), np.float32)  # Component::Backpropagate zeroes in_diff first
        lib.orc_rowconv_backpropagate(self.w, self.D, self.K, od, od.shape[1], T, S, lens, self.in_buf, idb, self.w_diff, in_diff, self.D)
        return in_diff

    def update(self, lr, mmt):
        lib.orc_rowconv_update(self.w, self.w_corr, self.w_diff, self.D, self.K, lr, mmt)


class Fsmn:
    def __init__(self, D, P, F, rng, scale=0.3):
        self.D, self.P, self.F = D, P, F
        self.coef = ((rng.random((P + F + 1, D)) - 0.5) * 2 * scale).astype(np.float32)
        self.corr = np.zeros_like(self.coef)

    def propagate(self, x):
        x = c32(x); out = np.zeros_like(x)
        lib.orc_fsmn_propagate(self.coef, self.D, self.P, self.F, x, x.shape[1], x.shape[0], out, self.D)
        return out

    def backpropagate(self, x, od, clip):
        x = c32(x); od = c32(od); in_diff = np.zeros_like(x)
        lib.orc_fsmn_backpropagate(self.coef, self.D, self.P, self.F, x, x.shape[1], od, od.shape[1], x.shape[0], clip, self.corr, in_diff, self.D)
        return in_diff

    def update(self, lr):
        lib.orc_fsmn_update(self.coef, self.corr, self.D, self.P, self.F, lr)


# ---- front-end components (oracle/aslp_oracle_conv.c) ------------------------------------------------------------------
class Linear:
    """LinearTransform (nnet-linear-transform.h): W [out x in] + momentum buffer"""
    def __init__(self, W):
        self.W = c32(W).copy(); self.corr = np.zeros_like(self.W)

    def propagate(self, x):
        x = c32(x); out = np.empty((x.shape[0], self.W.shape[0]), np.float32)
        lib.orc_linear_propagate(out, out.shape[1], x, x.shape[1], x.shape[0], self.W, self.W.shape[1], self.W.shape[1], self.W.shape[0])
        return out

    def backpropagate(self, od):
        od = c32(od); idf = np.empty((od.shape[0], self.W.shape[1]), np.float32)
        lib.orc_linear_backpropagate(idf, idf.shape[1], od, od.shape[1], od.shape[0], self.W, self.W.shape[1], self.W.shape[1], self.W.shape[0])
        return idf

    def update(self, x, od, lr, mmt=0.0, l2=0.0, l1=0.0, coef=1.0):
        x, od = c32(x), c32(od)
        o = AffineOpts(lr, mmt, l2, l1, coef, 1.0, 0.0)
        lib.orc_linear_update(self.W, self.W.shape[1], self.corr, self.corr.shape[1], x, x.shape[1], od, od.shape[1], x.shape[0], self.W.shape[1],
                              self.W.shape[0], C.byref(o))


class Affine:
    """AffineTransform (nnet-affine-transform.h:186-245): W [out x in], bias, their momentum buffers"""
    def __init__(self, W, b):
        self.W = c32(W).copy(); self.b = c32(b).copy()
        self.Wc = np.zeros_like(self.W); self.bc = np.zeros_like(self.b)

    def propagate(self, x):
        x = c32(x); out = np.empty((x.shape[0], self.W.shape[0]), np.float32)
        lib.orc_affine_propagate(out, out.shape[1], x, x.shape[1], x.shape[0], self.W, self.W.shape[1], self.b, self.W.shape[1], self.W.shape[0])
        return out

    def backpropagate(self, od):
        od = c32(od); idf = np.empty((od.shape[0], self.W.shape[1]), np.float32)
        lib.orc_affine_backpropagate(idf, idf.shape[1], od, od.shape[1], od.shape[0], self.W, self.W.shape[1], self.W.shape[1], self.W.shape[0])
        return idf

    def update(self, x, od, lr, mmt=0.0, l2=0.0, l1=0.0):
        x, od = c32(x), c32(od)
        o = AffineOpts(lr, mmt, l2, l1, 1.0, 1.0, 0.0)
        lib.orc_affine_update(self.W, self.W.shape[1], self.b, self.Wc, self.Wc.shape[1], self.bc, x, x.shape[1], od, od.shape[1], x.shape[0],
                              self.W.shape[1], self.W.shape[0], C.byref(o))


class Conv:
    """ConvolutionalComponent (nnet-convolutional-component.h) in the reference's per-patch structure"""
    def __init__(self, filters, bias, in_dim, patch_dim, patch_step, patch_stride):
        self.filters = c32(filters).copy(); self.bias = c32(bias).copy()
        self.in_dim, self.pd, self.ps, self.pst = in_dim, patch_dim, patch_step, patch_stride
        self.F, self.K = self.filters.shape
        self.P = 1 + (patch_stride - patch_dim) // patch_step
        assert self.K == (in_dim // patch_stride) * patch_dim
        self.fgrad = np.zeros_like(self.filters); self.bgrad = np.zeros_like(self.bias)

    def propagate(self, x):
        x = c32(x); N = x.shape[0]
        out = np.empty((N, self.F * self.P), np.float32)
        self.patches = np.empty((N, self.K * self.P), np.float32)
        lib.orc_conv_propagate(out, out.shape[1], self.patches, x, x.shape[1], N, self.in_dim, self.filters, self.K, self.bias, self.F, self.pd, self.ps,
                               self.pst)
        return out

    def backpropagate(self, od):
        od = c32(od); N = od.shape[0]
        idf = np.empty((N, self.in_dim), np.float32)
        pd = np.empty((N, self.K * self.P), np.float32)
        lib.orc_conv_backpropagate(idf, self.in_dim, pd, od, od.shape[1], N, self.in_dim, self.filters, self.K, self.F, self.pd, self.ps, self.pst)
        return idf

    def update(self, od, lr, coef=1.0, bias_coef=1.0, max_norm=0.0):
        od = c32(od)
        lib.orc_conv_update(self.filters, self.K, self.bias, self.fgrad, self.bgrad, self.patches, od, od.shape[1], od.shape[0], self.in_dim, self.F,
                            self.pd, self.ps, self.pst, lr, coef, bias_coef, max_norm)


def max_pool(x, size, step, stride):
    x = c32(x); n_p = x.shape[1] // stride
    out = np.empty((x.shape[0], (1 + (n_p - size) // step) * stride), np.float32)
    lib.orc_max_pool_propagate(out, out.shape[1], x, x.shape[1], x.shape[0], x.shape[1], size, step, stride)
    return out


def max_pool_backprop(x, out, od, size, step, stride):
    x, out, od = c32(x), c32(out), c32(od)
    idf = np.empty_like(x)
    lib.orc_max_pool_backpropagate(idf, idf.shape[1], x, x.shape[1], out, out.shape[1], od, od.shape[1], x.shape[0], x.shape[1], size, step, stride)
    return idf


def length_norm(x):
    x = c32(x); out = np.empty_like(x); sc = np.empty(x.shape[0], np.float32)
    lib.orc_length_norm_propagate(out, out.shape[1], sc, x, x.shape[1], x.shape[0], x.shape[1])
    return out, sc


def length_norm_backprop(od, sc):
    od = c32(od); idf = np.empty_like(od)
    lib.orc_length_norm_backpropagate(idf, idf.shape[1], od, od.shape[1], c32(sc), od.shape[0], od.shape[1])
    return idf


def group_pnorm(x, out_cols, power):
    x = c32(x); y = np.empty((x.shape[0], out_cols), np.float32)
    lib.orc_group_pnorm(y, out_cols, x, x.shape[1], x.shape[0], out_cols, x.shape[1] // out_cols, power)
    return y


def group_pnorm_deriv(x, y, power):
    x, y = c32(x), c32(y); d = np.empty_like(x)
    lib.orc_group_pnorm_deriv(d, d.shape[1], x, x.shape[1], y, y.shape[1], x.shape[0], x.shape[1], x.shape[1] // y.shape[1], power)
    return d


def group_max(x, out_cols):
    x = c32(x); y = np.empty((x.shape[0], out_cols), np.float32)
    lib.orc_group_max(y, out_cols, x, x.shape[1], x.shape[0], out_cols, x.shape[1] // out_cols)
    return y


def group_max_deriv(x, y):
    x, y = c32(x), c32(y); d = np.empty_like(x)
    lib.orc_group_max_deriv(d, d.shape[1], x, x.shape[1], y, y.shape[1], x.shape[0], x.shape[1], x.shape[1] // y.shape[1])
    return d


def mul_rows_group_mat(y, src):
    y = c32(y).copy(); src = c32(src)
    lib.orc_mul_rows_group_mat(y, y.shape[1], src, src.shape[1], y.shape[0], y.shape[1], y.shape[1] // src.shape[1])
    return y
