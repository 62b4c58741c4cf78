"""ctypes binding of libaslp_hip.so (the C ABI declared in include/*.h).

The library is the product: if it is missing this module raises -- there is no CPU or
PyTorch fallback for any op (oracle/ is test infrastructure and is never imported here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaslp_hip.so")


class MatrixDim(C.Structure):
    """cu-matrixdim.h:52-56"""
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("stride", C.c_int32)]


class Dim3(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("z", C.c_uint32)]


class PlanesOut(C.Structure):
    """aslp_planes_out (include/aslp_kernels.h)"""
    _fields_ = [("hi", C.c_void_p), ("lo", C.c_void_p), ("ld", C.c_int), ("slot", C.c_void_p), ("parts", C.c_void_p), ("nparts", C.c_int), ("planes_written", C.c_int)]


class GemmEpilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("clip", C.c_float), ("W", C.c_void_p), ("ldw", C.c_int),
                ("w_alpha", C.c_float), ("act_out", C.c_void_p), ("ld_act", C.c_int), ("act", C.c_int),
                ("colsum", C.c_void_p), ("colsum_beta", C.c_float), ("colsum_w", C.c_void_p), ("colsum_w_alpha", C.c_float),
                ("colstats", C.c_void_p), ("colstats_ld", C.c_int), ("c_src", C.c_void_p), ("ld_c_src", C.c_int),
                ("planes", PlanesOut), ("planes_of", C.c_int), ("wmax_parts", C.c_void_p), ("cmax_parts", C.c_void_p),
                ("bound_w_parts", C.c_void_p), ("bound_c_parts", C.c_void_p), ("bound_n", C.c_int)]


class GruSeq(C.Structure):
    """aslp_gru_seq (include/aslp_kernels.h)"""
    _fields_ = [("y", C.c_void_p), ("d", C.c_void_p), ("w_zr", C.c_void_p), ("w_m", C.c_void_p), ("ldw_zr", C.c_int), ("ldw_m", C.c_int),
                ("ld", C.c_int), ("T", C.c_int), ("S", C.c_int), ("H", C.c_int), ("s_begin", C.c_int), ("s_count", C.c_int)]


class RnnVecGrad(C.Structure):
    """aslp_rnn_vec_grad (include/aslp_kernels.h)"""
    _fields_ = [("d", C.c_void_p), ("x", C.c_void_p), ("ldx", C.c_int), ("n", C.c_int), ("corr", C.c_void_p), ("param", C.c_void_p)]


class SodSolver(C.Structure):
    """aslp_sod_solver (include/aslp_kernels.h)"""
    _fields_ = [("solver", C.c_int), ("lr", C.c_float), ("momentum", C.c_float), ("gamma", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("corr1", C.c_float), ("corr2", C.c_float)]


class CtcComputeInfo(C.Structure):
    """warp-ctc/include/ctc.h:45-60 (ctcComputeInfo: loc + union{num_threads, stream})"""
    _fields_ = [("loc", C.c_int), ("stream_or_threads", C.c_void_p)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libaslp_hip.so not found at %s: build it with `make -C kaldi-aslp_amd` "
            "(or __graft_entry__.build()). There is no fallback path." % LIB_PATH)
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_vp, _f, _i, _d3, _md = C.c_void_p, C.c_float, C.c_int, Dim3, MatrixDim


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


# runtime
_sig("aslp_set_stream", None, _vp)
_sig("aslp_get_stream", _vp)
_sig("aslp_get_last_error", _i, C.c_char_p, _i)
_sig("aslp_device_sync", _i)
_sig("aslp_version", C.c_char_p)

# elementwise (B1)
for _n in ("cudaF_set_const", "cudaF_add", "cudaF_scale", "cudaF_apply_pow", "cudaF_apply_floor", "cudaF_apply_ceiling"):
    _sig(_n, None, _d3, _d3, _vp, _f, _md)
for _n in ("cudaF_apply_log", "cudaF_apply_exp", "cudaF_apply_heaviside", "cudaF_invert_elements"):
    _sig(_n, None, _d3, _d3, _vp, _md)
_sig("cudaF_mul_elements", None, _d3, _d3, _vp, _vp, _md, _i)
_sig("cudaF_mul_cols_vec", None, _d3, _d3, _vp, _vp, _md)
_sig("cudaF_mul_rows_vec", None, _d3, _d3, _vp, _vp, _md)
_sig("cudaF_add_mat", None, _d3, _d3, _f, _vp, _vp, _md, _i, _i)
_sig("cudaF_add_vec_to_cols", None, _d3, _d3, _f, _vp, _f, _vp, _md)
_sig("cudaF_add_vec_to_rows", None, _d3, _d3, _f, _vp, _f, _vp, _md)
_sig("cudaF_add_mat_diag_vec", None, _d3, _d3, _f, _vp, _md, _vp, _i, _i, _vp, _f)
_sig("cudaF_add_mat_mat_elements", None, _d3, _d3, _vp, _vp, _vp, _md, _i, _i, _f, _f)
_sig("cudaF_add_row_sum_mat", None, _d3, _d3, _vp, _vp, _md, _i, _i, _f, _f)
_sig("cudaF_add_conv_mat_mat_elements", None, _d3, _d3, _vp, _vp, _vp, _md, _i, _i, _f, _f)
_sig("cudaF_sigmoid", None, _d3, _d3, _vp, _vp, _md, _i)
_sig("cudaF_tanh", None, _d3, _d3, _vp, _vp, _md, _i)
_sig("cudaF_diff_sigmoid", None, _d3, _d3, _vp, _vp, _vp, _md, _i, _i)
_sig("cudaF_diff_tanh", None, _d3, _d3, _vp, _vp, _vp, _md, _i, _i)
_sig("cudaF_regularize_l1", None, _d3, _d3, _vp, _vp, _f, _f, _md, _i)
# gathers
_sig("cudaF_copy_cols", None, _d3, _d3, _vp, _vp, _vp, _md, _i)
_sig("cudaF_add_cols", None, _d3, _d3, _vp, _vp, _vp, _md, _i)
_sig("cudaF_copy_rows", None, _d3, _d3, _vp, _vp, _vp, _md, _i)
_sig("cudaF_add_rows", None, _d3, _d3, _f, _vp, _vp, _vp, _md, _i)
_sig("cudaF_randomize", None, _d3, _d3, _vp, _vp, _vp, _md, _md)
_sig("cudaF_splice", None, _d3, _d3, _vp, _vp, _vp, _md, _md)
_sig("cudaF_copy", None, _d3, _d3, _vp, _vp, _vp, _md, _md)
_sig("cudaI32_set_const", None, _d3, _d3, _vp, C.c_int32, _md)
# reductions
_sig("cudaF_softmax_reduce", None, C.c_size_t, C.c_size_t, _vp, _vp, _md, _i)
_sig("cudaF_log_softmax_reduce", None, C.c_size_t, C.c_size_t, _vp, _vp, _md, _i)
_sig("cudaF_find_row_max_id", None, _d3, _d3, _vp, _vp, _vp, C.c_int32, _md)
_sig("cudaF_diff_xent", None, _d3, _d3, _vp, _vp, _vp, _md)
_sig("cudaF_add_diag_mat_mat", None, _i, _i, _f, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _f)
_sig("cudaF_add_vec_vec", None, _i, _i, _f, _vp, _vp, _vp, _f, _i)
_sig("cudaF_vec_sum", None, _i, _i, _vp, _vp, _i, _i)
_sig("aslp_add_row_sum_mat_vec", None, _f, _vp, _md, _f, _vp)
_sig("aslp_add_col_sum_mat_vec", None, _f, _vp, _md, _f, _vp)
_sig("aslp_add_row_sum_mat_vec_sgd", None, _f, _vp, _md, _f, _vp, _vp, _f)
_sig("aslp_vec_axpy2", None, _f, _vp, _vp, _vp, _vp, _i)
_sig("aslp_find_row_max_id", None, _vp, _md, _vp)
_sig("aslp_matrix_sum", None, _vp, _md, _vp)
_sig("aslp_copy_mat", None, _vp, _md, _vp, _i)
_sig("aslp_copy_mat_trans", None, _vp, _md, _vp, _i)
_sig("aslp_vec_axpy", None, _f, _vp, _vp, _i)
_sig("aslp_f2d", None, _vp, _vp, _i)
_sig("aslp_d2f", None, _vp, _vp, _i)
# gemm (B2)
_sig("aslp_sgemm", _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _f, _vp, _i)
_sig("aslp_sgemm_ex", _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _i, _f, _vp, _i, C.POINTER(GemmEpilogue))
_sig("aslp_sgemm_pair_ex", _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _i, _f, _vp, _vp, _i, C.POINTER(GemmEpilogue), C.POINTER(GemmEpilogue))
_sig("aslp_gemm_split16", None, _i)
_sig("aslp_gemm_split16_tile", None, _i)
_sig("aslp_planes_new", _vp)
_sig("aslp_planes_free", None, _vp)
_sig("aslp_planes_convert", _i, _vp, _vp, _md)
_sig("aslp_planes_reserve", None, _vp, _i, _i)
_sig("aslp_copy_mat_planes", _i, _vp, _md, _vp, _i, C.POINTER(PlanesOut))
_sig("aslp_planes_as_output", None, _vp, C.POINTER(PlanesOut))
_sig("aslp_sgemm_planes_ex", _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _i, _vp, _f, _vp, _i, C.POINTER(GemmEpilogue))
_sig("aslp_gemm_last_parts", _i)
_sig("aslp_params_changed", None)
_sig("aslp_keep_weight_planes", None, _i)
_sig("aslp_absmax_parts", None, _vp, _md, _vp)
_sig("aslp_weight_bound", None, _vp, _i, _vp, _i, _vp, _vp, _i, _f, _f, _f, _f, _vp)
_sig("aslp_gemm_profile", None, _i)
_sig("aslp_gemm_profile_reset", None)
_sig("aslp_gru_seq_supported", _i, C.POINTER(GruSeq), _i)
_sig("aslp_gemm_force_tile", None, _i)
_sig("aslp_gemm_last_tile", _i)
_sig("aslp_gemm_profile_get", C.c_long, _i, C.POINTER(C.c_double), C.POINTER(C.c_double))
_sig("aslp_gemm_profile_tile", _i, _i, C.c_char_p, _i)
_sig("aslp_gemm_profile_dump", None)
_sig("aslp_lstm_seq_polls", C.c_uint, _i)
_sig("aslp_lstm_seq_timing", None, _i, C.POINTER(C.c_ulonglong))
_sig("aslp_region_profile", None, _i)
_sig("aslp_region_reset", None)
_sig("aslp_region_get", C.c_long, C.c_char_p, C.POINTER(C.c_double))
# fused
_sig("aslp_bn_forward", None, _vp, _md, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f)
_sig("aslp_bn_backward", None, _vp, _md, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i)
# depthwise temporal filters (csrc/temporal.hip)
_sig("aslp_fsmn_filter", None, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i)
_sig("aslp_fsmn_backward", None, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f)
_sig("aslp_rowconv_forward", None, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp)
_sig("aslp_rowconv_backward_fused", None, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _f, _f, _i)
_sig("aslp_bn_apply", None, _vp, _md, _vp, _i, _vp, _vp, _vp, _vp)
_sig("aslp_xent_eval", None, _vp, _md, _vp, _i, _vp, _vp, _vp, _i, _vp)
_sig("aslp_bn_forward_act", None, _vp, _md, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i)
_sig("aslp_bn_forward_stats", _i, _vp, _md, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i, _i)
_sig("aslp_bn_backward_act", None, _vp, _md, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _i)
_sig("aslp_bn_backward_step", None, _md, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _vp, _i, _vp, _vp)
_sig("aslp_bn_backward_step_p", None, _md, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _vp, _i, _vp, _vp, C.POINTER(PlanesOut))
_sig("aslp_bn_panel_supported", _i, _i, _i)
_sig("aslp_softmax_xent_supported", _i, _i)
_sig("aslp_xent_eval_p", _i, _vp, _md, _vp, _vp, _vp, _i, _vp, _i, C.POINTER(PlanesOut))
_sig("aslp_xent_eval_rows", _i, _vp, _md, _vp, _vp, _vp, _i, _vp, _i, C.POINTER(PlanesOut))
_sig("aslp_xent_sum_rowstats", None, _vp, _i, _i, _vp)
_sig("aslp_device_shared", None, _i)
_sig("aslp_coop_convert", None, _i)
_sig("aslp_softmax_xent_eval", None, _vp, _md, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i)
_sig("aslp_dropout_forward", None, _vp, _i, _vp, _md, _vp, _i, _f, C.c_ulonglong)
_sig("aslp_dropout_backward", None, _vp, _i, _vp, _md, _vp, _i, _f)
_sig("aslp_apply_clamp", None, _vp, _md, _f, _f)
_sig("aslp_vec_diff", None, _vp, _vp, _vp, _i)
_sig("aslp_sod_solve", None, C.POINTER(SodSolver), _vp, _vp, _vp, _vp, _vp, _i)
_sig("aslp_rnn_vec_grads", None, C.POINTER(RnnVecGrad), _i, _i, _i, _f, _f, _f)
_sig("aslp_scatter_add", None, _vp, _md, _vp, _vp, _vp, _i)
_sig("aslp_splice_backward", None, _vp, _md, _vp, _i, _vp, _i)
_sig("aslp_diff_relu", None, _vp, _vp, _vp, _md, _i, _i)
_sig("aslp_max_norm_rows", None, _vp, _md, _f)


# CTC (B5, include/aslp_ctc.h)
_sig("compute_ctc_loss", _i, _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _i, _i, C.POINTER(C.c_float), _vp, CtcComputeInfo)
_sig("get_workspace_size", _i, C.POINTER(C.c_int), C.POINTER(C.c_int), _i, _i, CtcComputeInfo, C.POINTER(C.c_size_t))
_sig("ctcGetStatusString", C.c_char_p, _i)
_sig("get_warpctc_version", _i)


def check_error():
    """Raise if any launch since the last call failed (the reference throws via KALDI_ERR)."""
    buf = C.create_string_buffer(1024)
    if lib.aslp_get_last_error(buf, 1024):
        raise RuntimeError("libaslp_hip: " + buf.value.decode())


D3 = Dim3(1, 1, 1)
