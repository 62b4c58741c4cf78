"""Python mirror of the host engine's Nnet / Xent (include/aslp_nnet.h), keeping the reference's
method names (src/aslp-nnet/nnet-nnet.h:38-193, nnet-loss.h:66-117).  Tensors are torch CUDA
tensors; every method is a thin call into libaslp_hip.so -- no arithmetic happens in Python."""
import ctypes as C

import torch

from ._lib import lib, check_error
from .ops import ptr, dim, _chk

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_H = C.c_void_p


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)


_sig("aslp_nnet_last_error", C.c_char_p)
_sig("aslp_set_verbose", None, _i)
_sig("aslp_nnet_init_from_proto", _i, C.c_char_p, C.c_uint, C.POINTER(_H))
_sig("aslp_nnet_read", _i, C.c_char_p, C.POINTER(_H))
_sig("aslp_nnet_write", _i, _H, C.c_char_p, _i)
_sig("aslp_nnet_copy", _i, _H, C.POINTER(_H))
_sig("aslp_nnet_free", None, _H)
_sig("aslp_nnet_set_train_options", _i, _H, _f, _f, _f, _f)
for _n in ("aslp_nnet_input_dim", "aslp_nnet_output_dim", "aslp_nnet_num_components", "aslp_nnet_num_params"):
    _sig(_n, _i, _H)
_sig("aslp_nnet_component_marker", _i, _H, _i, C.c_char_p, _i)
_sig("aslp_nnet_info", _i, _H, C.c_char_p, _i)
_sig("aslp_nnet_set_link_aliasing", _i, _H, _i)
_sig("aslp_nnet_set_layer_fusion", _i, _H, _i)
_sig("aslp_nnet_set_update_overlap", _i, _H, _i)
_sig("aslp_nnet_propagate", _i, _H, _vp, _i, _i, _i, _vp, _i)
_sig("aslp_nnet_feedforward", _i, _H, _vp, _i, _i, _i, _vp, _i)
_sig("aslp_nnet_backpropagate", _i, _H, _vp, _i, _i, _i, _vp, _i)
_sig("aslp_nnet_reset_lstm_streams", _i, _H, C.POINTER(C.c_int32), _i)
_sig("aslp_nnet_set_seq_lengths", _i, _H, C.POINTER(C.c_int32), _i)
_sig("aslp_nnet_set_chunk_size", _i, _H, _i)
_sig("aslp_nnet_get_params", _i, _H, C.POINTER(_f), _i)
_sig("aslp_nnet_get_gpu_params", _i, _H, C.POINTER(_vp), C.POINTER(_i), _i)
_sig("aslp_nnet_param_writers_announce", _i, _H)
_sig("aslp_nnet_get_acc_stats", _i, _H, C.POINTER(_vp), C.POINTER(_i), _i, C.POINTER(C.POINTER(C.c_double)), _i, C.POINTER(_i))
_sig("aslp_nnet_component_output", _i, _H, _i, C.POINTER(_f), _i, _i)
_sig("aslp_nnet_component_out_diff", _i, _H, _i, C.POINTER(_f), _i, _i)
_sig("aslp_xent_create", _i, C.POINTER(_H))
_sig("aslp_xent_free", None, _H)
_sig("aslp_xent_eval_batch", _i, _H, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i)
_sig("aslp_xent_report", _i, _H, C.c_char_p, _i)
_sig("aslp_xent_get_stats", _i, _H, C.POINTER(C.c_double))
_sig("aslp_nnet_train_step_xent", _i, _H, _H, _vp, _i, _i, _i, _vp, _vp)
_i32p = C.POINTER(C.c_int32)
_sig("aslp_warpctc_create", _i, C.POINTER(_H))
_sig("aslp_warpctc_free", None, _H)
_sig("aslp_warpctc_eval", _i, _H, _i32p, _i, _vp, _i, _i, _i, _i32p, _i32p, _vp, _i, C.POINTER(_f))
_sig("aslp_warpctc_error_rate", _i, _H, _i32p, _i, _vp, _i, _i, _i, _i32p, _i32p)
_sig("aslp_warpctc_report", _i, _H, C.c_char_p, _i)
_sig("aslp_warpctc_get_stats", _i, _H, C.POINTER(C.c_double))
_sig("aslp_nnet_train_step_warpctc", _i, _H, _H, _vp, _i, _i, _i, _i32p, _i, _i32p, _i32p)


def _ok(rc):
    if rc != 0:
        raise RuntimeError(lib.aslp_nnet_last_error().decode())
    check_error()


class Xent:
    def __init__(self):
        self.h = _H()
        _ok(lib.aslp_xent_create(C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.aslp_xent_free(self.h)
            self.h = None

    def Eval(self, frame_weights, net_out, diff, targets=None, labels=None):
        _chk(net_out), _chk(diff), _chk(frame_weights)
        _ok(lib.aslp_xent_eval_batch(self.h, ptr(net_out), net_out.shape[0], net_out.shape[1], dim(net_out).stride,
                                     ptr(targets), dim(targets).stride if targets is not None else 0, ptr(labels),
                                     ptr(frame_weights), ptr(diff), dim(diff).stride))

    def Report(self):
        buf = C.create_string_buffer(4096)
        _ok(lib.aslp_xent_report(self.h, buf, 4096))
        return buf.value.decode()

    def GetStats(self):
        st = (C.c_double * 5)()
        _ok(lib.aslp_xent_get_stats(self.h, st))
        return dict(zip(("frames", "correct", "loss", "entropy", "likelyhood"), list(st)))


_sig("aslp_eesenctc_create", _i, C.POINTER(_H))
_sig("aslp_eesenctc_free", None, _H)
_sig("aslp_eesenctc_eval", _i, _H, _i32p, _i, _vp, _i, _i, _i, _i32p, _i32p, _vp, _i, C.POINTER(_f))
_sig("aslp_eesenctc_error_rate", _i, _H, _i32p, _i, _vp, _i, _i, _i, _i32p, _i32p)
_sig("aslp_eesenctc_report", _i, _H, C.c_char_p, _i)
_sig("aslp_eesenctc_get_stats", _i, _H, C.POINTER(C.c_double))
_sig("aslp_randomizer_mask_generate", _i, _i, _i, _i32p)
_sig("aslp_matrix_randomizer_create", _i, _i, _i, C.POINTER(_H))
_sig("aslp_matrix_randomizer_free", None, _H)
_sig("aslp_matrix_randomizer_add_data", _i, _H, _vp, _i, _i, _i)
_sig("aslp_matrix_randomizer_randomize", _i, _H, _i32p, _i)
_sig("aslp_matrix_randomizer_next", _i, _H)
_sig("aslp_matrix_randomizer_stage_begin", _i, _H)
_sig("aslp_matrix_randomizer_stage_add", _i, _H, _vp, _i, _i)
_sig("aslp_matrix_randomizer_stage_commit", _i, _H)
_sig("aslp_matrix_randomizer_stage_state", _i, _H, C.POINTER(_i))
_sig("aslp_matrix_randomizer_state", _i, _H, C.POINTER(_i))
_sig("aslp_matrix_randomizer_value", _i, _H, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i))


def randomizer_mask(size, seed=777):
    """RandomizerMask::Generate (nnet-randomizer.cc:38-44); seed < 0 keeps the current rand() state"""
    import numpy as np
    out = np.empty(size, np.int32)
    _ok(lib.aslp_randomizer_mask_generate(int(seed), int(size), out.ctypes.data_as(_i32p)))
    return out


class MatrixRandomizer:
    """aslp-nnet/nnet-randomizer.h:67-102 on device memory"""

    def __init__(self, randomizer_size=32768, minibatch_size=256):
        self.h = _H()
        _ok(lib.aslp_matrix_randomizer_create(randomizer_size, minibatch_size, C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.aslp_matrix_randomizer_free(self.h)
            self.h = None

    def AddData(self, m):
        _chk(m)
        _ok(lib.aslp_matrix_randomizer_add_data(self.h, ptr(m), m.shape[0], m.shape[1], dim(m).stride))

    def _state(self):
        st = (_i * 3)()
        _ok(lib.aslp_matrix_randomizer_state(self.h, st))
        return st

    def IsFull(self): return bool(self._state()[0])
    def Done(self): return bool(self._state()[1])
    def NumFrames(self): return int(self._state()[2])
    def Next(self): _ok(lib.aslp_matrix_randomizer_next(self.h))

    def Randomize(self, mask):
        a = _i32arr(mask)
        _ok(lib.aslp_matrix_randomizer_randomize(self.h, a, len(mask)))

    # staged refill (include/aslp_nnet.h): host rows go up on a copy stream while the current cache is consumed
    def StageBegin(self): _ok(lib.aslp_matrix_randomizer_stage_begin(self.h))

    def StageAdd(self, host_rows):
        import numpy as np
        a = np.ascontiguousarray(host_rows, dtype=np.float32)
        _ok(lib.aslp_matrix_randomizer_stage_add(self.h, a.ctypes.data_as(_vp), a.shape[0], a.shape[1]))

    def StageCommit(self): _ok(lib.aslp_matrix_randomizer_stage_commit(self.h))

    def StageFull(self):
        st = (_i * 2)()
        _ok(lib.aslp_matrix_randomizer_stage_state(self.h, st))
        return bool(st[0])

    def Value(self):
        """copy of the current minibatch as a torch tensor (the C++ API returns a view)"""
        p, r, c, s = _vp(), _i(), _i(), _i()
        _ok(lib.aslp_matrix_randomizer_value(self.h, C.byref(p), C.byref(r), C.byref(c), C.byref(s)))
        out = torch.empty(r.value, c.value, device="cuda")
        from ._lib import MatrixDim
        lib.aslp_copy_mat(ptr(out), MatrixDim(r.value, c.value, c.value), p, s.value)
        check_error()
        return out


def _i32arr(v):
    return (C.c_int32 * max(1, len(v)))(*[int(x) for x in v])


def _flat(labels):
    flat = [int(x) for l in labels for x in l]
    return _i32arr(flat), _i32arr([len(l) for l in labels])


class WarpCtc:
    """aslp-nnet/warp-ctc.h:29: Eval / ErrorRate / Report on device-resident activations."""

    def __init__(self):
        self.h = _H()
        _ok(lib.aslp_warpctc_create(C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.aslp_warpctc_free(self.h)
            self.h = None

    def Eval(self, frame_num_utt, net_out, labels, diff=None):
        """returns (diff, costs); net_out rows are t*num_utt + s, pre-softmax activations"""
        import numpy as np
        _chk(net_out)
        if diff is None:
            diff = torch.empty_like(net_out)
        n = len(frame_num_utt)
        fl, ll = _flat(labels[:n])
        costs = np.zeros(n, np.float32)
        _ok(lib.aslp_warpctc_eval(self.h, _i32arr(frame_num_utt), n, ptr(net_out), net_out.shape[0], net_out.shape[1],
                                  dim(net_out).stride, fl, ll, ptr(diff), dim(diff).stride, costs.ctypes.data_as(C.POINTER(_f))))
        return diff, costs

    def ErrorRate(self, frame_num_utt, net_out, labels):
        _chk(net_out)
        n = len(frame_num_utt)
        fl, ll = _flat(labels[:n])
        _ok(lib.aslp_warpctc_error_rate(self.h, _i32arr(frame_num_utt), n, ptr(net_out), net_out.shape[0], net_out.shape[1],
                                        dim(net_out).stride, fl, ll))

    def Report(self):
        buf = C.create_string_buffer(4096)
        _ok(lib.aslp_warpctc_report(self.h, buf, 4096))
        return buf.value.decode()

    def GetStats(self):
        st = (C.c_double * 5)()
        _ok(lib.aslp_warpctc_get_stats(self.h, st))
        return dict(zip(("obj", "frames", "sequences", "error_tokens", "ref_tokens"), list(st)))


class Ctc:
    """aslp-nnet/ctc-loss.h:39 (Eesen-style): Eval / EvalParallel / ErrorRate[MSeq] / Report on Softmax outputs."""

    def __init__(self):
        self.h = _H()
        _ok(lib.aslp_eesenctc_create(C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.aslp_eesenctc_free(self.h)
            self.h = None

    def _eval(self, frame_num_utt, net_out, labels, diff):
        import numpy as np
        _chk(net_out)
        if diff is None:
            diff = torch.empty_like(net_out)
        n = len(labels) if frame_num_utt is None else len(frame_num_utt)
        fl, ll = _flat(labels[:n])
        costs = np.zeros(n, np.float32)
        _ok(lib.aslp_eesenctc_eval(self.h, None if frame_num_utt is None else _i32arr(frame_num_utt), n, ptr(net_out), net_out.shape[0],
                                   net_out.shape[1], dim(net_out).stride, fl, ll, ptr(diff), dim(diff).stride,
                                   costs.ctypes.data_as(C.POINTER(_f))))
        return diff, costs

    def Eval(self, net_out, label, diff=None):
        return self._eval(None, net_out, [label], diff)

    def EvalParallel(self, frame_num_utt, net_out, labels, diff=None):
        return self._eval(frame_num_utt, net_out, labels, diff)

    def ErrorRate(self, net_out, label):
        fl, ll = _flat([label])
        _ok(lib.aslp_eesenctc_error_rate(self.h, None, 1, ptr(net_out), net_out.shape[0], net_out.shape[1], dim(net_out).stride, fl, ll))

    def ErrorRateMSeq(self, frame_num_utt, net_out, labels):
        n = len(frame_num_utt)
        fl, ll = _flat(labels[:n])
        _ok(lib.aslp_eesenctc_error_rate(self.h, _i32arr(frame_num_utt), n, ptr(net_out), net_out.shape[0], net_out.shape[1],
                                         dim(net_out).stride, fl, ll))

    def Report(self):
        buf = C.create_string_buffer(4096)
        _ok(lib.aslp_eesenctc_report(self.h, buf, 4096))
        return buf.value.decode()

    def GetStats(self):
        st = (C.c_double * 5)()
        _ok(lib.aslp_eesenctc_get_stats(self.h, st))
        return dict(zip(("obj", "frames", "sequences", "error_tokens", "ref_tokens"), list(st)))


class Nnet:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def Init(cls, proto_text, seed=777):
        h = _H()
        _ok(lib.aslp_nnet_init_from_proto(proto_text.encode(), seed, C.byref(h)))
        return cls(h)

    @classmethod
    def Read(cls, path):
        h = _H()
        _ok(lib.aslp_nnet_read(str(path).encode(), C.byref(h)))
        return cls(h)

    def Copy(self):
        h = _H()
        _ok(lib.aslp_nnet_copy(self.h, C.byref(h)))
        return Nnet(h)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib.aslp_nnet_free(self.h)
            self.h = None

    def Write(self, path, binary=True):
        _ok(lib.aslp_nnet_write(self.h, str(path).encode(), int(binary)))

    def SetTrainOptions(self, learn_rate=0.008, momentum=0.0, l2_penalty=0.0, l1_penalty=0.0):
        _ok(lib.aslp_nnet_set_train_options(self.h, learn_rate, momentum, l2_penalty, l1_penalty))

    def InputDim(self): return lib.aslp_nnet_input_dim(self.h)
    def OutputDim(self): return lib.aslp_nnet_output_dim(self.h)
    def NumComponents(self): return lib.aslp_nnet_num_components(self.h)
    def NumParams(self): return lib.aslp_nnet_num_params(self.h)

    def Marker(self, c):
        buf = C.create_string_buffer(128)
        _ok(lib.aslp_nnet_component_marker(self.h, c, buf, 128))
        return buf.value.decode()

    def Info(self):
        buf = C.create_string_buffer(1 << 16)
        _ok(lib.aslp_nnet_info(self.h, buf, 1 << 16))
        return buf.value.decode()

    def SetLinkAliasing(self, on): _ok(lib.aslp_nnet_set_link_aliasing(self.h, int(on)))
    def SetLayerFusion(self, on): _ok(lib.aslp_nnet_set_layer_fusion(self.h, int(on)))
    def SetUpdateOverlap(self, on): _ok(lib.aslp_nnet_set_update_overlap(self.h, int(on)))

    def Propagate(self, x, out=None):
        _chk(x)
        if out is None:
            out = torch.empty(x.shape[0], self.OutputDim(), device=x.device)
        _ok(lib.aslp_nnet_propagate(self.h, ptr(x), x.shape[0], x.shape[1], dim(x).stride, ptr(out), dim(out).stride))
        return out

    def Feedforward(self, x, out=None):
        _chk(x)
        if out is None:
            out = torch.empty(x.shape[0], self.OutputDim(), device=x.device)
        _ok(lib.aslp_nnet_feedforward(self.h, ptr(x), x.shape[0], x.shape[1], dim(x).stride, ptr(out), dim(out).stride))
        return out

    def Backpropagate(self, out_diff, want_in_diff=False):
        _chk(out_diff)
        in_diff = torch.empty(out_diff.shape[0], self.InputDim(), device=out_diff.device) if want_in_diff else None
        _ok(lib.aslp_nnet_backpropagate(self.h, ptr(out_diff), out_diff.shape[0], out_diff.shape[1], dim(out_diff).stride,
                                        ptr(in_diff), dim(in_diff).stride if in_diff is not None else 0))
        return in_diff

    def ResetLstmStreams(self, flags):
        a = (C.c_int32 * len(flags))(*[int(f) for f in flags])
        _ok(lib.aslp_nnet_reset_lstm_streams(self.h, a, len(flags)))

    def SetSeqLengths(self, lens):
        a = (C.c_int32 * len(lens))(*[int(f) for f in lens])
        _ok(lib.aslp_nnet_set_seq_lengths(self.h, a, len(lens)))

    def SetChunkSize(self, n): _ok(lib.aslp_nnet_set_chunk_size(self.h, int(n)))

    def GetParams(self):
        import numpy as np
        n = self.NumParams()
        buf = np.empty(n, np.float32)
        _ok(lib.aslp_nnet_get_params(self.h, buf.ctypes.data_as(C.POINTER(_f)), n))
        return buf

    def GetGpuParams(self, writers_announce=False):
        """[(device_ptr, n_floats)] in the reference's tensor order (nnet-nnet.cc:314-325).  writers_announce: the caller promises
        lib.aslp_params_changed() after every write through these pointers (include/aslp_nnet.h); without it the net stops keeping
        anything derived from its weights from step to step (always correct, a little slower)."""
        n = lib.aslp_nnet_get_gpu_params(self.h, None, None, 0)
        if n < 0:
            _ok(1)
        ptrs = (_vp * n)()
        sizes = (_i * n)()
        lib.aslp_nnet_get_gpu_params(self.h, ptrs, sizes, n)
        if writers_announce:
            _ok(lib.aslp_nnet_param_writers_announce(self.h))
        return [(ptrs[i], sizes[i]) for i in range(n)]

    def ComponentOutput(self, c, rows, cols):
        import numpy as np
        buf = np.empty((rows, cols), np.float32)
        _ok(lib.aslp_nnet_component_output(self.h, c, buf.ctypes.data_as(C.POINTER(_f)), rows, cols))
        return buf

    def ComponentOutDiff(self, c, rows, cols):
        import numpy as np
        buf = np.empty((rows, cols), np.float32)
        _ok(lib.aslp_nnet_component_out_diff(self.h, c, buf.ctypes.data_as(C.POINTER(_f)), rows, cols))
        return buf

    def TrainStepWarpCtc(self, ctc, x, frame_num_utt, labels):
        """SetSeqLengths -> Propagate -> WarpCtc::Eval + ErrorRate -> Backpropagate(+Update)."""
        _chk(x)
        n = len(frame_num_utt)
        fl, ll = _flat(labels[:n])
        _ok(lib.aslp_nnet_train_step_warpctc(self.h, ctc.h, ptr(x), x.shape[0], x.shape[1], dim(x).stride, _i32arr(frame_num_utt), n, fl, ll))

    def TrainStepXent(self, xent, x, labels, frame_weights=None):
        """Propagate -> Xent::Eval -> Backpropagate(+Update), all on the device."""
        _chk(x), _chk(labels, torch.int32)
        _ok(lib.aslp_nnet_train_step_xent(self.h, xent.h, ptr(x), x.shape[0], x.shape[1], dim(x).stride, ptr(labels), ptr(frame_weights)))
