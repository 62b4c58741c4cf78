"""ctypes binding of libaslp_parallel.so (include/aslp_parallel.h): the product's own model-sync layer -- RcclComm and the
BSP / BMUF / EASGD / ASGD workers of kaldi-aslp_amd/parallel/{comm,workers}.cpp -- for Python hosts (bench.py --gpus N,
tests).  No arithmetic here and no torch.distributed: the collectives are the library's ncclAllReduce / ncclSend /
ncclRecv calls on the tensors where they live.  The library is loaded on first use (it pulls in RCCL) and a missing
library is an error, not a fallback.

Interface names follow src/aslp-parallel/itf.h:26-42 (InitParam / Synchronize / Stop / Rank / NumNodes / IsMainNode).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaslp_parallel.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        from . import _lib as hip  # libaslp_hip.so first (RTLD_GLOBAL): libaslp_parallel.so links against it
        assert hip.lib is not None
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libaslp_parallel.so not found at %s: build it with `make -C kaldi-aslp_amd`. There is no fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.aslp_parallel_last_error.restype = C.c_char_p
        for name, args in [
            ("aslp_comm_create_rccl", [i, i, C.c_char_p, C.c_char_p, i, C.POINTER(vp)]),
            ("aslp_comm_create_shm", [i, i, C.c_char_p, C.c_char_p, i, C.POINTER(vp)]),
            ("aslp_comm_rank", [vp]), ("aslp_comm_num_nodes", [vp]), ("aslp_comm_barrier", [vp]), ("aslp_comm_ranks_seen", [vp]),
            ("aslp_comm_allreduce_sum_f32", [vp, vp, sz]), ("aslp_comm_allreduce_sum_f64", [vp, vp, sz]),
            ("aslp_comm_allreduce_sum_host_i32", [vp, C.POINTER(C.c_int32), sz]),
            ("aslp_comm_allreduce_sum_host_f64", [vp, C.POINTER(C.c_double), sz]),
            ("aslp_comm_send_f32", [vp, i, vp, sz]), ("aslp_comm_recv_f32", [vp, i, vp, sz]),
            ("aslp_comm_exchange_f32", [vp, i, vp, vp, sz]),
            ("aslp_worker_create", [vp, C.c_char_p, f, f, C.POINTER(vp)]),
            ("aslp_worker_init_param", [vp, C.POINTER(vp), C.POINTER(i), i]),
            ("aslp_worker_init_param_nnet", [vp, vp]),
            ("aslp_worker_synchronize", [vp, i, C.POINTER(i)]), ("aslp_worker_stop", [vp]),
            ("aslp_server_run", [vp, C.c_char_p, f, f, i, C.POINTER(vp), C.POINTER(i), i]),
        ]:
            fn = getattr(L, name)
            fn.restype = i
            fn.argtypes = args
        L.aslp_comm_transport.restype = C.c_char_p
        L.aslp_comm_transport.argtypes = [vp]
        L.aslp_comm_free.restype = None
        L.aslp_comm_free.argtypes = [vp]
        L.aslp_worker_free.restype = None
        L.aslp_worker_free.argtypes = [vp]
        _lib = L
    return _lib


class _StdoutToStderr:
    """RCCL prints a version banner on the C-level stdout when a communicator comes up or goes away; a host whose stdout
    is a protocol (bench.py: one JSON line) gets it on stderr instead."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def _ok(rc):
    if rc != 0:
        raise RuntimeError((lib().aslp_parallel_last_error() or b"").decode(errors="replace"))


class RcclComm:
    """mpi-node.h:19-97 on RCCL.  rank / num_nodes default to the launcher's environment (RANK / WORLD_SIZE, PMI_*, OMPI_*)."""

    _create = "aslp_comm_create_rccl"

    def __init__(self, id_file=None, rank=-1, num_nodes=-1, token=None, timeout_s=900):
        self.h = C.c_void_p()
        L = lib()
        with _StdoutToStderr():
            rc = getattr(L, self._create)(rank, num_nodes, id_file.encode() if id_file else None, token.encode() if token else None,
                                          timeout_s, C.byref(self.h))
        _ok(rc)

    def close(self):
        if getattr(self, "h", None):
            with _StdoutToStderr():
                lib().aslp_comm_free(self.h)
            self.h = None

    def Rank(self): return lib().aslp_comm_rank(self.h)
    def NumNodes(self): return lib().aslp_comm_num_nodes(self.h)
    def RanksSeen(self): return lib().aslp_comm_ranks_seen(self.h)          # RcclComm: ncclCommCount of the live communicator
    def Transport(self): return lib().aslp_comm_transport(self.h).decode()  # "rccl" | "shm"
    def MainNode(self): return 0
    def IsMainNode(self): return self.Rank() == 0
    def Barrier(self): _ok(lib().aslp_comm_barrier(self.h))

    def AllReduce(self, t):
        """in place on a contiguous fp32 / fp64 device tensor"""
        import torch
        assert t.is_contiguous() and t.is_cuda
        if t.dtype == torch.float32:
            _ok(lib().aslp_comm_allreduce_sum_f32(self.h, t.data_ptr(), t.numel()))
        elif t.dtype == torch.float64:
            _ok(lib().aslp_comm_allreduce_sum_f64(self.h, t.data_ptr(), t.numel()))
        else:
            raise TypeError("AllReduce: fp32 / fp64 device tensors only")
        return t

    def AllReduceHostInt(self, values):
        a = (C.c_int32 * len(values))(*values)
        _ok(lib().aslp_comm_allreduce_sum_host_i32(self.h, a, len(values)))
        return list(a)

    def AllReduceHostDouble(self, values):
        a = (C.c_double * len(values))(*values)
        _ok(lib().aslp_comm_allreduce_sum_host_f64(self.h, a, len(values)))
        return list(a)

    def MaxOverRanks(self, x):
        """max of one host double per rank (every rank gets it): a sum all-reduce of a one-hot vector, the only reduction MpiNode has"""
        v = [0.0] * self.NumNodes()
        v[self.Rank()] = float(x)
        return max(self.AllReduceHostDouble(v))

    def Send(self, peer, t): _ok(lib().aslp_comm_send_f32(self.h, peer, t.data_ptr(), t.numel()))
    def Recv(self, peer, t): _ok(lib().aslp_comm_recv_f32(self.h, peer, t.data_ptr(), t.numel()))
    def Exchange(self, peer, send, recv): _ok(lib().aslp_comm_exchange_f32(self.h, peer, send.data_ptr(), recv.data_ptr(), send.numel()))


class ShmComm(RcclComm):
    """The same communicator for ranks that are separate processes sharing GPUs (all on one device included): rendezvous file and
    control pipe as RcclComm, tensors staged through a shared-memory segment (parallel/comm.cpp ShmComm)."""
    _create = "aslp_comm_create_shm"


def ProcessComm(id_file=None, rank=-1, num_nodes=-1, token=None, timeout_s=900, transport=None):
    """RcclComm unless transport / ASLP_COMM_TRANSPORT says "shm" (what the worker tools do)"""
    t = transport or os.environ.get("ASLP_COMM_TRANSPORT") or "rccl"
    return (ShmComm if t == "shm" else RcclComm)(id_file, rank=rank, num_nodes=num_nodes, token=token, timeout_s=timeout_s)


class Worker:
    """itf.h:26-42 over the native workers.  kind: "bsp" | "bmuf" (learn_rate, momentum) | "easgd" (alpha) | "asgd"."""

    def __init__(self, comm, kind, p0=0.0, p1=0.0):
        self.comm = comm
        self.h = C.c_void_p()
        _ok(lib().aslp_worker_create(comm.h, kind.encode(), p0, p1, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().aslp_worker_free(self.h)
            self.h = None

    def InitParam(self, params):
        """params: an Nnet (its GetGpuParams tensors are aliased) or a list of contiguous fp32 device tensors / (ptr, n) pairs"""
        if hasattr(params, "h") and not isinstance(params, (list, tuple)):
            _ok(lib().aslp_worker_init_param_nnet(self.h, params.h))
            return
        pairs = [(p.data_ptr(), p.numel()) if hasattr(p, "data_ptr") else (int(p[0]), int(p[1])) for p in params]
        self._keep = params
        ptrs = (C.c_void_p * len(pairs))(*[p for p, _ in pairs])
        sizes = (C.c_int * len(pairs))(*[n for _, n in pairs])
        _ok(lib().aslp_worker_init_param(self.h, ptrs, sizes, len(pairs)))

    def Synchronize(self, num_worker_samples):
        more = C.c_int(0)
        _ok(lib().aslp_worker_synchronize(self.h, int(num_worker_samples), C.byref(more)))
        return bool(more.value)

    def Stop(self): _ok(lib().aslp_worker_stop(self.h))
    def Rank(self): return self.comm.Rank()
    def NumNodes(self): return self.comm.NumNodes()
    def IsMainNode(self): return self.comm.IsMainNode()


def BspWorker(comm): return Worker(comm, "bsp")
def BmufWorker(comm, learn_rate, momentum): return Worker(comm, "bmuf", learn_rate, momentum)
def EasgdWorker(comm, alpha): return Worker(comm, "easgd", alpha)
def AsgdWorker(comm): return Worker(comm, "asgd")


def ServerRun(comm, kind, params, p0=0.5, p1=0.0, sync_period=0):
    """rank 0 of easgd / asgd / masgd: serves until every worker has finished (easgd-server.cc:63-86)"""
    pairs = [(p.data_ptr(), p.numel()) for p in params]
    ptrs = (C.c_void_p * len(pairs))(*[p for p, _ in pairs])
    sizes = (C.c_int * len(pairs))(*[n for _, n in pairs])
    _ok(lib().aslp_server_run(comm.h, kind.encode(), p0, p1, sync_period, ptrs, sizes, len(pairs)))
