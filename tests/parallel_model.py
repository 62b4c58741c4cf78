"""TEST INFRASTRUCTURE: a torch.distributed model of the data-parallel model-sync workers -- BSP, BMUF, EASGD (worker + server) -- for
the world-size-2 gloo tests on CPU (tests/test_parallel_cpu.py).  The product's workers are kaldi-aslp_amd/parallel/*.cpp.

Mirror of the reference's aslp-parallel interface (src/aslp-parallel/itf.h:26-42):
`InitParam(params)`, `Synchronize(num_worker_samples) -> bool`, `Stop()`, `ReduceAccStat(...)`,
`Rank() / NumNodes() / IsMainNode()`.  The reference stages every parameter tensor through the
host and issues one blocking MPI_Allreduce per tensor (mpi-node.h:68-75, bsp-worker.cc:46-55);
here the tensors are packed into ONE flat device buffer and reduced with ONE collective over
`torch.distributed` -- backend "nccl" is RCCL over xGMI on the 8 GPUs of a node; backend "gloo" on
CPU tensors is used by the world_size-2 protocol tests.  One process per GPU; the model memory is
aliased, never owned (like the reference's CuSubVector, bsp-worker.h:45-48).

The protocol quirks are kept: the int sample count is all-reduced first; a global count of 0 ends
the run; a worker that ran out of data keeps calling Synchronize(0) (contributing a zero-scaled
model) until everybody has (bsp-worker.cc:33-65).
"""
import torch
import torch.distributed as dist


class _DevicePtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, n, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}


def alias_device_params(gpu_params, typestr="<f4"):
    """[(device_ptr, n)] (Nnet::GetGpuParams order) -> list of 1-D torch tensors aliasing that memory."""
    out = []
    for p, n in gpu_params:
        if n == 0:
            continue
        out.append(torch.as_tensor(_DevicePtr(p, n, typestr), device="cuda"))
    return out


class MpiNodeLike:
    """mpi-node.h:19-97 on torch.distributed (the process group must already be initialised)."""

    def __init__(self, group=None):
        self.group = group
        self.rank_ = dist.get_rank(group) if dist.is_initialized() else 0
        self.num_nodes_ = dist.get_world_size(group) if dist.is_initialized() else 1

    def Rank(self): return self.rank_
    def NumNodes(self): return self.num_nodes_
    def MainNode(self): return 0
    def IsMainNode(self): return self.rank_ == 0

    def Barrier(self):
        if self.num_nodes_ > 1:
            dist.barrier(self.group)

    def AllReduce(self, t):
        if self.num_nodes_ > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def ReduceAccStat(self, counts, data):
        """BatchNorm running statistics at the end of an epoch (mpi-node.h:77-93): `counts` is a list of
        python floats (num_acc_frames per BN layer), `data` a list of double tensors; both are summed
        over ranks.  Returns the reduced counts."""
        self.Barrier()
        if not data and not counts:
            return counts
        dev = data[0].device if data else torch.device("cpu")
        c = torch.tensor(list(counts), dtype=torch.float64, device=dev)
        self.AllReduce(c)
        if data:
            flat = torch.cat([d.reshape(-1) for d in data])
            self.AllReduce(flat)
            o = 0
            for d in data:
                d.copy_(flat[o:o + d.numel()].view_as(d))
                o += d.numel()
        return c.tolist()


class _FlatParams:
    def __init__(self, params):
        self.params = [p.reshape(-1) for p in params]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.device = dev

    def pack(self, scale=None):
        if not self.params:
            return self.flat
        torch.cat(self.params, out=self.flat)
        if scale is not None:
            self.flat.mul_(scale)
        return self.flat

    def unpack(self, src=None):
        src = self.flat if src is None else src
        o = 0
        for p in self.params:
            p.copy_(src[o:o + p.numel()])
            o += p.numel()


def _params_of(model_or_params):
    if hasattr(model_or_params, "GetGpuParams"):
        return alias_device_params(model_or_params.GetGpuParams())
    return list(model_or_params)


class BspWorker(MpiNodeLike):
    """Synchronous weighted model averaging (bsp-worker.cc:33-65)."""

    def __init__(self, model_or_params=None, group=None):
        super().__init__(group)
        self.fp = None
        if model_or_params is not None:
            self.InitParam(model_or_params)

    def InitParam(self, model_or_params):
        self.fp = _FlatParams(_params_of(model_or_params))

    def Synchronize(self, num_worker_samples):
        n = torch.tensor([int(num_worker_samples)], dtype=torch.int32, device=self.fp.device)
        self.AllReduce(n)
        num_all = int(n.item())
        if num_all <= 0:
            return False  # all workers finished their data
        factor = float(num_worker_samples) / num_all
        assert 0.0 <= factor <= 1.0
        flat = self.fp.pack(scale=factor)   # theta_k * n_k / sum n
        self.AllReduce(flat)                # one collective for the whole model
        self.fp.unpack()
        return True

    def Stop(self):
        while self.Synchronize(0):
            pass


class BmufWorker(MpiNodeLike):
    """Block-momentum update filtering (bmuf-worker.cc:37-68)."""

    def __init__(self, model_or_params=None, learn_rate=1.0, momentum=0.9, group=None):
        super().__init__(group)
        self.learn_rate_, self.momentum_ = learn_rate, momentum
        if model_or_params is not None:
            self.InitParam(model_or_params)

    def InitParam(self, model_or_params):
        self.fp = _FlatParams(_params_of(model_or_params))
        self.prev = self.fp.pack().clone()           # w_g(t-1)
        self.prev_grad = torch.zeros_like(self.prev)  # d(t-1)

    def Synchronize(self, num_worker_samples):
        n = torch.tensor([int(num_worker_samples)], dtype=torch.int32, device=self.fp.device)
        self.AllReduce(n)
        if int(n.item()) <= 0:
            return False
        grad = self.fp.pack()
        grad.sub_(self.prev)                 # 1. g = w(t) - w_g(t-1)
        self.AllReduce(grad)                 # 2. SUM over workers (not a mean: bmuf-worker.cc:49)
        lr = (1.0 - self.momentum_) * self.learn_rate_
        grad.mul_(lr).add_(self.prev_grad, alpha=self.momentum_)   # 4. d(t) = m d(t-1) + (1-m) lr g
        self.prev.add_(grad)                 # 5. w(t) = w_g(t-1) + d(t)
        self.fp.unpack(self.prev)
        self.prev_grad.copy_(grad)           # 6.
        return True

    def Stop(self):
        while self.Synchronize(0):
            pass


class SodWorker(MpiNodeLike):
    """"Synchronous optimize the difference" (sod-worker.cc:36-68): the summed model deltas w(t-1) - w(t) are handed to one of
    the solvers of optimizer.h:40-171 as a gradient, and the solver steps this worker's own current model."""

    DEFAULTS = dict(solver="momentum", lr=0.01, momentum=0.9, adagrad_lr=0.01, rmsprop_lr=0.001, adam_lr=0.001, adadelta_gamma=0.95,
                    adam_beta1=0.9, adam_beta2=0.999)   # optimizer.h:180-184

    def __init__(self, model_or_params=None, group=None, **config):
        super().__init__(group)
        unknown = set(config) - set(self.DEFAULTS)
        if unknown:
            raise ValueError("unknown optimizer options: %s" % sorted(unknown))
        self.cfg = dict(self.DEFAULTS, **config)
        if self.cfg["solver"] not in ("sgd", "momentum", "adagrad", "rmsprop", "adadelta", "adam"):
            raise ValueError("Unknown solver type %s" % self.cfg["solver"])
        self.t = 1
        if model_or_params is not None:
            self.InitParam(model_or_params)

    def InitParam(self, model_or_params):
        self.fp = _FlatParams(_params_of(model_or_params))
        self.prev = self.fp.pack().clone()
        self.s1, self.s2 = torch.zeros_like(self.prev), torch.zeros_like(self.prev)

    def _solve(self, g, w):
        c, k = self.cfg, self.cfg["solver"]
        floor_rsqrt = lambda v: v.clamp(min=1e-8).sqrt_().reciprocal_()
        if k == "sgd":
            w.add_(g, alpha=-c["lr"])
        elif k == "momentum":
            self.s1.mul_(c["momentum"]).add_(g, alpha=c["lr"])
            w.sub_(self.s1)
        elif k in ("adagrad", "rmsprop"):
            if k == "adagrad":
                self.s1.addcmul_(g, g)
            else:
                self.s1.mul_(0.9).addcmul_(g, g, value=0.1)
            w.add_(floor_rsqrt(self.s1).mul_(g), alpha=-(c["adagrad_lr"] if k == "adagrad" else c["rmsprop_lr"]))
        elif k == "adadelta":
            gm = c["adadelta_gamma"]
            self.s1.mul_(gm).addcmul_(g, g, value=1 - gm)
            d = floor_rsqrt(self.s1).mul_(self.s2.clamp(min=1e-8).sqrt_()).mul_(g)
            w.sub_(d)
            self.s2.mul_(gm).addcmul_(d, d, value=1 - gm)
        else:
            b1, b2 = c["adam_beta1"], c["adam_beta2"]
            self.s1.mul_(b1).add_(g, alpha=1 - b1)
            self.s2.mul_(b2).addcmul_(g, g, value=1 - b2)
            w.add_(floor_rsqrt(self.s2 * (1.0 / (1 - b2 ** self.t))).mul_(self.s1), alpha=-c["adam_lr"] / (1 - b1 ** self.t))

    def Synchronize(self, num_worker_samples):
        n = torch.tensor([int(num_worker_samples)], dtype=torch.int32, device=self.fp.device)
        self.AllReduce(n)
        if int(n.item()) <= 0:
            return False
        w = self.fp.pack()
        grad = self.prev - w                 # 1. w(t-1) - w(t)
        self.AllReduce(grad)                 # 2. summed over workers
        self._solve(grad, w)                 # 4. on the LOCAL model: workers are not re-unified
        self.t += 1
        self.fp.unpack()
        self.prev.copy_(w)                   # 5.
        return True

    def Stop(self):
        while self.Synchronize(0):
            pass


K_MSG_SYNCHRONIZE, K_MSG_FINISHED = 0, 1   # itf.h:19-22


class EasgdWorker(MpiNodeLike):
    """Elastic-averaging SGD worker (easgd-worker.cc:37-80): rank 0 is the server."""

    def __init__(self, model_or_params=None, alpha=0.5, group=None):
        super().__init__(group)
        self.alpha_ = alpha
        if model_or_params is not None:
            self.InitParam(model_or_params)

    def InitParam(self, model_or_params):
        self.fp = _FlatParams(_params_of(model_or_params))
        self.server = torch.empty_like(self.fp.flat)

    def Synchronize(self, num_worker_samples=0):
        msg = torch.tensor([K_MSG_SYNCHRONIZE], dtype=torch.int32, device=self.fp.device)
        dist.send(msg, dst=self.MainNode(), group=self.group)
        w = self.fp.pack()
        # worker -> server and server -> worker (MPI_Sendrecv per tensor in the reference; one flat
        # exchange here).  The server posts the matching recv/send pair.
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, w, self.MainNode(), self.group),
                                       dist.P2POp(dist.irecv, self.server, self.MainNode(), self.group)])
        for r in reqs:
            r.wait()
        w.mul_(1.0 - self.alpha_).add_(self.server, alpha=self.alpha_)   # x_w = (1-a) x_w + a x_s
        self.fp.unpack()
        return True

    def Stop(self):
        msg = torch.tensor([K_MSG_FINISHED], dtype=torch.int32, device=self.fp.device)
        dist.send(msg, dst=self.MainNode(), group=self.group)


class EasgdServer(MpiNodeLike):
    """easgd-server.cc:37-86: serves workers in arrival order until all have sent kMsgFinished."""

    def __init__(self, model_or_params=None, alpha=0.5, group=None):
        super().__init__(group)
        self.alpha_ = alpha
        if model_or_params is not None:
            self.InitParam(model_or_params)

    def InitParam(self, model_or_params):
        self.fp = _FlatParams(_params_of(model_or_params))
        self.worker = torch.empty_like(self.fp.flat)

    def Run(self):
        num_running = self.NumNodes() - 1
        msg = torch.zeros(1, dtype=torch.int32, device=self.fp.device)
        while num_running > 0:
            src = dist.recv(msg, group=self.group)   # any source
            m = int(msg.item())
            if m == K_MSG_FINISHED:
                num_running -= 1
            elif m == K_MSG_SYNCHRONIZE:
                self.Update(src)
        self.fp.unpack()

    def Update(self, worker_rank):
        s = self.fp.pack()
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, s, worker_rank, self.group),
                                       dist.P2POp(dist.irecv, self.worker, worker_rank, self.group)])
        for r in reqs:
            r.wait()
        s.mul_(1.0 - self.alpha_).add_(self.worker, alpha=self.alpha_)   # x_s = (1-a) x_s + a x_w
        self.fp.unpack()


class AsgdWorker(MpiNodeLike):
    """asgd-worker.cc:37-71: sends the model DELTA since the last exchange to the server (rank 0) and takes the server's
    model in return.  The same worker talks to AsgdServer and MasgdServer."""

    def __init__(self, model_or_params=None, group=None):
        super().__init__(group)
        if model_or_params is not None:
            self.InitParam(model_or_params)

    def InitParam(self, model_or_params):
        self.fp = _FlatParams(_params_of(model_or_params))
        self.prev = self.fp.pack().clone()

    def Synchronize(self, num_worker_samples=0):
        msg = torch.tensor([K_MSG_SYNCHRONIZE], dtype=torch.int32, device=self.fp.device)
        dist.send(msg, dst=self.MainNode(), group=self.group)
        w = self.fp.pack()
        w.sub_(self.prev)                               # w(t) - w(t-1)
        dist.send(w, dst=self.MainNode(), group=self.group)
        dist.recv(w, src=self.MainNode(), group=self.group)   # blocks until the server answers (at once, or at its barrier)
        self.fp.unpack()
        self.prev.copy_(w)
        return True

    def Stop(self):
        msg = torch.tensor([K_MSG_FINISHED], dtype=torch.int32, device=self.fp.device)
        dist.send(msg, dst=self.MainNode(), group=self.group)


class AsgdServer(MpiNodeLike):
    """asgd-server.cc:39-102: x_s += alpha * delta in arrival order.  With sync_period > 0, once sync_period exchanges have
    been counted the server stops answering: each worker that arrives waits, and when all running workers wait they all get
    the same model (a barrier every sync_period exchanges), the count drops by sync_period."""

    def __init__(self, model_or_params=None, alpha=1.0, sync_period=1000, group=None):
        super().__init__(group)
        assert 0.0 <= alpha <= 1.0
        self.alpha_, self.sync_period_ = alpha, sync_period
        if model_or_params is not None:
            self.InitParam(model_or_params)

    def InitParam(self, model_or_params):
        self.fp = _FlatParams(_params_of(model_or_params))
        self.delta = torch.empty_like(self.fp.flat)
        self._init_state()

    def _init_state(self):
        pass

    def _apply(self, server, delta, worker_rank):
        server.add_(delta, alpha=self.alpha_)

    def Run(self):
        num_running = self.NumNodes() - 1
        count, waited = 0, []
        msg = torch.zeros(1, dtype=torch.int32, device=self.fp.device)
        server = self.fp.pack()
        while num_running > 0:
            src = dist.recv(msg, group=self.group)
            m = int(msg.item())
            if m == K_MSG_FINISHED:
                num_running -= 1
            elif m == K_MSG_SYNCHRONIZE:
                count += 1
                if self.sync_period_ > 0 and count >= self.sync_period_:
                    waited.append(src)
                dist.recv(self.delta, src=src, group=self.group)
                self._apply(server, self.delta, src)
                if count < self.sync_period_ or self.sync_period_ <= 0:
                    dist.send(server, dst=src, group=self.group)
            if self.sync_period_ > 0 and count >= self.sync_period_ and len(waited) == num_running and num_running != 0:
                for wr in waited:
                    dist.send(server, dst=wr, group=self.group)
                count -= self.sync_period_
                waited = []
        self.fp.unpack()


class MasgdServer(AsgdServer):
    """masgd-server.cc:93-118 as compiled (MASGD_TYPE == LMASGD, masgd-server.h:23): one momentum buffer PER WORKER,
    d_k = delta + momentum * d_k;  x_s += d_k."""

    def __init__(self, model_or_params=None, sync_period=1000, momentum=0.9, group=None):
        self.momentum_ = momentum
        super().__init__(model_or_params, alpha=1.0, sync_period=sync_period, group=group)

    def _init_state(self):
        self.diffs = [torch.zeros_like(self.fp.flat) for _ in range(self.NumNodes() - 1)]

    def _apply(self, server, delta, worker_rank):
        d = self.diffs[worker_rank - 1]
        d.mul_(self.momentum_).add_(delta)
        server.add_(d)
