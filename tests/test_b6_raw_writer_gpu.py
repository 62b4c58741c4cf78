"""Seam B6 (src/aslp-parallel/itf.h:26-42) driven the way the reference's own workers drive it: `InitParam` keeps the raw device
pointers `Nnet::GetGpuParams` handed out ONCE (bsp-worker.h:45-48), and every `Synchronize` reads and writes the model through them with
blocking null-stream copies (`CuSubVector::CopyToVec` / `CopyFromVec`, bsp-worker.cc:41-55) -- no Nnet accessor, no announcement, no
device-wide synchronisation in between.  The engine runs its weight updates on a non-blocking side stream; a training step must therefore
END with those updates ordered in front of everything the null stream does next whenever the pointers are out with a silent writer."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_nnet_gpu import make_dnn, oracle_params

pytestmark = pytest.mark.gpu
TOL = 1e-4
hipMemcpyHostToDevice, hipMemcpyDeviceToHost = 1, 2


def _hip_memcpy(aslp):
    """hipMemcpy of the HIP runtime the product library itself is linked to (dlsym through its handle)."""
    f = aslp.lib.hipMemcpy
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return f


class RawBspWriter:
    """bsp-worker.cc:33-58 for replicas that live in one process: per tensor CopyToVec (D2H), scale by n_k / sum n on the host,
    sum over the replicas (what MPI_Allreduce returns), CopyFromVec (H2D)."""

    def __init__(self, aslp, param_sets, order="as_handed_out"):
        self.memcpy = _hip_memcpy(aslp)
        self.param_sets = param_sets       # one [(ptr, n)] per replica, taken once
        # the interface promises no tensor order: the reference walks the list as handed out; a worker that reduces in buckets from the output
        # layer down (the order gradients become ready in) starts with the tensors whose updates were issued on the side stream
        idx = list(range(len(param_sets[0])))
        self.order = idx if order == "as_handed_out" else idx[::-1]

    def Synchronize(self, samples):
        total = float(sum(samples))
        for t in self.order:
            n = self.param_sets[0][t][1]
            if n == 0:
                continue
            acc = np.zeros(n, np.float32)
            host = np.empty(n, np.float32)
            for k in reversed(range(len(self.param_sets))):     # the replica that stepped last is read first, as its own process would
                assert self.memcpy(host.ctypes.data, self.param_sets[k][t][0], 4 * n, hipMemcpyDeviceToHost) == 0
                acc += host * np.float32(samples[k] / total)
            for ps in self.param_sets:
                assert self.memcpy(ps[t][0], acc.ctypes.data, 4 * n, hipMemcpyHostToDevice) == 0


def _oracle_tensors(oracle, d, bn):
    """mutable views of the oracle replica's parameters, GetGpuParams order"""
    L = oracle.lib.orc_dnn_num_layers(d)
    out = []
    for l in range(L):
        r, c = C.c_int(), C.c_int()
        wp = oracle.lib.orc_dnn_weight(d, l, C.byref(r), C.byref(c))
        out.append(np.ctypeslib.as_array(wp, shape=(r.value * c.value,)))
        out.append(np.ctypeslib.as_array(oracle.lib.orc_dnn_bias(d, l), shape=(r.value,)))
        if bn and l < L - 1:
            out.append(np.ctypeslib.as_array(oracle.lib.orc_dnn_bn_shift(d, l), shape=(r.value,)))
            out.append(np.ctypeslib.as_array(oracle.lib.orc_dnn_bn_scale(d, l), shape=(r.value,)))
    return out


# (40, 4096, 1, 3000, 1024): a tiny lowest layer under a large output layer -- the main stream is done with a step ~80 us before the side stream's
# launch has written the output layer's weights and bias, the widest window a silent writer can fall into; walking the tensors from the top
# it reads that bias within microseconds of the step's last main-stream kernel
@pytest.mark.parametrize("order", ["as_handed_out", "top_down"])
@pytest.mark.parametrize("dims", [(440, 2048, 3, 3000, 1024), (40, 4096, 1, 3000, 1024)])
def test_reference_shaped_bsp_writer_needs_no_sync(aslp, oracle, dev, tmp_path, dims, order):
    """Two replicas, five steps of BSP with a sync after every step, against two oracle replicas averaged in closed form.  Nothing but
    `TrainStepXent` and null-stream `hipMemcpy` touches the device inside the loop."""
    in_dim, hid, nh, out_dim, mb = dims
    bn, lr, K = 1, 0.008, 2
    ds, nets, xes, psets = [], [], [], []
    for k in range(K):
        d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, bn, mb, seed=21)   # same initial model on every replica
        net = aslp.Nnet.Read(path)
        net.SetTrainOptions(learn_rate=lr, momentum=0.0)
        ds.append(d); nets.append(net); xes.append(aslp.Xent())
        psets.append(net.GetGpuParams())          # once, silently: what InitParam keeps
    # (weight matrices are handed out with their row pitch, nnet-affine-transform.h:166-170: the writer averages the padding with the rest,
    #  like the reference's; the closed form below runs on the oracle's dense tensors and is compared through GetParams)
    assert len(psets[0]) == len(_oracle_tensors(oracle, ds[0], bn))
    writer = RawBspWriter(aslp, psets, order)
    rng = np.random.default_rng(17)
    batches = [[(rng.standard_normal((mb, in_dim)).astype(np.float32), rng.integers(0, out_dim, mb).astype(np.int32)) for _ in range(K)]
               for _ in range(5)]
    dev_batches = [[(torch.from_numpy(x).to(dev), torch.from_numpy(l).to(dev)) for x, l in step] for step in batches]
    torch.cuda.synchronize()
    prev = oracle_params(oracle, ds[0], bn).astype(np.float64)
    for step in range(5):
        for k in range(K):
            nets[k].TrainStepXent(xes[k], *dev_batches[step][k])
        writer.Synchronize([mb] * K)              # right behind the last step: no synchronize, no accessor
        # closed form on the oracle replicas
        for k in range(K):
            oracle.lib.orc_dnn_train_step(ds[k], batches[step][k][0], batches[step][k][1], lr, 0.0)
        tens = [_oracle_tensors(oracle, d, bn) for d in ds]
        for t in range(len(tens[0])):
            avg = sum(tens[k][t] * np.float32(1.0 / K) for k in range(K)).astype(np.float32)
            for k in range(K):
                tens[k][t][:] = avg
        # (checked after the exchange, so nothing here orders the step in front of it.)  The model after the exchange, and -- far more
        # sensitive to a tensor that was read or overwritten while its update was still running -- what the step and the exchange together
        # moved it by, read back as (before - after) / lr: a stale tile is a missing update, an error of order one there
        want = oracle_params(oracle, ds[0], bn)
        g_o = (prev - want) / lr
        for k in range(K):
            got = nets[k].GetParams()
            assert oracle.rel_err(got, want) < TOL and oracle.max_err(got, want) < 10 * TOL, (step, k)
            g_e = (prev - got) / lr
            assert np.linalg.norm(g_e - g_o) / np.linalg.norm(g_o) < 10 * TOL + 2.0 ** -23 * np.linalg.norm(want) / lr / np.linalg.norm(g_o), ("applied update", step, k)
        assert np.array_equal(nets[0].GetParams(), nets[1].GetParams()), step
        prev = want.astype(np.float64)
    for d in ds:
        oracle.lib.orc_dnn_destroy(d)
