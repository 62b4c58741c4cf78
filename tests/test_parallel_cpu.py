"""World-size-2 (and 3) protocol tests on the gloo backend with CPU tensors.  What they run is tests/parallel_model.py, a
torch.distributed MODEL of the sync protocols (test infrastructure, not shipped in the package): the product's workers are C++
on HIP (kaldi-aslp_amd/parallel/*.cpp) and cannot run without a GPU, so on CPU the protocol -- count first, zero-count ranks
keep joining, a global zero ends the run, the update rules -- is pinned on the model, and on the GPU box the SAME closed forms
are applied to the product's workers as separate OS processes (tests/test_rccl_multigpu_gpu.py, transport "shm": three and
four processes on one device) and as threads (tests/test_parallel_gpu.py).  Expected values are the closed forms of the reference's update rules
(src/aslp-parallel/bsp-worker.cc:33-65, bmuf-worker.cc:37-68, easgd-worker.cc:37-67,
easgd-server.cc:63-86, mpi-node.h:77-93), evaluated with numpy in the parent process."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _load_parallel():
    """tests/parallel_model.py: the protocol model (test infrastructure; the product's workers are kaldi-aslp_amd/parallel/*.cpp and need a GPU)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("aslp_parallel", os.path.join(ROOT, "tests", "parallel_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _init(rank, world, port):
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _make_params(rank, shapes, seed=0):
    g = torch.Generator().manual_seed(seed * 100 + rank)
    return [torch.randn(*s, generator=g) for s in shapes]


SHAPES = [(7, 5), (5,), (3, 4), (1,)]


def _bsp_proc(rank, world, port, out):
    _init(rank, world, port)
    P = _load_parallel()
    params = _make_params(rank, SHAPES)
    w = P.BspWorker(params)
    res = []
    # round 1: unequal counts; round 2: rank 1 has run out of data (n = 0) but keeps participating
    counts = [(300, 100), (256, 0)]
    for r, c in enumerate(counts):
        ok = w.Synchronize(c[rank])
        res.append((ok, [p.clone().numpy() for p in params]))
        for p in params:  # local "training" between syncs
            p.add_(0.1 * (rank + 1) * (r + 1))
    # everybody out of data: the drain loop ends on a global count of 0
    w.Stop()
    res.append((w.Synchronize(0), None))
    out[rank] = res
    dist.destroy_process_group()


def test_bsp_world2():
    world, port = 2, _free_port()
    with mp.Manager() as man:
        out = man.dict()
        mp.spawn(_bsp_proc, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    p0, p1 = [[t.numpy() for t in _make_params(r, SHAPES)] for r in range(2)]
    # round 1: theta = (300 th0 + 100 th1) / 400 on both ranks
    want1 = [(np.float32(300 / 400) * a + np.float32(100 / 400) * b) for a, b in zip(p0, p1)]
    for r in range(2):
        ok, got = out[r][0]
        assert ok
        for g, w in zip(got, want1):
            np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-6)
    # round 2: rank 1 contributes a zero-scaled model, everyone gets rank 0's
    w0 = [w + np.float32(0.1) for w in want1]
    for r in range(2):
        ok, got = out[r][1]
        assert ok
        for g, w in zip(got, w0):
            np.testing.assert_allclose(g, w, rtol=1e-6, atol=1e-6)
    assert out[0][2][0] is False and out[1][2][0] is False


def _bmuf_proc(rank, world, port, out):
    _init(rank, world, port)
    P = _load_parallel()
    params = _make_params(0, SHAPES)  # all ranks start from the same model, like the reference
    w = P.BmufWorker(params, learn_rate=1.0, momentum=0.75)
    res = []
    for r in range(2):
        for p in params:
            p.add_(0.01 * (rank + 1) * (r + 1))  # local progress
        w.Synchronize(10)
        res.append([p.clone().numpy() for p in params])
    out[rank] = res
    dist.destroy_process_group()


def test_bmuf_world2():
    world, port = 2, _free_port()
    with mp.Manager() as man:
        out = man.dict()
        mp.spawn(_bmuf_proc, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    m, lr = 0.75, 1.0
    base = [t.numpy() for t in _make_params(0, SHAPES)]
    # step 1: g = sum_k (w_k - w_g) = 0.01 + 0.02; d = (1-m) lr g; w = w_g + d   (sum, not mean: bmuf-worker.cc:49)
    d1 = (1 - m) * lr * 0.03
    # step 2: local progress 0.02 / 0.04 on top of w_g(1) -> g = 0.06; d = m d1 + (1-m) lr g
    d2 = m * d1 + (1 - m) * lr * 0.06
    for r in range(2):
        for g, b in zip(out[r][0], base):
            np.testing.assert_allclose(g, b + d1, rtol=1e-5, atol=1e-6)
        for g, b in zip(out[r][1], base):
            np.testing.assert_allclose(g, b + d1 + d2, rtol=1e-5, atol=1e-6)


def _sod_proc(rank, world, port, out, solver):
    _init(rank, world, port)
    P = _load_parallel()
    params = _make_params(rank, SHAPES, seed=5)
    w = P.SodWorker(params, solver=solver)
    res = []
    for step in range(3):
        for p in params:
            p.add_(0.01 * (rank + 1) * (step + 1))
        ok = w.Synchronize(100 if (rank == 0 or step < 2) else 0)  # rank 1 reports no new frames at the last step, still takes part
        res.append((ok, [p.clone().numpy() for p in params]))
    w.Stop()
    out[rank] = res
    dist.destroy_process_group()


@pytest.mark.parametrize("solver", ["sgd", "momentum", "adagrad", "rmsprop", "adadelta", "adam"])
def test_sod_world2(solver):
    """sod-worker.cc:36-68 + optimizer.h:40-171 over gloo, two ranks: every step the summed deltas -(0.01 + 0.02)(step + 1) go
    through the solver (defaults of optimizer.h:180-184) on each rank's own model."""
    world, port = 2, _free_port()
    with mp.Manager() as man:
        out = man.dict()
        mp.spawn(_sod_proc, args=(world, port, out, solver), nprocs=world, join=True)
        out = dict(out)
    fl = lambda v: max(v, 1e-8)
    for r in range(2):
        w = [t.numpy().astype(np.float64) for t in _make_params(r, SHAPES, seed=5)]
        s1 = s2 = 0.0
        for step in range(3):
            g = -(0.01 + 0.02) * (step + 1)
            local = 0.01 * (r + 1) * (step + 1)
            t = step + 1
            if solver == "sgd":
                d = 0.01 * g
            elif solver == "momentum":
                s1 = 0.01 * g + 0.9 * s1
                d = s1
            elif solver == "adagrad":
                s1 = g * g + s1
                d = 0.01 * g / np.sqrt(fl(s1))
            elif solver == "rmsprop":
                s1 = 0.1 * g * g + 0.9 * s1
                d = 0.001 * g / np.sqrt(fl(s1))
            elif solver == "adadelta":
                s1 = 0.05 * g * g + 0.95 * s1
                d = g * np.sqrt(fl(s2)) / np.sqrt(fl(s1))
                s2 = 0.05 * d * d + 0.95 * s2
            else:
                s1 = 0.1 * g + 0.9 * s1
                s2 = 0.001 * g * g + 0.999 * s2
                d = 0.001 / (1 - 0.9 ** t) * s1 / np.sqrt(fl(s2 / (1 - 0.999 ** t)))
            w = [x + local - d for x in w]
            ok, got = out[r][step]
            assert ok
            for a, b in zip(got, w):
                np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


def _easgd_proc(rank, world, port, out):
    _init(rank, world, port)
    P = _load_parallel()
    params = _make_params(rank, SHAPES, seed=3)
    if rank == 0:
        s = P.EasgdServer(params, alpha=0.5)
        s.Run()
        out[rank] = [p.clone().numpy() for p in params]
    else:
        w = P.EasgdWorker(params, alpha=0.5)
        w.Synchronize()
        w.Stop()
        out[rank] = [p.clone().numpy() for p in params]
    dist.destroy_process_group()


def test_easgd_server_and_one_worker():
    world, port = 2, _free_port()
    with mp.Manager() as man:
        out = man.dict()
        mp.spawn(_easgd_proc, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    s0, w0 = [[t.numpy() for t in _make_params(r, SHAPES, seed=3)] for r in range(2)]
    # both sides move half-way toward the other's PRE-exchange value (easgd-worker.cc:52-60, easgd-server.cc:75-83)
    for g, a, b in zip(out[0], s0, w0):
        np.testing.assert_allclose(g, 0.5 * a + 0.5 * b, rtol=1e-6, atol=1e-6)
    for g, a, b in zip(out[1], s0, w0):
        np.testing.assert_allclose(g, 0.5 * b + 0.5 * a, rtol=1e-6, atol=1e-6)


def _accstat_proc(rank, world, port, out):
    _init(rank, world, port)
    P = _load_parallel()
    node = P.MpiNodeLike()
    data = [torch.full((4,), float(rank + 1), dtype=torch.float64), torch.arange(3, dtype=torch.float64) * (rank + 1)]
    counts = node.ReduceAccStat([100.0 * (rank + 1), 7.0], data)
    out[rank] = (counts, [d.numpy() for d in data], node.Rank(), node.NumNodes(), node.IsMainNode())
    dist.destroy_process_group()


def _asgd_proc(rank, world, port, out, kind):
    _init(rank, world, port)
    P = _load_parallel()
    params = _make_params(0, SHAPES, seed=9)  # everybody starts from the same model
    if rank == 0:
        s = P.AsgdServer(params, alpha=0.5, sync_period=0) if kind == "asgd" else P.MasgdServer(params, sync_period=0, momentum=0.5)
        s.Run()
        out[rank] = [p.clone().numpy() for p in params]
    else:
        w = P.AsgdWorker(params)
        res = []
        for step in range(2):
            for p in params:
                p.add_(0.25 * (step + 1))
            w.Synchronize()
            res.append([p.clone().numpy() for p in params])
        w.Stop()
        out[rank] = res
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["asgd", "masgd"])
def test_asgd_server_and_one_worker(kind):
    """asgd-worker.cc:37-71 against asgd-server.cc:96-102 (x_s += alpha delta) and masgd-server.cc:104-106 (per-worker
    momentum on the deltas): the worker leaves each exchange holding the server's model."""
    world, port = 2, _free_port()
    with mp.Manager() as man:
        out = man.dict()
        mp.spawn(_asgd_proc, args=(world, port, out, kind), nprocs=world, join=True)
        out = dict(out)
    base = [t.numpy() for t in _make_params(0, SHAPES, seed=9)]
    if kind == "asgd":   # server: +0.5 * 0.25, then the worker moves +0.5 from there and the server takes half of it
        after = [0.125, 0.125 + 0.25]
    else:                # d1 = 0.25; d2 = 0.5 + 0.5 * 0.25
        after = [0.25, 0.25 + 0.625]
    for step in range(2):
        for g, b in zip(out[1][step], base):
            np.testing.assert_allclose(g, b + after[step], rtol=1e-6, atol=1e-6)
    for g, b in zip(out[0], base):
        np.testing.assert_allclose(g, b + after[1], rtol=1e-6, atol=1e-6)


def _asgd_barrier_proc(rank, world, port, out):
    _init(rank, world, port)
    P = _load_parallel()
    params = _make_params(0, SHAPES, seed=9)
    if rank == 0:
        P.AsgdServer(params, alpha=1.0, sync_period=2).Run()
        out[rank] = [p.clone().numpy() for p in params]
    else:
        w = P.AsgdWorker(params)
        for p in params:
            p.add_(0.1 * rank)
        w.Synchronize()
        out[rank] = [p.clone().numpy() for p in params]
        w.Stop()
    dist.destroy_process_group()


def test_asgd_periodic_barrier_world3():
    """sync_period = 2 with two workers: the first exchange is answered at once or held, the second completes the period --
    whichever order they arrive in, a worker that was held gets the model that contains BOTH deltas (asgd-server.cc:62-88)."""
    world, port = 3, _free_port()
    with mp.Manager() as man:
        out = man.dict()
        mp.spawn(_asgd_barrier_proc, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    base = [t.numpy() for t in _make_params(0, SHAPES, seed=9)]
    for g, b in zip(out[0], base):
        np.testing.assert_allclose(g, b + 0.3, rtol=1e-6, atol=1e-6)
    got = sorted(float((out[r][0] - base[0]).ravel()[0]) for r in (1, 2))
    # first arrival answered at once (its own delta only), second held until the barrier (both deltas)
    assert abs(got[1] - 0.3) < 1e-5 and (abs(got[0] - 0.1) < 1e-5 or abs(got[0] - 0.2) < 1e-5)


def test_reduce_acc_stat_world2():
    world, port = 2, _free_port()
    with mp.Manager() as man:
        out = man.dict()
        mp.spawn(_accstat_proc, args=(world, port, out), nprocs=world, join=True)
        out = dict(out)
    for r in range(2):
        counts, data, rank, n, main = out[r]
        assert counts == [300.0, 14.0]
        np.testing.assert_array_equal(data[0], np.full(4, 3.0))
        np.testing.assert_array_equal(data[1], np.arange(3) * 3.0)
        assert rank == r and n == 2 and main == (r == 0)
