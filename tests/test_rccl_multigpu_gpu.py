"""The model-sync layer across REAL ranks -- separate OS processes -- on both process-per-rank transports of
kaldi-aslp_amd/parallel/comm.cpp, through libaslp_parallel.so:
  rccl   one process per GPU (ncclAllReduce / ncclSend / ncclRecv).  Needs >= 2 GPUs: on a one-GPU box these cases skip (RCCL
         refuses two ranks on one device) and the single-rank test at the bottom is what runs of it -- it still goes through
         ncclCommInitRank and ncclAllReduce, since the collectives are no longer bypassed for a group of one.
  shm    the same processes ALL ON ONE DEVICE: same rendezvous record, same control pipe and arrival order at the server, tensors
         staged through a shared-memory segment (ShmComm).  This is what exercises the multi-process machinery of the product --
         three and four OS processes sharing the one GPU of the test box -- and it runs everywhere.

Mirrors src/aslp-parallel/reduce-barrier-test.cc:15-31 (ranks issue different numbers of all-reduces, then drain) and checks
BSP / BMUF / EASGD against the closed forms of bsp-worker.cc:33-65, bmuf-worker.cc:37-68, easgd-worker.cc:37-80 +
easgd-server.cc:63-86 -- the same closed forms tests/test_parallel_gpu.py applies to the threads-as-ranks harness."""
import json
import os
import secrets
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def ngpus():
    import torch
    return torch.cuda.device_count()   # counting does not initialise the device


def run_ranks(tmp_path, n, *args, timeout=1200, transport="rccl"):
    token = secrets.token_hex(6)
    comm_file = str(tmp_path / ("comm_" + token))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r if transport == "rccl" else 0), WORLD_SIZE=str(n), ASLP_COMM_FILE=comm_file,
                   ASLP_COMM_TOKEN=token, ASLP_COMM_TRANSPORT=transport)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_rank_main.py")] + [str(a) for a in args],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    res = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            assert p.returncode == 0, "rank failed:\n" + e.decode(errors="replace")[-3000:]
            line = [l for l in o.decode().splitlines() if l.startswith("RCCL_RANK_RESULT ")][-1]
            res.append(json.loads(line[len("RCCL_RANK_RESULT "):]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return sorted(res, key=lambda d: d["rank"])


def initial(n, dim):
    size = dim + dim // 2 + 1
    return [(f32(r + 1) + f32(0.01) * np.arange(size, dtype=f32)).astype(f32) for r in range(n)]


needs2 = pytest.mark.skipif(ngpus() < 2, reason="needs >= 2 GPUs (RCCL refuses two ranks on one device)")
TRANSPORTS = [pytest.param("rccl", marks=needs2), "shm"]


def ranks_for(transport, most):
    """rccl: one per GPU, at most `most`; shm: `most` processes on the one device (capped so a small box is not swamped)"""
    return min(ngpus(), most) if transport == "rccl" else min(most, 4)


@pytest.mark.parametrize("transport", TRANSPORTS)
def test_reduce_barrier_like_the_reference(tmp_path, transport):
    n = ranks_for(transport, 8)
    res = run_ranks(tmp_path, n, "reduce-barrier", 0, 0, transport=transport)
    # round i (0-based) of the counting phase: ranks still counting contribute 1, ranks already draining contribute 0
    total = max(r + 4 for r in range(n))
    expect = [sum(1 for q in range(n) if i < q + 4) for i in range(total)]
    for r, d in enumerate(res):
        assert d["world"] == n
        seen = d["sums"] + d["drain"]
        assert seen[:total] == expect, (r, seen)
        assert seen[total:] == [0] * (len(seen) - total) and seen[-1] == 0
        assert d["dev_f32"] == n * (n + 1) / 2 and d["dev_f32_all_equal"]
        assert d["dev_f64"] == 0.5 * n * (n + 1) / 2


@pytest.mark.parametrize("transport", TRANSPORTS)
def test_bsp_closed_form_over_rccl(tmp_path, transport):
    n, dim, steps = ranks_for(transport, 4), 37, 5
    res = run_ranks(tmp_path, n, "bsp", dim, steps, transport=transport)
    w = initial(n, dim)
    for s in range(steps):
        active = [r for r in range(n) if s < steps - r]
        for r in active:
            w[r] = (w[r] + f32(0.5 * (r + 1) + 0.25 * s)).astype(f32)
        cnt = [100 * (r + 1) + s if r in active else 0 for r in range(n)]
        tot = sum(cnt)
        acc = np.zeros_like(w[0], dtype=np.float64)
        for r in range(n):
            acc += (w[r] * f32(f32(cnt[r]) / f32(tot))).astype(f32)   # ring order is RCCL's: compare at fp32 tolerance
        w = [acc.astype(f32) for _ in range(n)]
        for r in active:
            np.testing.assert_allclose(np.array(res[r]["trace"][s], f32), w[r], rtol=3e-6, atol=0)
    for r in range(n):
        np.testing.assert_allclose(np.array(res[r]["final"], f32), w[r], rtol=3e-6, atol=0)
    for r in range(1, n):   # every rank holds the SAME bits after a sum all-reduce
        assert res[r]["final"] == res[0]["final"]


@pytest.mark.parametrize("transport", TRANSPORTS)
def test_bmuf_closed_form_over_rccl(tmp_path, transport):
    n, dim, steps, lr, mom = ranks_for(transport, 3), 20, 4, 0.8, 0.6
    res = run_ranks(tmp_path, n, "bmuf", dim, steps, lr, mom, transport=transport)
    w = initial(n, dim)
    prev = [x.copy() for x in w]
    dprev = [np.zeros_like(x) for x in w]
    for s in range(steps):
        active = [r for r in range(n) if s < steps - r]
        for r in active:
            w[r] = (w[r] + f32(0.5 * (r + 1) + 0.25 * s)).astype(f32)
        g = [(w[r] - prev[r]).astype(f32) for r in range(n)]
        G = np.sum(np.array(g, np.float64), axis=0).astype(f32)
        coef = f32((1.0 - f32(mom)) * f32(lr))
        for r in range(n):
            d = (G * coef + f32(mom) * dprev[r]).astype(f32)
            w[r] = (prev[r] + d).astype(f32)
            prev[r], dprev[r] = w[r].copy(), d
        for r in active:
            np.testing.assert_allclose(np.array(res[r]["trace"][s], f32), w[r], rtol=5e-6, atol=2e-6)
    for r in range(n):
        np.testing.assert_allclose(np.array(res[r]["final"], f32), w[r], rtol=5e-6, atol=2e-6)


@pytest.mark.parametrize("transport", TRANSPORTS)
def test_easgd_server_and_workers_over_rccl(tmp_path, transport):
    """rank 0 serves, the workers report through the control pipe and are served in arrival order (easgd-server.cc:37-86); with "shm": a
    server and two workers as THREE OS processes on one device"""
    n, dim, steps, alpha = ranks_for(transport, 3), 16, 3, 0.5
    res = run_ranks(tmp_path, n, "easgd", dim, steps, alpha, transport=transport)
    w = initial(n, dim)
    server = w[0].copy()
    a = f32(alpha)
    for s in range(steps):
        for r in range(1, n):   # arrival order at the server: worker 1, 2, ..., 1, 2, ...
            w[r] = (w[r] + f32(0.5 * r + 0.25 * s)).astype(f32)
            xs, xw = server.copy(), w[r].copy()
            server = ((f32(1) - a) * xs + a * xw).astype(f32)      # easgd-server.cc:70-78
            w[r] = ((f32(1) - a) * xw + a * xs).astype(f32)        # easgd-worker.cc:59-66
            np.testing.assert_allclose(np.array(res[r]["trace"][s], f32), w[r], rtol=3e-6, atol=1e-6)
    np.testing.assert_allclose(np.array(res[0]["final"], f32), server, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("transport", TRANSPORTS)
def test_send_recv_ring_and_exchange(tmp_path, transport):
    n = ranks_for(transport, 8)
    n -= n % 2
    # 5,000,000 floats = 20 MB per message: more than a 16 MB shared-memory slot, so the chunked path runs too
    res = run_ranks(tmp_path, n, "p2p", 5000000 if transport == "shm" else 4096, 0, transport=transport)
    for r, d in enumerate(res):
        assert d["ring"] == float((r - 1) % n + 1) and d["ring_all_equal"]
        assert d["exchanged"] == float((r ^ 1) + 1)


def test_single_rank_group_runs_through_rccl(tmp_path):
    """A group of one: ncclCommInitRank + ncclAllReduce (host counts and device buffers) + the BSP worker really execute --
    the one-GPU box's share of this file."""
    res = run_ranks(tmp_path, 1, "reduce-barrier", 0, 0)
    d = res[0]
    assert d["world"] == 1 and d["sums"] == [1, 1, 1, 1] and d["drain"] == [0]
    assert d["dev_f32"] == 1.0 and d["dev_f64"] == 0.5
    res = run_ranks(tmp_path, 1, "bsp", 37, 3)
    w = initial(1, 37)[0]
    for s in range(3):
        w = (w + f32(0.5 + 0.25 * s)).astype(f32)
        np.testing.assert_allclose(np.array(res[0]["trace"][s], f32), w, rtol=1e-6)
