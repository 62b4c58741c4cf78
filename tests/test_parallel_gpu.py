"""Native model-sync workers (kaldi-aslp_amd/parallel: BspWorker / BmufWorker on a Comm).  A one-GPU box cannot host two
RCCL ranks, so the arithmetic and the protocol (sample-count all-reduce first, finished workers keep joining with 0
samples, a global count of 0 ends the run) are exercised with N ranks as threads of one process (ThreadComm) against the
closed forms of bsp-worker.cc:33-65 / bmuf-worker.cc:37-68; the RCCL path is exercised with a group of one."""
import os
import subprocess

import numpy as np
import pytest

import kaldi_formats as kf
from test_tools_gpu import minibatches, tool, write_corpus
from test_nnet_gpu import make_dnn, oracle_params

pytestmark = pytest.mark.gpu
f32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def xent_frames(err):
    """the frame count of the LAST Xent report in a tool's log"""
    import re
    return int(float(re.findall(r"AvgLoss: \S+ \(Xent\), Likelyhood: \S+ Frame: (\S+)", err)[-1]))


def run_selftest(kind, n, dim, steps, *extra):
    p = tool("aslp-parallel-selftest", kind, str(n), str(dim), str(steps), *[str(e) for e in extra])
    out = {}
    for line in p.stdout.decode().splitlines():
        f = line.split()
        out[(int(f[0]), int(f[1]))] = np.array(f[2:], f32)
    return out


def initial(n, dim):
    size = dim + dim // 2 + 1
    return [(f32(r + 1) + f32(0.01) * np.arange(size, dtype=f32)).astype(f32) for r in range(n)]


@pytest.mark.parametrize("n", [2, 3])
def test_bsp_worker_threads(n):
    dim, steps = 37, 5
    got = run_selftest("bsp", n, dim, steps)
    w = initial(n, dim)
    for s in range(steps):
        active = [r for r in range(n) if s < steps - r]
        for r in active:
            w[r] = (w[r] + f32(0.5 * (r + 1) + 0.25 * s)).astype(f32)
        cnt = [100 * (r + 1) + s if r in active else 0 for r in range(n)]
        tot = sum(cnt)
        acc = np.zeros_like(w[0])
        for r in range(n):  # rank order, like the harness' reduction
            acc = (acc + w[r] * f32(f32(cnt[r]) / f32(tot))).astype(f32) if r else (w[0] * f32(f32(cnt[0]) / f32(tot))).astype(f32)
        w = [acc.copy() for _ in range(n)]
        for r in active:
            np.testing.assert_allclose(got[(s, r)], acc, rtol=2e-6, atol=0)
    for r in range(n):  # every rank ends with the same model, the last average
        np.testing.assert_allclose(got[(steps, r)], w[r], rtol=2e-6, atol=0)


def test_bmuf_worker_threads():
    n, dim, steps, lr, mom = 3, 20, 4, 0.8, 0.6
    got = run_selftest("bmuf", n, dim, steps, lr, mom)
    w = initial(n, dim)
    # every rank starts from its own model here, so w_g(t-1) differs per rank at the first round: follow each rank
    prev = [x.copy() for x in w]
    dprev = [np.zeros_like(x) for x in w]
    for s in range(steps):
        active = [r for r in range(n) if s < steps - r]
        for r in active:
            w[r] = (w[r] + f32(0.5 * (r + 1) + 0.25 * s)).astype(f32)
        g = [(w[r] - prev[r]).astype(f32) for r in range(n)]
        G = g[0].copy()
        for r in range(1, n):
            G = (G + g[r]).astype(f32)
        coef = f32((1.0 - f32(mom)) * f32(lr))
        for r in range(n):
            d = (G * coef + f32(mom) * dprev[r]).astype(f32)
            w[r] = (prev[r] + d).astype(f32)
            prev[r], dprev[r] = w[r].copy(), d
        for r in active:
            np.testing.assert_allclose(got[(s, r)], w[r], rtol=3e-6, atol=1e-6)
    for r in range(n):
        np.testing.assert_allclose(got[(steps, r)], w[r], rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("solver", ["sgd", "momentum", "adagrad", "rmsprop", "adadelta", "adam"])
def test_sod_worker_threads(solver):
    """SodWorker (sod-worker.cc:36-68) with the six solvers of optimizer.h at their default settings, three ranks as threads:
    summed deltas w(t-1) - w(t) drive the solver on every rank's own model -- float64 numpy restatement, 1e-5."""
    n, dim, steps = 3, 20, 5
    got = run_selftest("sod:" + solver, n, dim, steps)
    w = [x.astype(np.float64) for x in initial(n, dim)]
    prev = [x.copy() for x in w]
    s1 = [np.zeros_like(x) for x in w]
    s2 = [np.zeros_like(x) for x in w]
    fl = lambda v: np.maximum(v, 1e-8)
    for s in range(steps):
        active = [r for r in range(n) if s < steps - r]
        for r in active:
            w[r] = w[r] + (0.5 * (r + 1) + 0.25 * s)
        g = sum(prev[r] - w[r] for r in range(n))
        t = s + 1
        for r in range(n):
            if solver == "sgd":
                w[r] = w[r] - 0.01 * g
            elif solver == "momentum":
                s1[r] = 0.01 * g + 0.9 * s1[r]
                w[r] = w[r] - s1[r]
            elif solver in ("adagrad", "rmsprop"):
                s1[r] = g * g + s1[r] if solver == "adagrad" else 0.1 * g * g + 0.9 * s1[r]
                w[r] = w[r] - (0.01 if solver == "adagrad" else 0.001) * g / np.sqrt(fl(s1[r]))
            elif solver == "adadelta":
                s1[r] = 0.05 * g * g + 0.95 * s1[r]
                d = g * np.sqrt(fl(s2[r])) / np.sqrt(fl(s1[r]))
                w[r] = w[r] - d
                s2[r] = 0.05 * d * d + 0.95 * s2[r]
            else:
                s1[r] = 0.1 * g + 0.9 * s1[r]
                s2[r] = 0.001 * g * g + 0.999 * s2[r]
                w[r] = w[r] - 0.001 / (1 - 0.9 ** t) * s1[r] / np.sqrt(fl(s2[r] / (1 - 0.999 ** t)))
            prev[r] = w[r].copy()
        for r in active:
            np.testing.assert_allclose(got[(s, r)], w[r], rtol=1e-5, atol=1e-6)
    for r in range(n):
        np.testing.assert_allclose(got[(steps, r)], w[r], rtol=1e-5, atol=1e-6)
        if r:  # the models are stepped, not re-unified: ranks keep their initial offsets plus what their own data added
            assert not np.allclose(got[(steps, r)], got[(steps, 0)])


@pytest.mark.parametrize("kind,a,b", [("easgd", 0.5, 0), ("easgd", 0.25, 0), ("asgd", 0.5, 0), ("masgd", 0.5, 0)])
def test_served_workers_threads(kind, a, b):
    """EasgdWorker / AsgdWorker against EasgdServer / AsgdServer / the MASGD server (easgd-*.cc, asgd-*.cc, masgd-server.cc) with
    rank 0 as the server and two workers taking turns, so the arrival order is fixed and the closed forms can be followed in
    numpy.  (The periodic barrier of the asgd server is exercised with one worker below: with strict turns a held worker
    would keep the next one from arriving.)"""
    n, dim, steps = 3, 11, 3
    got = run_selftest(kind, n, dim, steps, a, b)
    base = (f32(1) + f32(0.01) * np.arange(dim, dtype=f32)).astype(np.float64)
    server = base.copy()
    w = {r: base.copy() for r in (1, 2)}
    prev = {r: base.copy() for r in (1, 2)}
    diffs = {r: np.zeros(dim) for r in (1, 2)}
    for s in range(steps):
        for r in (1, 2):
            w[r] = w[r] + (0.5 * r + 0.25 * s)
            if kind == "easgd":
                ws, ss = w[r].copy(), server.copy()
                w[r] = (1 - a) * ws + a * ss
                server = (1 - a) * ss + a * ws
            else:
                delta = w[r] - prev[r]
                if kind == "asgd":
                    server = server + a * delta
                else:
                    diffs[r] = delta + a * diffs[r]
                    server = server + diffs[r]
                w[r] = server.copy()
                prev[r] = server.copy()
            np.testing.assert_allclose(got[(s, r)], w[r], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(got[(steps, 0)], server, rtol=2e-6, atol=1e-6)


def test_pair_sync_threads():
    """PairSync (nnet-mpi-sync.cc:75-128, inside aslp-nnet-train-simple-mpi): while both ranks train each keeps the average; the
    rank that runs out of data first follows its peer, which keeps its own model; the loop ends when both are done."""
    dim, steps = 9, 5
    got = run_selftest("pair", 2, dim, steps)
    w = [(f32(1) + f32(0.01) * np.arange(dim, dtype=f32)).astype(np.float64) for _ in range(2)]
    data = [steps, steps - 2]
    done = [False, False]
    k = 0
    while not all(done):
        for r in range(2):
            if k < data[r]:
                w[r] = w[r] + (0.5 * (r + 1) + 0.25 * k)
            else:
                done[r] = True
        a, b = w[0].copy(), w[1].copy()
        for r, (mine, peer) in enumerate(((a, b), (b, a))):
            if done[1 - r]:
                pass                       # the peer has finished: keep the own model
            elif done[r]:
                w[r] = peer.copy()         # finished: adopt the peer's
            else:
                w[r] = (mine + peer) / 2
        for r in range(2):
            np.testing.assert_allclose(got[(k, r)], w[r], rtol=2e-6, atol=1e-6)
        k += 1
    assert k == steps + 1 and (k, 0) not in got


def test_asgd_periodic_barrier_single_worker_threads():
    """asgd-server.cc:62-88 with sync_period 2 and one worker: exchanges 2, 4, ... are answered through the barrier branch
    (every running worker -- the only one -- waits), the counter drops by the period; the arithmetic is unchanged."""
    n, dim, steps = 2, 7, 5
    got = run_selftest("asgd", n, dim, steps, 0.5, 2)
    base = (f32(1) + f32(0.01) * np.arange(dim, dtype=f32)).astype(np.float64)
    server, w, prev = base.copy(), base.copy(), base.copy()
    for s in range(steps):
        w = w + (0.5 + 0.25 * s)
        server = server + 0.5 * (w - prev)
        w, prev = server.copy(), server.copy()
        np.testing.assert_allclose(got[(s, 1)], w, rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(got[(steps, 0)], server, rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("worker", ["bsp", "bmuf", "sod"])
def test_frame_worker_tool_group_of_one(aslp, oracle, dev, tmp_path, worker):
    """aslp-nnet-train-frame-worker through RCCL with one rank.  BSP with one worker is the identity (factor 1), so the
    model equals aslp-nnet-train-frame's bit for bit; BMUF with lr 1 / momentum 0 likewise (w = w_g + (w - w_g)); SOD with the
    sgd solver at --lr=0 steps by nothing (the flags, InitParam and the sync schedule are what this case covers; the solvers
    themselves are checked in test_sod_worker_threads)."""
    in_dim, hid, nh, out_dim, mb = 24, 64, 2, 40, 32
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, 1, mb, seed=21)
    _, feats, posts = write_corpus(tmp_path, np.random.default_rng(7), 12, in_dim, out_dim)
    common = ["--learn-rate=0.004", "--momentum=0.5", "--minibatch-size=%d" % mb, "--randomizer-size=150", "--randomizer-seed=9",
              "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"), str(path)]
    tool("aslp-nnet-train-frame", *common, str(tmp_path / "ref.nnet"))
    extra = ["--worker-type=%s" % worker, "--sync-period=64"]
    if worker == "bmuf":
        extra += ["--bmuf-learn-rate=1.0", "--bmuf-momentum=0.0"]
    if worker == "sod":
        extra += ["--solver=sgd", "--lr=0"]
    p = tool("aslp-nnet-train-frame-worker", *extra, *common, str(tmp_path / "w.nnet"))
    err = p.stderr.decode()
    assert "Mpi cluster info total 1 worker rank 0" in err and "All worker finished their data" in err and "AvgLoss:" in err
    a, b = aslp.Nnet.Read(tmp_path / "ref.nnet").GetParams(), aslp.Nnet.Read(tmp_path / "w.nnet").GetParams()
    # The worker is NOT aslp-nnet-train-frame with a sync in it: the reference's worker does not look at what ReadData returns
    # (aslp-nnet-train-frame-worker.cc:147 against aslp-nnet-train-frame.cc:110-111), so when the LAST cache fill holds less than one minibatch (as with this corpus) its
    # loop runs once more on the minibatch of the step before.  Oracle chain over the same minibatches with the last one taken twice:
    steps = list(minibatches(aslp, feats, posts, mb, 9, 150))
    for x, t, _ in steps + steps[-1:]:
        oracle.lib.orc_dnn_train_step(d, np.ascontiguousarray(x), np.array([fr[0][0] for fr in t], np.int32), 0.004, 0.5)
    want = oracle_params(oracle, d, 1)
    oracle.lib.orc_dnn_destroy(d)
    assert oracle.rel_err(b, want) < 1e-4
    assert oracle.rel_err(a, want) > 1e-3       # (the step more is visible: train-frame's model is not the worker's)
    assert xent_frames(err) == (len(steps) + 1) * mb


def test_train_server_tool_and_served_worker_flags(aslp, oracle, dev, tmp_path):
    """aslp-nnet-train-server alone in its group (no worker ever reports): it must come up on RCCL, read the model, find nothing
    to serve, join the statistics reduction and write the model back unchanged; a worker tool asked for a served protocol
    without a server says what it needs.  (Server + workers as separate processes need one GPU each; the protocols
    themselves run in test_served_workers_threads.)"""
    in_dim, hid, nh, out_dim, mb = 24, 64, 2, 40, 32
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, 1, mb, seed=21)
    oracle.lib.orc_dnn_destroy(d)
    for st in ("easgd", "asgd", "masgd"):
        p = tool("aslp-nnet-train-server", "--server-type=%s" % st, "--alpha=0.5", "--sync-period=10", str(path), str(tmp_path / ("s_%s.nnet" % st)))
        err = p.stderr.decode()
        assert "Mpi cluster info total 1 server rank 0" in err and "All worker finished" in err
        assert np.array_equal(aslp.Nnet.Read(tmp_path / ("s_%s.nnet" % st)).GetParams(), aslp.Nnet.Read(path).GetParams())
    p = tool("aslp-nnet-train-server", "--server-type=nope", str(path), str(tmp_path / "x.nnet"), ok=False)
    assert p.returncode != 0 and b"Unsupported server type: nope" in p.stderr
    p = tool("aslp-nnet-train-server", ok=False)
    assert p.returncode == 1 and b"Usage:  aslp-nnet-train-server [options] <model-in> <model-out>" in p.stderr
    write_corpus(tmp_path, np.random.default_rng(7), 4, in_dim, out_dim)
    p = tool("aslp-nnet-train-frame-worker", "--worker-type=easgd", "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"), str(path),
             str(tmp_path / "w.nnet"), ok=False)
    assert p.returncode != 0 and b"needs aslp-nnet-train-server as rank 0" in p.stderr
    # the two-rank averaging tool refuses any other group size, like the reference ("num of jobs must be 2")
    p = tool("aslp-nnet-train-simple-mpi", "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"), str(path), str(tmp_path / "m.nnet"),
             ok=False)
    assert p.returncode != 0 and b"num of jobs must be 2" in p.stderr


def test_lc_blstm_worker_tool_group_of_one(aslp, dev, tmp_path):
    """aslp-nnet-train-lc-blstm-streams-worker with one BSP rank equals aslp-nnet-train-blstm-streams-lc bit for bit."""
    from test_tools_gpu import LC_PROTO
    (tmp_path / "lc.proto").write_text(LC_PROTO)
    tool("aslp-nnet-init", "--seed=31", str(tmp_path / "lc.proto"), str(tmp_path / "lc.init"))
    rng = np.random.default_rng(21)
    keys = ["u%02d" % i for i in range(8)]
    lens = [int(x) for x in rng.integers(6, 30, 8)]
    feats = [rng.standard_normal((n, 12)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, 10)), 1.0)] for _ in range(n)] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    common = ["--learn-rate=0.01", "--momentum=0.9", "--num-stream=3", "--chunk-size=6"]
    io = ["ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"), str(tmp_path / "lc.init")]
    tool("aslp-nnet-train-blstm-streams-lc", *common, "--right-splice=3", *io, str(tmp_path / "a.nnet"))
    p = tool("aslp-nnet-train-lc-blstm-streams-worker", *common, "--right_splice=3", "--sync-period=20", *io, str(tmp_path / "b.nnet"))
    assert b"synchronize once" in p.stderr and b"All worker finished their data" in p.stderr
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "a.nnet").GetParams(), aslp.Nnet.Read(tmp_path / "b.nnet").GetParams())


def test_lstm_stream_worker_tool_group_of_one(aslp, dev, tmp_path):
    """aslp-nnet-train-lstm-stream-worker with one BSP rank equals aslp-nnet-train-lstm-streams bit for bit."""
    from test_tools_gpu import LSTM_PROTO
    (tmp_path / "l.proto").write_text(LSTM_PROTO)
    tool("aslp-nnet-init", "--seed=51", str(tmp_path / "l.proto"), str(tmp_path / "l.init"))
    rng = np.random.default_rng(22)
    keys = ["s%02d" % i for i in range(7)]
    lens = [int(x) for x in rng.integers(4, 20, 7)]
    feats = [rng.standard_normal((n, 12)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, 10)), 1.0)] for _ in range(n)] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    common = ["--learn-rate=0.02", "--momentum=0.9", "--num-stream=3", "--batch-size=5", "--targets-delay=2"]
    io = ["ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"), str(tmp_path / "l.init")]
    tool("aslp-nnet-train-lstm-streams", *common, *io, str(tmp_path / "a.nnet"))
    p = tool("aslp-nnet-train-lstm-stream-worker", *common, "--sync-period=12", "--verbose=2", *io, str(tmp_path / "b.nnet"))
    assert b"synchronize once" in p.stderr and b"All worker finished their data" in p.stderr
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "a.nnet").GetParams(), aslp.Nnet.Read(tmp_path / "b.nnet").GetParams())


def run_tools_together(cmds, env_extra, timeout=900):
    """start every command (name, args) as its own OS process -- the ranks of one launch, all on GPU 0 -- and wait for all of them"""
    import secrets
    from test_tools_gpu import BIN
    from test_tools_gpu import REF_MAINS
    token = secrets.token_hex(6)
    procs = []
    for r, (name, args) in enumerate(cmds):
        env = dict(os.environ, ASLP_COMM_TOKEN=token, ASLP_COMM_TRANSPORT="shm", **env_extra)
        argv = [os.path.join(BIN, name), "--rank=%d" % r, "--num-workers=%d" % len(cmds)] + list(args)
        ref_exe = os.path.join(ROOT, "kaldi-aslp_amd", "bin_ref", name)
        if REF_MAINS and os.path.exists(ref_exe):
            # the REFERENCE's own main of that name on the engine's sync layer (include/aslp_compat_kaldi_parallel.h): no --rank / --num-workers /
            # --comm-file there -- rank and size come from the launcher's environment like under mpirun, the rendezvous file from ASLP_COMM_FILE
            comm_file = [a.split("=", 1)[1] for a in args if a.startswith("--comm-file=")]
            argv = [ref_exe] + [a for a in args if not a.startswith("--comm-file=")]
            env.update(RANK=str(r), WORLD_SIZE=str(len(cmds)), ASLP_COMM_FILE=comm_file[0] if comm_file else "")
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    out = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            out.append((p.returncode, e.decode(errors="replace")))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rc, err in out:
        assert rc == 0, err[-3000:]
    return [err for _, err in out]


def test_bsp_frame_workers_as_two_processes_on_one_gpu(aslp, oracle, dev, tmp_path):
    """aslp-nnet-train-frame-worker x 2 as separate OS processes on ONE device (--comm-transport / ASLP_COMM_TRANSPORT=shm: rendezvous record,
    sample-count all-reduce, parameter all-reduce from two address spaces).  Both ranks read the SAME shard: their models stay equal,
    the BSP average n_k / N * w_k summed over the two is then w itself in fp32 (0.5 w + 0.5 w), so rank 0's model must equal the
    one-rank aslp-nnet-train-frame-worker run bit for bit -- any lost, doubled or mis-ordered contribution shows."""
    in_dim, hid, nh, out_dim, mb = 24, 64, 2, 40, 32
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, 1, mb, seed=21)
    oracle.lib.orc_dnn_destroy(d)
    write_corpus(tmp_path, np.random.default_rng(7), 12, in_dim, out_dim)
    common = ["--learn-rate=0.004", "--momentum=0.5", "--minibatch-size=%d" % mb, "--randomizer-size=150", "--randomizer-seed=9",
              "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"), str(path)]
    # (one rank alone is the yardstick -- not aslp-nnet-train-frame: the worker runs one step more at the end of its data, see
    # test_frame_worker_tool_group_of_one)
    tool("aslp-nnet-train-frame-worker", "--worker-type=bsp", "--sync-period=64", *common, str(tmp_path / "ref.nnet"))
    comm = "--comm-file=%s" % (tmp_path / "comm")
    errs = run_tools_together([("aslp-nnet-train-frame-worker", ["--worker-type=bsp", "--sync-period=64", "--gpu-id=0", comm] + common + [str(tmp_path / ("w%d.nnet" % r))])
                               for r in range(2)], {})
    for r, err in enumerate(errs):
        assert "Mpi cluster info total 2 worker rank %d" % r in err and "All worker finished their data" in err, err[-2000:]
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "ref.nnet").GetParams(), aslp.Nnet.Read(tmp_path / "w0.nnet").GetParams())
    assert not os.path.exists(tmp_path / "comm") and not os.path.exists(str(tmp_path / "comm") + ".ctl")   # the launch cleaned up after itself


@pytest.mark.parametrize("server_type", ["easgd", "asgd"])
def test_server_and_two_frame_workers_as_three_processes_on_one_gpu(aslp, oracle, dev, tmp_path, server_type):
    """aslp-nnet-train-server (rank 0) + two aslp-nnet-train-frame-worker processes, THREE OS processes on one device: the workers report
    through the control pipe, the server takes them in arrival order and exchanges whole parameter sets with each (easgd-server.cc:37-86,
    asgd-server.cc) until both have said they are done.  The arrival order is the operating system's, so the check is on the protocol:
    everybody finishes, the server's model moved away from the initial one towards the workers', all models are finite, and every worker
    ends within the elastic pull of the server."""
    in_dim, hid, nh, out_dim, mb = 24, 64, 2, 40, 32
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, 1, mb, seed=21)
    oracle.lib.orc_dnn_destroy(d)
    rng = np.random.default_rng(9)
    for r in (1, 2):
        sub = tmp_path / ("shard%d" % r)
        sub.mkdir()
        write_corpus(sub, rng, 10, in_dim, out_dim)
    comm = "--comm-file=%s" % (tmp_path / "comm")
    cmds = [("aslp-nnet-train-server", ["--server-type=%s" % server_type, "--alpha=0.5", "--sync-period=4", "--gpu-id=0", comm, str(path), str(tmp_path / "server.nnet")])]
    for r in (1, 2):
        cmds.append(("aslp-nnet-train-frame-worker", ["--worker-type=%s" % server_type, "--alpha=0.5", "--sync-period=64", "--gpu-id=0", comm, "--learn-rate=0.004",
                                                      "--minibatch-size=%d" % mb, "--randomizer-size=150", "ark:%s/shard%d/feats.ark" % (tmp_path, r),
                                                      "ark:%s/shard%d/post.ark" % (tmp_path, r), str(path), str(tmp_path / ("w%d.nnet" % r))]))
    errs = run_tools_together(cmds, {})
    assert "Mpi cluster info total 3 server rank 0" in errs[0] and "All worker finished" in errs[0], errs[0][-2000:]
    for r in (1, 2):
        assert "Mpi cluster info total 3 worker rank %d" % r in errs[r] and "AvgLoss:" in errs[r], errs[r][-2000:]
    init = aslp.Nnet.Read(path).GetParams()
    server = aslp.Nnet.Read(tmp_path / "server.nnet").GetParams()
    assert np.isfinite(server).all() and not np.array_equal(server, init)
    # workers 1 and 2 do not write a model (only the main node does, like the reference); the server's model carries both workers' training:
    # it differs from what either shard alone would give, and its loss on a shard is below the initial model's
    assert np.linalg.norm(server - init) > 0
