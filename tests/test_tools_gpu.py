"""The command-line tools (kaldi-aslp_amd/bin/aslp-nnet-*) run as the recipes run them: files
in Kaldi formats in, model / posterior tables out.  Every result is compared with (a) the same computation driven
through the engine's API on the same data order (bit-identical: the tools add I/O, not arithmetic) and (b) the CPU oracle
stepping through the same minibatches (1e-4)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import kaldi_formats as kf
import nnet_io
from test_nnet_gpu import make_dnn, oracle_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "kaldi-aslp_amd", "bin")
TOL = 1e-4


def scheduler_reads(stderr_text, pipeline):
    """What the reference's bash scheduler extracts from a tool's log: the scheduler's own pipeline (restated here as a one-line
    shell string, nothing copied to disk), fed the tool's stderr.  Returns the stripped stdout."""
    p = subprocess.run(["bash", "-c", pipeline], input=stderr_text.encode(), stdout=subprocess.PIPE, check=True)
    return p.stdout.decode().strip()


# aslp_scripts/aslp_nnet/train_scheduler.sh:87-88,120,129: loss = 4th token, loss type = 5th token of the LAST "AvgLoss:" line
SCHED_LOSS = "grep \"AvgLoss:\" | tail -n 1 | awk '{ print $4; }'"
SCHED_LOSS_TYPE = "grep \"AvgLoss:\" | tail -n 1 | awk '{ print $5; }'"
# aslp_scripts/aslp_nnet/train_scheduler_ctc.sh:90,125,134: accuracy = 11th token of the LAST "TOKEN_ACCURACY" line
SCHED_TOKEN_ACC = "grep \"TOKEN_ACCURACY\" | tail -n 1 | awk '{ print $11; }'"


REF_MAINS = os.environ.get("ASLP_TEST_REF_MAINS") == "1"


def tool(name, *args, ok=True):
    exe = os.path.join(BIN, name)
    # ASLP_TEST_REF_MAINS=1: wherever the REFERENCE's own main() of that name was built against this engine (kaldi-aslp_amd/bin_ref/, seam B4:
    # `make -C kaldi-aslp_amd refmains`), the tests drive THAT binary -- the engine's tool tests double as the reference mains' tests
    if REF_MAINS and os.path.exists(os.path.join(ROOT, "kaldi-aslp_amd", "bin_ref", name)):
        exe = os.path.join(ROOT, "kaldi-aslp_amd", "bin_ref", name)
        if name.startswith("aslp-nnet-forward"):   # (the reference's forward tools default to --use-gpu=no; this engine has no CPU compute path and says so)
            args = ("--use-gpu=yes",) + tuple(args)
    if not os.path.exists(exe):  # a tree without the built tools (they are not in git): build them, in-tree, once
        subprocess.run(["make", "-C", os.path.join(ROOT, "kaldi-aslp_amd"), "-j8"], check=True, capture_output=True, timeout=1800)
    assert os.path.exists(exe), "%s not built (make -C kaldi-aslp_amd)" % exe
    # generous: on a freshly started box the first process that maps librccl / the HIP code objects can take minutes to page in
    p = subprocess.run([exe] + list(args), capture_output=True, timeout=1800)
    if ok:
        assert p.returncode == 0, p.stderr.decode()[-3000:]
    return p


PROTO = """<NnetProto>
<AffineTransform> <InputDim> 20 <OutputDim> 48 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.1
<BatchNormalization> <InputDim> 48 <OutputDim> 48
<Sigmoid> <InputDim> 48 <OutputDim> 48
<AffineTransform> <InputDim> 48 <OutputDim> 30 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1
<Softmax> <InputDim> 30 <OutputDim> 30
</NnetProto>
"""


def test_init_copy_info(aslp, dev, tmp_path):
    (tmp_path / "nnet.proto").write_text(PROTO)
    p = tool("aslp-nnet-init", "--binary=false", "--seed=123", str(tmp_path / "nnet.proto"), str(tmp_path / "a.txt"))
    assert b"Written initialized model to" in p.stderr and b"aslp-nnet-init --binary=false --seed=123" in p.stderr
    ref = aslp.Nnet.Init(PROTO, seed=123)
    a = aslp.Nnet.Read(tmp_path / "a.txt")
    # text carries 7 significant digits; binary is exact
    tool("aslp-nnet-init", "--seed=123", str(tmp_path / "nnet.proto"), str(tmp_path / "a.bin"))
    b = aslp.Nnet.Read(tmp_path / "a.bin")
    assert np.array_equal(b.GetParams(), ref.GetParams())
    np.testing.assert_allclose(a.GetParams(), ref.GetParams(), rtol=2e-6, atol=1e-9)
    # copy: binary -> text -> binary; the text form is stable under a second round trip; pipes work as model files
    tool("aslp-nnet-copy", "--binary=false", str(tmp_path / "a.bin"), str(tmp_path / "c.txt"))
    tool("aslp-nnet-copy", "--binary=false", "cat %s |" % (tmp_path / "c.txt"), str(tmp_path / "c2.txt"))
    assert (tmp_path / "c.txt").read_bytes() == (tmp_path / "c2.txt").read_bytes()
    txt = (tmp_path / "c.txt").read_text()
    assert txt.startswith("<Nnet> \n<InputLayer>") and txt.rstrip().endswith("</Nnet>")
    tool("aslp-nnet-copy", str(tmp_path / "c.txt"), "| cat > %s" % (tmp_path / "c.bin"))
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "c.bin").GetParams(), aslp.Nnet.Read(tmp_path / "c.txt").GetParams())
    p = tool("aslp-nnet-info", str(tmp_path / "a.bin"))
    info = p.stdout.decode()
    assert "num-components 7" in info and "<AffineTransform>" in info and info == ref.Info()
    # standard nnet1 form: no Input / Output layers, no graph fields; dot file of the graph
    tool("aslp-nnet-convert-to-standard", "--binary=false", str(tmp_path / "a.bin"), str(tmp_path / "std.txt"))
    std = (tmp_path / "std.txt").read_text()
    assert "<InputLayer>" not in std and "<OutputLayer>" not in std and std.count("<AffineTransform>") == 2
    tool("aslp-nnet-dot", str(tmp_path / "a.bin"), str(tmp_path / "g.dot"))
    assert (tmp_path / "g.dot").read_text().lstrip().startswith("digraph")
    # wrong usage: usage text on stderr, exit status 1
    p = tool("aslp-nnet-info", ok=False)
    assert p.returncode == 1 and b"Usage:  aslp-nnet-info [options] <nnet-in>" in p.stderr
    p = tool("aslp-nnet-copy", str(tmp_path / "missing.nnet"), str(tmp_path / "x"), ok=False)
    assert p.returncode != 0 and b"Error opening input stream" in p.stderr


def write_corpus(tmp_path, rng, n_utt, in_dim, out_dim, soft=False):
    feats, posts, keys = [], [], []
    for i in range(n_utt):
        T = int(rng.integers(20, 60))
        feats.append(rng.standard_normal((T, in_dim)).astype(np.float32))
        if soft and i % 2:
            post = []
            for t in range(T):
                a, b = rng.choice(out_dim, 2, replace=False)
                post.append([(int(a), 0.75), (int(b), 0.25)])
        else:
            post = [[(int(rng.integers(0, out_dim)), 1.0)] for _ in range(T)]
        posts.append(post)
        keys.append("spk%02d-utt%03d" % (i % 3, i))
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    return keys, feats, posts


def minibatches(aslp, feats, posts, mb, seed, randomizer_size, shuffle=True, weights=None):
    """FrameDataReader / the train-simple loop restated: fill the cache until it is full (utterance granularity, 'full'
    = more than randomizer_size frames), shuffle with one libc-rand mask, hand out whole minibatches, carry the rest."""
    first = True
    cache_x, cache_t, cache_w = np.zeros((0, feats[0].shape[1]), np.float32), [], np.zeros(0, np.float32)
    i = 0
    while i < len(feats) or len(cache_t) >= mb:
        while i < len(feats) and not len(cache_t) > randomizer_size:
            cache_x = np.concatenate([cache_x, feats[i]])
            cache_t = cache_t + posts[i]
            cache_w = np.concatenate([cache_w, weights[i] if weights is not None else np.ones(len(posts[i]), np.float32)])
            i += 1
        if shuffle:
            mask = aslp.randomizer_mask(len(cache_t), seed if first else -1)
            first = False
            cache_x, cache_t, cache_w = cache_x[mask], [cache_t[j] for j in mask], cache_w[mask]
        b = 0
        while len(cache_t) - b >= mb:
            yield cache_x[b:b + mb], cache_t[b:b + mb], cache_w[b:b + mb]
            b += mb
        cache_x, cache_t, cache_w = cache_x[b:], cache_t[b:], cache_w[b:]
        if i >= len(feats):
            break


@pytest.mark.parametrize("bn", [0, 1])
def test_train_frame_matches_api_and_oracle(aslp, oracle, dev, tmp_path, bn):
    in_dim, hid, nh, out_dim, mb = 24, 64, 2, 40, 32
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, bn, mb, seed=21)
    rng = np.random.default_rng(7)
    keys, feats, posts = write_corpus(tmp_path, rng, 12, in_dim, out_dim)
    # drop one utterance's targets: the tool must warn and skip it
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts) if k != keys[4]]))
    lr, seed, rsize = 0.004, 99, 150
    p = tool("aslp-nnet-train-frame", "--learn-rate=%g" % lr, "--momentum=0.5", "--minibatch-size=%d" % mb, "--randomizer-size=%d" % rsize,
             "--randomizer-seed=%d" % seed, "--report-period=200", "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"),
             str(path), str(tmp_path / "out.nnet"))
    err = p.stderr.decode()
    assert "%s, missing targets" % keys[4] in err and "TRAINING STARTED" in err and "AvgLoss:" in err and "FRAME_ACCURACY >>" in err
    assert "[TRAINING, RANDOMIZED," in err and "fps" in err
    got = aslp.Nnet.Read(tmp_path / "out.nnet").GetParams()
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr, momentum=0.5)
    xent = aslp.Xent()
    f2 = [f for k, f in zip(keys, feats) if k != keys[4]]
    p2 = [q for k, q in zip(keys, posts) if k != keys[4]]
    n_mb = 0
    for x, t, _ in minibatches(aslp, f2, p2, mb, seed, rsize):
        lab = np.array([fr[0][0] for fr in t], np.int32)
        y = net.Propagate(torch.from_numpy(x).to(dev))
        diff = torch.empty_like(y)
        xent.Eval(torch.ones(mb, device=dev), y, diff, labels=torch.from_numpy(lab).to(dev))
        net.Backpropagate(diff)
        oracle.lib.orc_dnn_train_step(d, np.ascontiguousarray(x), lab, lr, 0.5)
        n_mb += 1
    assert n_mb >= 8
    assert np.array_equal(got, net.GetParams())                      # the tool adds I/O, not arithmetic
    assert oracle.rel_err(got, oracle_params(oracle, d, bn)) < TOL   # and the arithmetic is the reference's
    rep = xent.Report().splitlines()
    assert rep[0] in err and rep[1] in err
    # the scheduler's view of that log (train_scheduler.sh:120): a float that IS the average loss, and the loss type
    st = xent.GetStats()
    sched = float(scheduler_reads(err, SCHED_LOSS))
    assert abs(sched - (st["loss"] - st["entropy"]) / st["frames"]) <= 1e-4 * abs(sched)
    assert scheduler_reads(err, SCHED_LOSS_TYPE) == "(Xent),"
    # cross-validation: no model written, same log vocabulary, loss of the TRAINED model below the initial one's
    p = tool("aslp-nnet-train-frame", "--cross-validate=true", "--minibatch-size=%d" % mb, "ark:%s" % (tmp_path / "feats.ark"),
             "ark:%s" % (tmp_path / "post.ark"), str(tmp_path / "out.nnet"))
    assert b"CROSS-VALIDATION STARTED" in p.stderr and b"[CROSS-VALIDATION, RANDOMIZED" in p.stderr
    cv_loss = float(scheduler_reads(p.stderr.decode(), SCHED_LOSS))     # train_scheduler.sh:87,129 (accept / reject compares these)
    assert 0.0 < cv_loss < 20.0 and "%.4f" % cv_loss                     # printf "%.4f" of the scheduler works on it
    oracle.lib.orc_dnn_destroy(d)


def test_train_simple_weights_tolerance_and_soft_targets(aslp, oracle, dev, tmp_path):
    in_dim, hid, nh, out_dim, mb = 20, 48, 1, 30, 16
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, 0, mb, seed=4)
    oracle.lib.orc_dnn_destroy(d)
    rng = np.random.default_rng(8)
    keys, feats, posts = write_corpus(tmp_path, rng, 8, in_dim, out_dim, soft=True)
    w = [rng.uniform(0.0, 2.0, len(p)).astype(np.float32) for p in posts]
    w[2] = w[2][:-3]                       # 3 frames short: inside --length-tolerance=5 -> everything is cut to the minimum
    posts_disk = list(posts)
    posts_disk[5] = posts[5] + posts[5][:9]  # 9 frames long: dropped with a warning
    uw = {k: 1.0 for k in keys}
    uw[keys[1]] = 0.0                      # weight 0: utterance removed
    uw[keys[3]] = 0.5
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts_disk)]))
    (tmp_path / "w.ark").write_bytes(kf.archive([(k, kf.vector_bin(x)) for k, x in zip(keys, w)]))
    (tmp_path / "uw.txt").write_text("".join("%s %g\n" % (k, uw[k]) for k in keys))
    lr, seed = 0.01, 5
    args = ["--learn-rate=%g" % lr, "--minibatch-size=%d" % mb, "--randomizer-size=100", "--randomizer-seed=%d" % seed,
            "--frame-weights=ark:%s" % (tmp_path / "w.ark"), "--utt-weights=ark,t:%s" % (tmp_path / "uw.txt"),
            "scp:%s" % (tmp_path / "feats.scp"), "ark:%s" % (tmp_path / "post.ark"), str(path)]
    # features through an scp with byte offsets, made by the host-only table tool
    tool("aslp-table-copy", "ark:%s" % (tmp_path / "feats.ark"), "ark,scp:%s,%s" % (tmp_path / "f2.ark", tmp_path / "feats.scp"))
    p = tool("aslp-nnet-train-simple", *args, str(tmp_path / "out.nnet"))
    err = p.stderr.decode()
    assert "%s, length mismatch of targets" % keys[5] in err
    assert "Done 6 files, 0 with no tgt_mats, 1 with other errors. [TRAINING, RANDOMIZED" in err
    got = aslp.Nnet.Read(tmp_path / "out.nnet").GetParams()
    net = aslp.Nnet.Read(path)
    net.SetTrainOptions(learn_rate=lr)
    xent = aslp.Xent()
    use = [i for i in range(8) if i not in (1, 5)]
    f2, p2, w2 = [], [], []
    for i in use:
        n = min(len(feats[i]), len(posts[i]), len(w[i]))
        f2.append(feats[i][:n]); p2.append(posts[i][:n]); w2.append(w[i][:n] * np.float32(uw[keys[i]]))
    for x, t, fw in minibatches(aslp, f2, p2, mb, seed, 100, weights=w2):
        tgt = np.zeros((mb, out_dim), np.float32)
        for r, fr in enumerate(t):
            for c, v in fr:
                tgt[r, c] = v
        y = net.Propagate(torch.from_numpy(x).to(dev))
        diff = torch.empty_like(y)
        xent.Eval(torch.from_numpy(fw).to(dev), y, diff, targets=torch.from_numpy(tgt).to(dev))
        net.Backpropagate(diff)
    assert oracle.rel_err(got, net.GetParams()) < 1e-6   # one-hot frames take the label path in the tool, dense here
    assert xent.Report().splitlines()[1] in err
    # --randomize=false keeps the frame order; cross-validation never shuffles
    p = tool("aslp-nnet-train-simple", "--randomize=false", *args, str(tmp_path / "out2.nnet"))
    assert b"NOT-RANDOMIZED" in p.stderr
    net2 = aslp.Nnet.Read(path)
    net2.SetTrainOptions(learn_rate=lr)
    xe2 = aslp.Xent()
    for x, t, fw in minibatches(aslp, f2, p2, mb, seed, 100, shuffle=False, weights=w2):
        tgt = np.zeros((mb, out_dim), np.float32)
        for r, fr in enumerate(t):
            for c, v in fr:
                tgt[r, c] = v
        y = net2.Propagate(torch.from_numpy(x).to(dev))
        diff = torch.empty_like(y)
        xe2.Eval(torch.from_numpy(fw).to(dev), y, diff, targets=torch.from_numpy(tgt).to(dev))
        net2.Backpropagate(diff)
    assert oracle.rel_err(aslp.Nnet.Read(tmp_path / "out2.nnet").GetParams(), net2.GetParams()) < 1e-6


def test_forward_tool(aslp, oracle, dev, tmp_path):
    in_dim, hid, nh, out_dim, mb = 20, 48, 2, 30, 16
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, 0, mb, seed=6)
    oracle.lib.orc_dnn_destroy(d)
    rng = np.random.default_rng(9)
    keys, feats, _ = write_corpus(tmp_path, rng, 4, in_dim, out_dim)
    counts = rng.integers(1, 1000, out_dim)
    counts[3] = 0                                    # floored prior: the class is switched off for the decoder
    (tmp_path / "counts").write_text(" [ " + " ".join(str(c) for c in counts) + " ]\n")
    tool("aslp-nnet-forward", "--class-frame-counts=%s" % (tmp_path / "counts"), "--prior-scale=0.8", str(path),
         "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "out.ark"))
    got = kf.parse_bin_archive((tmp_path / "out.ark").read_bytes(), "matrix")
    assert [k for k, _ in got] == keys
    net = aslp.Nnet.Read(path)
    rel = counts / counts.sum()
    logp = np.log(rel + 1e-20)
    logp[rel < 1e-10] = np.sqrt(np.finfo(np.float32).max)
    for (k, o), f in zip(got, feats):
        y = net.Feedforward(torch.from_numpy(f).to(dev)).cpu().numpy()
        ref = np.log(y + np.float32(1e-20)) - np.float32(0.8) * logp.astype(np.float32)
        assert o.shape == ref.shape
        np.testing.assert_allclose(o, ref, rtol=1e-5, atol=1e-5)
        assert np.all(o[:, 3] < -1e18)
    # raw posteriors, text table to stdout
    p = tool("aslp-nnet-forward", "--apply-log=false", str(path), "ark:%s" % (tmp_path / "feats.ark"), "ark,t:-")
    first = p.stdout.decode().split("]")[0].split("[")[1].split()
    y0 = net.Feedforward(torch.from_numpy(feats[0]).to(dev)).cpu().numpy()
    np.testing.assert_allclose(np.array(first, np.float32).reshape(y0.shape), y0, rtol=2e-6, atol=1e-9)
    p = tool("aslp-nnet-forward", "--no-softmax=true", str(path), "ark:%s" % (tmp_path / "feats.ark"), "ark:/dev/null", ok=False)
    assert p.returncode != 0 and b"Cannot use both --apply-log=true --no-softmax=true" in p.stderr


LC_PROTO = """<NnetProto>
<BLstmProjectedStreamsLC> <InputDim> 12 <OutputDim> 16 <CellDim> 12 <ParamScale> 0.1 <ClipGradient> 5.0
<AffineTransform> <InputDim> 16 <OutputDim> 10 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1
<Softmax> <InputDim> 10 <OutputDim> 10
</NnetProto>
"""


def test_blstm_lc_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-blstm-streams-lc: stream bookkeeping (chunk + right context, rewind, history reset, padding,
    frame mask) restated here and driven through the API; the tool's model must come out bit-identical."""
    (tmp_path / "lc.proto").write_text(LC_PROTO)
    tool("aslp-nnet-init", "--seed=31", str(tmp_path / "lc.proto"), str(tmp_path / "lc.init"))
    rng = np.random.default_rng(11)
    n_utt, D, A, S, chunk, right = 9, 12, 10, 3, 6, 3
    keys = ["u%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(4, 26, n_utt)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, A)), 1.0)] for _ in range(n)] for n in lens]
    posts_disk = {k: p for k, p in zip(keys, posts)}
    del posts_disk[keys[2]]                                 # missing targets: skipped
    posts_disk[keys[6]] = posts[6][:-1]                     # length mismatch: skipped
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in posts_disk.items()]))
    lr, mom = 0.01, 0.9
    p = tool("aslp-nnet-train-blstm-streams-lc", "--learn-rate=%g" % lr, "--momentum=%g" % mom, "--num-stream=%d" % S, "--chunk-size=%d" % chunk,
             "--right-splice=%d" % right, "--report-period=3", "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"),
             str(tmp_path / "lc.init"), str(tmp_path / "lc.out"))
    err = p.stderr.decode()
    assert "u02, missing targets" in err and "u06, length miss-match between feats and targets, skip" in err
    got = aslp.Nnet.Read(tmp_path / "lc.out").GetParams()

    net = aslp.Nnet.Read(tmp_path / "lc.init")
    net.SetTrainOptions(learn_rate=lr, momentum=mom)
    net.SetChunkSize(chunk)
    xent = aslp.Xent()
    todo = [i for i in range(n_utt) if i not in (2, 6)]
    cur, length, which = [0] * S, [0] * S, [None] * S
    T = chunk + right
    valid_frames = steps = num_done = 0
    flags = [0] * S
    while True:
        for s in range(S):
            if cur[s] < length[s]:
                flags[s] = 0
                continue
            if todo:
                which[s] = todo.pop(0)
                cur[s], length[s], flags[s] = 0, lens[which[s]], 1
        if all(cur[s] >= length[s] for s in range(S)):
            break
        x = np.zeros((T * S, D), np.float32)
        lab = np.zeros(T * S, np.int32)
        mask = np.zeros(T * S, np.float32)
        for t in range(T):
            for s in range(S):
                r = t * S + s
                if cur[s] < length[s]:
                    mask[r] = 1.0 if t < chunk else 0.0
                    lab[r] = posts[which[s]][cur[s]][0][0]
                    x[r] = feats[which[s]][cur[s]]
                else:
                    lab[r] = posts[which[s]][length[s] - 1][0][0] if which[s] is not None else 0
                cur[s] += 1
        for s in range(S):
            cur[s] -= right
        net.ResetLstmStreams(flags)
        y = net.Propagate(torch.from_numpy(x).to(dev))
        diff = torch.empty_like(y)
        xent.Eval(torch.from_numpy(mask).to(dev), y, diff, labels=torch.from_numpy(lab).to(dev))
        net.Backpropagate(diff)
        valid_frames += int(mask.sum())
        steps += 1
        num_done += sum(flags)  # like the reference, a drained stream keeps its last flag: the count can exceed the files
    assert steps > 5
    assert "Done %d files, 1 with no tgt_mats, 1 with other errors. [TRAINING, NOT-RANDOMIZED" % num_done in err
    assert np.array_equal(got, net.GetParams())
    assert xent.GetStats()["frames"] == valid_frames
    rep = xent.Report().splitlines()
    assert rep[0] in err and rep[1] in err


CTC_PROTO = """<NnetProto>
<BLstmProjectedStreams> <InputDim> 12 <OutputDim> 16 <CellDim> 12 <ParamScale> 0.1 <ClipGradient> 5.0
<AffineTransform> <InputDim> 16 <OutputDim> 9 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1
</NnetProto>
"""


def test_warp_ctc_streams_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-warp-ctc-streams: utterance grouping (num-stream / frame-limit), padding to the longest, the
    learning rate divided by the valid frames of the group -- restated here through the API, bit-identical model."""
    (tmp_path / "ctc.proto").write_text(CTC_PROTO)
    tool("aslp-nnet-init", "--seed=41", str(tmp_path / "ctc.proto"), str(tmp_path / "ctc.init"))
    rng = np.random.default_rng(12)
    n_utt, D, A, S = 8, 12, 9, 3
    keys = ["c%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(12, 40, n_utt)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    labels = [[int(x) for x in rng.integers(1, A, max(1, n // 5))] for n in lens]
    lab_disk = {k: l for k, l in zip(keys, labels)}
    del lab_disk[keys[1]]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "lab.ark").write_bytes(kf.archive([(k, kf.int32vec_txt(l)) for k, l in lab_disk.items()]))
    lr, frame_limit = 0.5, 70
    p = tool("aslp-nnet-train-warp-ctc-streams", "--learn-rate=%g" % lr, "--momentum=0.9", "--num-stream=%d" % S, "--frame-limit=%d" % frame_limit,
             "--report-period=2", "ark:%s" % (tmp_path / "feats.ark"), "ark,t:%s" % (tmp_path / "lab.ark"), str(tmp_path / "ctc.init"),
             str(tmp_path / "ctc.out"))
    err = p.stderr.decode()
    assert "c01, missing targets" in err and "Done 7 files, 1 with no targets, 0 with other errors. [TRAINING" in err
    assert "TOKEN_ACCURACY" in err
    got = aslp.Nnet.Read(tmp_path / "ctc.out").GetParams()

    net = aslp.Nnet.Read(tmp_path / "ctc.init")
    ctc = aslp.WarpCtc()
    todo = [i for i in range(n_utt) if i != 1]
    groups = 0
    while todo:
        grp, mx = [], 0
        while todo:
            i = todo.pop(0)
            grp.append(i)
            mx = max(mx, lens[i])
            if len(grp) == S or len(grp) * mx > frame_limit:
                break
        n = len(grp)
        x = np.zeros((n * mx, D), np.float32)
        for s, i in enumerate(grp):
            x[np.arange(lens[i]) * n + s] = feats[i]
        fn = [lens[i] for i in grp]
        net.SetTrainOptions(learn_rate=np.float32(lr) / np.float32(sum(fn)), momentum=0.9)
        net.TrainStepWarpCtc(ctc, torch.from_numpy(x).to(dev), fn, [labels[i] for i in grp])
        groups += 1
    assert groups >= 3
    assert np.array_equal(got, net.GetParams())
    assert ctc.Report().strip().splitlines()[-1] in err
    # train_scheduler_ctc.sh:125: the 11th token of the last TOKEN_ACCURACY line is the accuracy in percent
    stt = ctc.GetStats()
    acc = float(scheduler_reads(err, SCHED_TOKEN_ACC))
    assert abs(acc - 100.0 * (1.0 - stt["error_tokens"] / stt["ref_tokens"])) < 1e-2


LSTM_PROTO = """<NnetProto>
<LstmProjectedStreams> <InputDim> 12 <OutputDim> 16 <CellDim> 12 <ParamScale> 0.1 <ClipGradient> 5.0
<AffineTransform> <InputDim> 16 <OutputDim> 10 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1
<Softmax> <InputDim> 10 <OutputDim> 10
</NnetProto>
"""


def test_lstm_streams_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-lstm-streams / SequenceDataReader: truncated BPTT batches with delayed targets, restated through the
    API -- including the reference's last step on the stale batch under an all-zero mask (it moves the weights when momentum
    is on) -- bit-identical model."""
    (tmp_path / "l.proto").write_text(LSTM_PROTO)
    tool("aslp-nnet-init", "--seed=51", str(tmp_path / "l.proto"), str(tmp_path / "l.init"))
    rng = np.random.default_rng(13)
    n_utt, D, A, S, B, delay = 7, 12, 10, 3, 5, 2
    keys = ["s%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(3, 19, n_utt)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, A)), 1.0)] for _ in range(n)] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    lr, mom = 0.02, 0.9
    p = tool("aslp-nnet-train-lstm-streams", "--learn-rate=%g" % lr, "--momentum=%g" % mom, "--num-stream=%d" % S, "--batch-size=%d" % B,
             "--targets-delay=%d" % delay, "--report-period=2", "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"),
             str(tmp_path / "l.init"), str(tmp_path / "l.out"))
    err = p.stderr.decode()
    got = aslp.Nnet.Read(tmp_path / "l.out").GetParams()
    net = aslp.Nnet.Read(tmp_path / "l.init")
    net.SetTrainOptions(learn_rate=lr, momentum=mom)
    xent = aslp.Xent()
    todo = list(range(n_utt))
    cur, length, which, flags = [0] * S, [0] * S, [None] * S, [0] * S
    x = lab = None
    steps = num_done = 0
    while True:
        for s in range(S):
            if cur[s] < length[s]:
                flags[s] = 0
                continue
            if todo:
                which[s] = todo.pop(0)
                cur[s], length[s], flags[s] = 0, lens[which[s]], 1
        done = all(cur[s] >= length[s] for s in range(S))
        mask = np.zeros(B * S, np.float32)
        if not done:
            x = np.zeros((B * S, D), np.float32)
            lab = np.zeros(B * S, np.int32)
            for t in range(B):
                for s in range(S):
                    r, u = t * S + s, which[s]
                    if cur[s] < length[s]:
                        mask[r] = 1.0
                        lab[r] = posts[u][cur[s]][0][0]
                    else:
                        lab[r] = posts[u][length[s] - 1][0][0]
                    x[r] = feats[u][min(cur[s] + delay, length[s] - 1)]
                    cur[s] += 1
        net.ResetLstmStreams(flags)
        y = net.Propagate(torch.from_numpy(x).to(dev))
        diff = torch.empty_like(y)
        xent.Eval(torch.from_numpy(mask).to(dev), y, diff, labels=torch.from_numpy(lab).to(dev))
        net.Backpropagate(diff)
        steps += 1
        num_done += sum(flags)
        if done and not todo:
            break
    assert steps > 6
    assert np.array_equal(got, net.GetParams())
    assert "Done %d files, [TRAINING, NOT-RANDOMIZED" % num_done in err
    if not REF_MAINS:   # (the reference's main forms the final report and drops the string, aslp-nnet-train-lstm-streams.cc:221; the engine's tool logs it)
        assert xent.Report().splitlines()[1] in err


FSMN_PROTO = """<NnetProto>
<AffineTransform> <InputDim> 12 <OutputDim> 32 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1
<CompactFsmn> <InputDim> 32 <OutputDim> 32 <PastContext> 3 <FutureContext> 2 <LearnRateCoef> 1.0
<Sigmoid> <InputDim> 32 <OutputDim> 32
<AffineTransform> <InputDim> 32 <OutputDim> 10 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1
<Softmax> <InputDim> 10 <OutputDim> 10
</NnetProto>
"""


def test_perutt_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-perutt (the FSMN recipes): one update per utterance with learn-rate / 1024."""
    (tmp_path / "f.proto").write_text(FSMN_PROTO)
    tool("aslp-nnet-init", "--seed=61", str(tmp_path / "f.proto"), str(tmp_path / "f.init"))
    rng = np.random.default_rng(14)
    n_utt, D, A = 6, 12, 10
    keys = ["p%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(8, 40, n_utt)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, A)), 1.0)] for _ in range(n)] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts) if k != keys[3]]))
    lr = 20.0
    p = tool("aslp-nnet-train-perutt", "--learn-rate=%g" % lr, "--momentum=0.5", "--drop-len=38", "ark:%s" % (tmp_path / "feats.ark"),
             "ark:%s" % (tmp_path / "post.ark"), str(tmp_path / "f.init"), str(tmp_path / "f.out"))
    err = p.stderr.decode()
    net = aslp.Nnet.Read(tmp_path / "f.init")
    net.SetTrainOptions(learn_rate=np.float32(lr) / 1024.0, momentum=0.5)
    xent = aslp.Xent()
    done = 0
    for i in range(n_utt):
        if i == 3 or lens[i] > 38:
            continue
        lab = np.array([fr[0][0] for fr in posts[i]], np.int32)
        y = net.Propagate(torch.from_numpy(feats[i]).to(dev))
        diff = torch.empty_like(y)
        xent.Eval(torch.ones(lens[i], device=dev), y, diff, labels=torch.from_numpy(lab).to(dev))
        net.Backpropagate(diff)
        done += 1
    assert "Done %d files, 1 with no tgt_mats, 0 with other errors. [TRAINING, NOT-RANDOMIZED" % done in err
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "f.out").GetParams(), net.GetParams())


def test_eesen_ctc_streams_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-ctc-streams (Eesen Ctc on posteriors; what run_ctc_*.sh call), same grouping as the Warp-CTC tool."""
    (tmp_path / "e.proto").write_text(CTC_PROTO.replace("</NnetProto>", "<Softmax> <InputDim> 9 <OutputDim> 9\n</NnetProto>"))
    tool("aslp-nnet-init", "--seed=71", str(tmp_path / "e.proto"), str(tmp_path / "e.init"))
    rng = np.random.default_rng(15)
    n_utt, D, A, S = 7, 12, 9, 3
    keys = ["e%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(12, 40, n_utt)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    labels = [[int(x) for x in rng.integers(1, A, max(1, n // 5))] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "lab.ark").write_bytes(kf.archive([(k, kf.int32vec_bin(l)) for k, l in zip(keys, labels)]))
    lr = 0.5
    p = tool("aslp-nnet-train-ctc-streams", "--learn-rate=%g" % lr, "--momentum=0.9", "--num-stream=%d" % S, "ark:%s" % (tmp_path / "feats.ark"),
             "ark:%s" % (tmp_path / "lab.ark"), str(tmp_path / "e.init"), str(tmp_path / "e.out"))
    err = p.stderr.decode()
    assert "Done 7 files, 0 with no targets, 0 with other errors. [TRAINING" in err and "TOKEN_ACCURACY" in err
    net = aslp.Nnet.Read(tmp_path / "e.init")
    ctc = aslp.Ctc()
    todo = list(range(n_utt))
    while todo:
        grp, todo = todo[:S], todo[S:]
        n, mx = len(grp), max(lens[i] for i in grp)
        x = np.zeros((n * mx, D), np.float32)
        for s, i in enumerate(grp):
            x[np.arange(lens[i]) * n + s] = feats[i]
        fn = [lens[i] for i in grp]
        net.SetTrainOptions(learn_rate=np.float32(lr) / np.float32(sum(fn)), momentum=0.9)
        net.SetSeqLengths(fn)
        y = net.Propagate(torch.from_numpy(x).to(dev))
        diff, _ = ctc.EvalParallel(fn, y, [labels[i] for i in grp])
        ctc.ErrorRateMSeq(fn, y, [labels[i] for i in grp])
        net.Backpropagate(diff)
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "e.out").GetParams(), net.GetParams())
    assert ctc.Report().strip().splitlines()[-1] in err


def test_forward_blstm_lc_tool(aslp, dev, tmp_path):
    """aslp-nnet-forward-blstm-lc: chunk + look-ahead blocks through a one-stream net, chunk part kept; the short last block
    keeps the previous block's tail rows in the input buffer, as the reference's loop does."""
    (tmp_path / "lc.proto").write_text(LC_PROTO)
    tool("aslp-nnet-init", "--seed=31", str(tmp_path / "lc.proto"), str(tmp_path / "lc.init"))
    rng = np.random.default_rng(16)
    D, A, chunk, right = 12, 10, 6, 3
    keys = ["f%d" % i for i in range(4)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in (5, 20, 31, 203)]   # (the last one has blocks of the full 80 rows)
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    p = tool("aslp-nnet-forward-blstm-lc", "--chunk-size=%d" % chunk, "--right-splice=%d" % right, "--apply-log=false", str(tmp_path / "lc.init"),
             "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "out.ark"))
    assert b"Done 4files in" in p.stderr
    got = kf.parse_bin_archive((tmp_path / "out.ark").read_bytes(), "matrix")
    net = aslp.Nnet.Read(tmp_path / "lc.init")
    net.SetChunkSize(chunk)
    B = 64 + 16   # NOT chunk + right: the reference adds the two options' DEFAULTS, in front of po.Read() (aslp-nnet-forward-blstm-lc.cc:49-58)
    for (k, o), f in zip(got, feats):
        n = len(f)
        net.ResetLstmStreams([1])
        buf = np.zeros((B, D), np.float32)
        ref = np.zeros((n, A), np.float32)
        for i in range((n - 1) // chunk + 1):
            off = i * chunk
            ln = B if off + B < n else n - off
            cp = chunk if off + chunk < n else n - off
            buf[:ln] = f[off:off + ln]
            y = net.Feedforward(torch.from_numpy(buf).to(dev)).cpu().numpy()
            ref[off:off + cp] = y[:cp]
        assert np.array_equal(o, ref), k
        assert np.allclose(o.sum(1), 1.0, atol=1e-4)


def test_train_frame_mimo_equals_two_independent_nets(aslp, dev, tmp_path):
    """aslp-nnet-train-frame-mimo on a graph net made of two independent branches (two InputLayers, two OutputLayers) must
    leave each branch exactly where aslp-nnet-train-frame leaves it when trained alone on its own tables (same seed: the
    one shuffle mask per cache fill is shared by all streams)."""
    rng = np.random.default_rng(17)
    dims = [(10, 24, 7), (14, 16, 5)]  # (in, hidden, classes) of the two branches
    W = []
    for di, h, a in dims:
        W.append((rng.standard_normal((h, di)).astype(np.float32) * 0.3, rng.standard_normal(h).astype(np.float32) * 0.1,
                  rng.standard_normal((a, h)).astype(np.float32) * 0.3, np.zeros(a, np.float32)))
    comps, cid = [], 2
    comps.append(dict(marker="<InputLayer>", dim_in=dims[0][0], dim_out=dims[0][0], id=0, inputs=[-1], offsets=[0]))
    comps.append(dict(marker="<InputLayer>", dim_in=dims[1][0], dim_out=dims[1][0], id=1, inputs=[-1], offsets=[0]))
    for b, ((di, h, a), (W1, b1, W2, b2)) in enumerate(zip(dims, W)):
        comps.append(dict(marker="<AffineTransform>", dim_in=di, dim_out=h, id=cid, inputs=[b], offsets=[0], data=nnet_io.affine(W1, b1)))
        comps.append(dict(marker="<Sigmoid>", dim_in=h, dim_out=h, id=cid + 1, inputs=[cid], offsets=[0]))
        comps.append(dict(marker="<AffineTransform>", dim_in=h, dim_out=a, id=cid + 2, inputs=[cid + 1], offsets=[0], data=nnet_io.affine(W2, b2)))
        comps.append(dict(marker="<Softmax>", dim_in=a, dim_out=a, id=cid + 3, inputs=[cid + 2], offsets=[0]))
        comps.append(dict(marker="<OutputLayer>", dim_in=a, dim_out=a, id=cid + 4, inputs=[cid + 3], offsets=[0]))
        cid += 5
        nnet_io.write_simple_nnet(tmp_path / ("b%d.nnet" % b), [("<AffineTransform>", di, h, nnet_io.affine(W1, b1)), ("<Sigmoid>", h, h, b""),
                                                                 ("<AffineTransform>", h, a, nnet_io.affine(W2, b2)), ("<Softmax>", a, a, b"")])
    nnet_io.write_graph_nnet(tmp_path / "mimo.nnet", comps)
    keys = ["m%02d" % i for i in range(9)]
    lens = [int(x) for x in rng.integers(15, 50, len(keys))]
    for b, (di, h, a) in enumerate(dims):
        feats = [rng.standard_normal((n, di)).astype(np.float32) for n in lens]
        posts = [[[(int(rng.integers(0, a)), 1.0)] for _ in range(n)] for n in lens]
        (tmp_path / ("f%d.ark" % b)).write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
        (tmp_path / ("p%d.ark" % b)).write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    opts = ["--learn-rate=0.01", "--momentum=0.5", "--minibatch-size=16", "--randomizer-size=120", "--randomizer-seed=3"]
    p = tool("aslp-nnet-train-frame-mimo", *opts, "--objective-function=xent:xent", "ark:%s" % (tmp_path / "f0.ark"), "ark:%s" % (tmp_path / "f1.ark"),
             "ark:%s" % (tmp_path / "p0.ark"), "ark:%s" % (tmp_path / "p1.ark"), str(tmp_path / "mimo.nnet"), str(tmp_path / "mimo.out"))
    assert b"Nnet num_input 2 num_output 2" in p.stderr and p.stderr.count(b"FRAME_ACCURACY") == 2
    got = aslp.Nnet.Read(tmp_path / "mimo.out").GetParams()
    parts = []
    for b in range(2):
        tool("aslp-nnet-train-frame", *opts, "ark:%s" % (tmp_path / ("f%d.ark" % b)), "ark:%s" % (tmp_path / ("p%d.ark" % b)),
             str(tmp_path / ("b%d.nnet" % b)), str(tmp_path / ("b%d.out" % b)))
        parts.append(aslp.Nnet.Read(tmp_path / ("b%d.out" % b)).GetParams())
    assert np.array_equal(got, np.concatenate(parts))
    # wrong number of tables for this net: usage, exit 1
    p = tool("aslp-nnet-train-frame-mimo", "ark:%s" % (tmp_path / "f0.ark"), "ark:%s" % (tmp_path / "p0.ark"), str(tmp_path / "mimo.nnet"), "x", ok=False)
    assert p.returncode == 1


def test_insert_tool(aslp, dev, tmp_path):
    """aslp-nnet-insert (layer-wise pretraining): hidden components of a second net go in front of the last updatable
    component, which is re-randomized with stddev-factor / sqrt(input dim) unless --randomize-next-component=false."""
    base = "<NnetProto>\n<AffineTransform> <InputDim> 10 <OutputDim> 64 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.1\n<Sigmoid> <InputDim> 64 <OutputDim> 64\n" \
           "<AffineTransform> <InputDim> 64 <OutputDim> 5 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1\n<Softmax> <InputDim> 5 <OutputDim> 5\n</NnetProto>\n"
    hid = "<NnetProto>\n<AffineTransform> <InputDim> 64 <OutputDim> 64 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.1\n<Sigmoid> <InputDim> 64 <OutputDim> 64\n</NnetProto>\n"
    (tmp_path / "base.proto").write_text(base)
    (tmp_path / "hid.proto").write_text(hid)
    tool("aslp-nnet-init", "--seed=1", str(tmp_path / "base.proto"), str(tmp_path / "base.nnet"))
    tool("aslp-nnet-init", "--seed=2", str(tmp_path / "hid.proto"), str(tmp_path / "hid.nnet"))
    b, h = aslp.Nnet.Read(tmp_path / "base.nnet"), aslp.Nnet.Read(tmp_path / "hid.nnet")
    pb, ph = b.GetParams(), h.GetParams()
    n1 = 64 * 10 + 64            # first affine of the base net
    # the second model comes through a pipe, like the recipes' "aslp-nnet-init hidden.proto - |"
    p = tool("aslp-nnet-insert", "--randomize-next-component=false", str(tmp_path / "base.nnet"), "cat %s |" % (tmp_path / "hid.nnet"), str(tmp_path / "o1.nnet"))
    assert b"Inserted 2 components at position 3" in p.stderr
    o1 = aslp.Nnet.Read(tmp_path / "o1.nnet")
    assert o1.NumComponents() == b.NumComponents() + 2
    assert [o1.Marker(i) for i in range(o1.NumComponents())] == ["<InputLayer>", "<AffineTransform>", "<Sigmoid>", "<AffineTransform>", "<Sigmoid>",
                                                                   "<AffineTransform>", "<Softmax>", "<OutputLayer>"]
    assert np.array_equal(o1.GetParams(), np.concatenate([pb[:n1], ph, pb[n1:]]))
    p = tool("aslp-nnet-insert", "--stddev-factor=0.2", "--srand=5", str(tmp_path / "base.nnet"), str(tmp_path / "hid.nnet"), str(tmp_path / "o2.nnet"))
    assert b"Randomized component index 5 with stddev 0.025" in p.stderr
    po = aslp.Nnet.Read(tmp_path / "o2.nnet").GetParams()
    assert np.array_equal(po[:n1 + ph.size], np.concatenate([pb[:n1], ph]))
    last = po[n1 + ph.size:]
    assert abs(last.std() - 0.025) < 0.004 and abs(last.mean()) < 0.004 and not np.array_equal(last, pb[n1:])
    y = aslp.Nnet.Read(tmp_path / "o2.nnet").Propagate(torch.randn(7, 10, device=dev))
    assert torch.allclose(y.sum(1), torch.ones(7, device=dev), atol=1e-5)


def test_forward_skip_tool(aslp, dev, tmp_path):
    """aslp-nnet-forward-skip: skip-width interleaved sub-sequences, each through the (recurrent) net on its own."""
    (tmp_path / "l.proto").write_text(LSTM_PROTO)
    tool("aslp-nnet-init", "--seed=51", str(tmp_path / "l.proto"), str(tmp_path / "l.init"))
    rng = np.random.default_rng(19)
    feats = [rng.standard_normal((n, 12)).astype(np.float32) for n in (17, 4, 9)]
    keys = ["k%d" % i for i in range(3)]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    w = 3
    tool("aslp-nnet-forward-skip", "--skip-width=%d" % w, "--apply-log=false", str(tmp_path / "l.init"), "ark:%s" % (tmp_path / "feats.ark"),
         "ark:%s" % (tmp_path / "out.ark"))
    got = kf.parse_bin_archive((tmp_path / "out.ark").read_bytes(), "matrix")
    net = aslp.Nnet.Read(tmp_path / "l.init")
    for (k, o), f in zip(got, feats):
        ref = np.zeros((len(f), 10), np.float32)
        for off in range(min(w, len(f))):
            sub = np.ascontiguousarray(f[off::w])
            net.SetSeqLengths([len(sub)])
            ref[off::w] = net.Feedforward(torch.from_numpy(sub).to(dev)).cpu().numpy()
        assert np.array_equal(o, ref), k
    if not REF_MAINS:   # (with its default of 0 the reference's main runs no pass at all and writes empty matrices; the engine's tool refuses)
        p = tool("aslp-nnet-forward-skip", str(tmp_path / "l.init"), "ark:%s" % (tmp_path / "feats.ark"), "ark:/dev/null", ok=False)
        assert p.returncode != 0 and b"--skip-width must be at least 1" in p.stderr


BLSTM_PROTO = """<NnetProto>
<BLstmProjectedStreams> <InputDim> 12 <OutputDim> 16 <CellDim> 12 <ParamScale> 0.1 <ClipGradient> 5.0
<AffineTransform> <InputDim> 16 <OutputDim> 10 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1
<Softmax> <InputDim> 10 <OutputDim> 10
</NnetProto>
"""


def test_blstm_streams_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-blstm-streams: whole utterances, groups of num-stream padded to the longest (zero features, empty
    targets, weight 0), SetSeqLengths, learning rate / valid frames -- restated through the API, bit-identical model."""
    (tmp_path / "b.proto").write_text(BLSTM_PROTO)
    tool("aslp-nnet-init", "--seed=81", str(tmp_path / "b.proto"), str(tmp_path / "b.init"))
    rng = np.random.default_rng(23)
    n_utt, D, A, S = 7, 12, 10, 3
    keys = ["w%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(5, 30, n_utt)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, A)), 1.0)] for _ in range(n)] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    lr = 0.4
    p = tool("aslp-nnet-train-blstm-streams", "--learn-rate=%g" % lr, "--momentum=0.9", "--num-stream=%d" % S, "ark:%s" % (tmp_path / "feats.ark"),
             "ark:%s" % (tmp_path / "post.ark"), str(tmp_path / "b.init"), str(tmp_path / "b.out"))
    assert re.search(rb"Done 7 files, 0 with no tgt_mats, 0 with other errors\. \[TRAINING, [0-9.e+-]+ min, fps", p.stderr), p.stderr.decode()[-2000:]
    net = aslp.Nnet.Read(tmp_path / "b.init")
    xent = aslp.Xent()
    todo = list(range(n_utt))
    while todo:
        grp, todo = todo[:S], todo[S:]
        n, mx = len(grp), max(lens[i] for i in grp)
        x = np.zeros((n * mx, D), np.float32)
        tgt = np.zeros((n * mx, A), np.float32)
        wgt = np.zeros(n * mx, np.float32)
        for s, i in enumerate(grp):
            rows = np.arange(lens[i]) * n + s
            x[rows] = feats[i]
            tgt[rows, [fr[0][0] for fr in posts[i]]] = 1.0
            wgt[rows] = 1.0
        net.SetSeqLengths([lens[i] for i in grp])
        net.SetTrainOptions(learn_rate=np.float32(lr) / np.float32(sum(lens[i] for i in grp)), momentum=0.9)
        y = net.Propagate(torch.from_numpy(x).to(dev))
        diff = torch.empty_like(y)
        xent.Eval(torch.from_numpy(wgt).to(dev), y, diff, targets=torch.from_numpy(tgt).to(dev))
        net.Backpropagate(diff)
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "b.out").GetParams(), net.GetParams())
    assert xent.Report().splitlines()[1] in p.stderr.decode()


def test_feature_transform_option(aslp, oracle, dev, tmp_path):
    """--feature-transform (a Splice net in front of the model, as the recipes build it): aslp-nnet-train-simple and
    aslp-nnet-forward with the transform equal the same tools fed with features spliced beforehand."""
    D, ctx = 8, 2
    tr = "<NnetProto>\n<Splice> <InputDim> %d <OutputDim> %d <BuildVector> -%d:%d </BuildVector>\n</NnetProto>\n" % (D, D * (2 * ctx + 1), ctx, ctx)
    (tmp_path / "tr.proto").write_text(tr)
    tool("aslp-nnet-init", str(tmp_path / "tr.proto"), str(tmp_path / "tr.nnet"))
    in_dim = D * (2 * ctx + 1)
    d, path = make_dnn(oracle, tmp_path, in_dim, 32, 1, 12, 0, 16, seed=3)
    oracle.lib.orc_dnn_destroy(d)
    rng = np.random.default_rng(31)
    keys = ["t%d" % i for i in range(5)]
    lens = [int(x) for x in rng.integers(20, 40, 5)]
    raw = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, 12)), 1.0)] for _ in range(n)] for n in lens]
    spliced = [np.concatenate([f[np.clip(np.arange(len(f)) + o, 0, len(f) - 1)] for o in range(-ctx, ctx + 1)], 1) for f in raw]
    (tmp_path / "raw.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, raw)]))
    (tmp_path / "spl.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, spliced)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    opts = ["--learn-rate=0.01", "--minibatch-size=16", "--randomizer-size=60", "--randomizer-seed=4"]
    tool("aslp-nnet-train-simple", *opts, "--feature-transform=%s" % (tmp_path / "tr.nnet"), "ark:%s" % (tmp_path / "raw.ark"),
         "ark:%s" % (tmp_path / "post.ark"), str(path), str(tmp_path / "a.nnet"))
    tool("aslp-nnet-train-simple", *opts, "ark:%s" % (tmp_path / "spl.ark"), "ark:%s" % (tmp_path / "post.ark"), str(path), str(tmp_path / "b.nnet"))
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "a.nnet").GetParams(), aslp.Nnet.Read(tmp_path / "b.nnet").GetParams())
    tool("aslp-nnet-forward", "--feature-transform=%s" % (tmp_path / "tr.nnet"), str(tmp_path / "a.nnet"), "ark:%s" % (tmp_path / "raw.ark"), "ark:%s" % (tmp_path / "fa.ark"))
    tool("aslp-nnet-forward", str(tmp_path / "a.nnet"), "ark:%s" % (tmp_path / "spl.ark"), "ark:%s" % (tmp_path / "fb.ark"))
    assert (tmp_path / "fa.ark").read_bytes() == (tmp_path / "fb.ark").read_bytes()


def test_train_mse_tool_matches_float64_sgd(aslp, oracle, dev, tmp_path):
    """aslp-nnet-train-mse: matrix targets read in step with the features, Mse diff = (y - t) * w, momentum SGD.  Checked
    against float64 numpy stepping through the same (unshuffled) minibatches; an utterance whose target has another
    number of rows is left out; tables in a different order are an error."""
    rng = np.random.default_rng(31)
    D, H, O, mb = 10, 24, 6, 16
    W1, b1 = rng.standard_normal((H, D)).astype(np.float32) * 0.3, rng.standard_normal(H).astype(np.float32) * 0.1
    W2, b2 = rng.standard_normal((O, H)).astype(np.float32) * 0.3, np.zeros(O, np.float32)
    nnet_io.write_simple_nnet(tmp_path / "r.nnet", [("<AffineTransform>", D, H, nnet_io.affine(W1, b1)), ("<Sigmoid>", H, H, b""),
                                                     ("<AffineTransform>", H, O, nnet_io.affine(W2, b2))])
    keys = ["r%02d" % i for i in range(8)]
    lens = [int(x) for x in rng.integers(10, 40, len(keys))]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    tgts = [rng.standard_normal((n, O)).astype(np.float32) for n in lens]
    tgts[2] = tgts[2][:-1]  # row count differs: warned about and left out
    (tmp_path / "f.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "t.ark").write_bytes(kf.archive([(k, kf.matrix_bin(t)) for k, t in zip(keys, tgts)]))
    lr, mom = 0.002, 0.5
    p = tool("aslp-nnet-train-mse", "--learn-rate=%g" % lr, "--momentum=%g" % mom, "--minibatch-size=%d" % mb, "--randomize=false",
             "--report-period=64", "ark:%s" % (tmp_path / "f.ark"), "ark:%s" % (tmp_path / "t.ark"), str(tmp_path / "r.nnet"), str(tmp_path / "r.out"))
    err = p.stderr.decode()
    assert "%s feat and target are not the same dim feat %d target %d" % (keys[2], lens[2], lens[2] - 1) in err
    assert "Done 7 files, 0 with no tgt_mats, 0 with other errors. [TRAINING, NOT-RANDOMIZED" in err and "AvgLoss:" in err
    X = np.concatenate([f for i, f in enumerate(feats) if i != 2]).astype(np.float64)
    Tg = np.concatenate([t for i, t in enumerate(tgts) if i != 2]).astype(np.float64)
    P = [a.astype(np.float64) for a in (W1, b1, W2, b2)]
    V = [np.zeros_like(a) for a in P]
    for s in range(0, len(X) - mb + 1, mb):
        x, t = X[s:s + mb], Tg[s:s + mb]
        h = 1.0 / (1.0 + np.exp(-(x @ P[0].T + P[1])))
        y = h @ P[2].T + P[3]
        d2 = y - t
        d1 = (d2 @ P[2]) * h * (1.0 - h)
        G = [d1.T @ x, d1.sum(0), d2.T @ h, d2.sum(0)]
        for k in range(4):
            V[k] = G[k] + mom * V[k]
            P[k] = P[k] - lr * V[k]
    got = aslp.Nnet.Read(tmp_path / "r.out").GetParams()
    assert oracle.rel_err(got, np.concatenate([a.ravel() for a in P])) < TOL
    # cross-validation on the trained model: no update, loss reported
    p = tool("aslp-nnet-train-mse", "--cross-validate=true", "--minibatch-size=%d" % mb, "ark:%s" % (tmp_path / "f.ark"), "ark:%s" % (tmp_path / "t.ark"),
             str(tmp_path / "r.out"))
    assert b"[CROSS-VALIDATION, RANDOMIZED" in p.stderr and b"AvgLoss:" in p.stderr
    # the two tables must list the same keys in the same order
    (tmp_path / "t2.ark").write_bytes(kf.archive([(k, kf.matrix_bin(t)) for k, t in list(zip(keys, tgts))[::-1]]))
    p = tool("aslp-nnet-train-mse", "ark:%s" % (tmp_path / "f.ark"), "ark:%s" % (tmp_path / "t2.ark"), str(tmp_path / "r.nnet"), str(tmp_path / "x.out"), ok=False)
    assert p.returncode != 0 and b"feat and target not in the same order" in p.stderr


def test_train_ctc_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-ctc: one utterance per update with the Eesen objective; missing targets and --drop-len leave utterances
    out; --token-symbol-table prints every greedy hypothesis in token names."""
    # a unidirectional net: with no SetSeqLengths / ResetLstmStreams call the LSTM runs as one stream, history carried over
    (tmp_path / "e.proto").write_text(LSTM_PROTO.replace("<OutputDim> 10", "<OutputDim> 9").replace("<InputDim> 10", "<InputDim> 9"))
    tool("aslp-nnet-init", "--seed=72", str(tmp_path / "e.proto"), str(tmp_path / "e.init"))
    rng = np.random.default_rng(16)
    n_utt, D, A = 6, 12, 9
    keys = ["c%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(12, 40, n_utt)]
    lens[4] = 55
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    labels = [[int(x) for x in rng.integers(1, A, max(1, n // 5))] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "lab.ark").write_bytes(kf.archive([(k, kf.int32vec_bin(l)) for k, l in zip(keys, labels) if k != keys[1]]))
    (tmp_path / "tokens.txt").write_text("".join("tok%d %d\n" % (i, i) for i in range(A)))
    lr = 0.01
    p = tool("aslp-nnet-train-ctc", "--learn-rate=%g" % lr, "--momentum=0.9", "--drop-len=50", "--token-symbol-table=%s" % (tmp_path / "tokens.txt"),
             "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "lab.ark"), str(tmp_path / "e.init"), str(tmp_path / "e.out"))
    err = p.stderr.decode()
    assert "%s, missing targets" % keys[1] in err and "%s, too long, droped" % keys[4] in err
    assert "Done 4 files, 1 with no targets, 0 with other errors. [TRAINING" in err and "TOKEN_ACCURACY" in err
    net = aslp.Nnet.Read(tmp_path / "e.init")
    net.SetTrainOptions(learn_rate=lr, momentum=0.9)
    ctc = aslp.Ctc()
    for i in range(n_utt):
        if i in (1, 4):
            continue
        y = net.Propagate(torch.from_numpy(feats[i]).to(dev))
        diff = ctc.Eval(y, labels[i])
        diff = diff[0] if isinstance(diff, tuple) else diff
        ctc.ErrorRate(y, labels[i])
        best = y.argmax(1).cpu().numpy()  # greedy path: collapse repeats, drop blanks (ctc-loss.cc:229-262)
        hyp = [int(a) for k, a in enumerate(best) if a != 0 and (k == 0 or a != best[k - 1])]
        net.Backpropagate(diff)
        line = [ln for ln in err.splitlines() if ln.startswith(keys[i] + " ")]
        assert len(line) == 1 and line[0].split()[1:] == ["tok%d" % h for h in hyp]
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "e.out").GetParams(), net.GetParams())
    assert ctc.Report().strip().splitlines()[-1] in err


def test_lstm_streams_skip_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-lstm-streams-skip: skip-width passes over the data, pass k on frames k, k + skip-width, ... of every
    utterance; each pass ends as soon as all streams are exhausted (no extra step on a stale batch, unlike the -streams tool)."""
    (tmp_path / "l.proto").write_text(LSTM_PROTO)
    tool("aslp-nnet-init", "--seed=52", str(tmp_path / "l.proto"), str(tmp_path / "l.init"))
    rng = np.random.default_rng(33)
    n_utt, D, A, S, B, delay, W = 6, 12, 10, 2, 4, 1, 3
    keys = ["k%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(4, 25, n_utt)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, A)), 1.0)] for _ in range(n)] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    lr, mom = 0.02, 0.9
    p = tool("aslp-nnet-train-lstm-streams-skip", "--learn-rate=%g" % lr, "--momentum=%g" % mom, "--num-stream=%d" % S, "--batch-size=%d" % B,
             "--targets-delay=%d" % delay, "--skip-width=%d" % W, "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"),
             str(tmp_path / "l.init"), str(tmp_path / "l.out"))
    err = p.stderr.decode()
    net = aslp.Nnet.Read(tmp_path / "l.init")
    net.SetTrainOptions(learn_rate=lr, momentum=mom)
    xent = aslp.Xent()
    steps = num_done = 0
    for off in range(W):
        sub_f = [f[off::W] for f in feats]
        sub_p = [q[off::W] for q in posts]
        sub_n = [len(q) for q in sub_p]
        todo = list(range(n_utt))
        cur, length, which, flags = [0] * S, [0] * S, [None] * S, [0] * S
        while True:
            for s in range(S):
                if cur[s] < length[s]:
                    flags[s] = 0
                    continue
                if todo:
                    which[s] = todo.pop(0)
                    cur[s], length[s], flags[s] = 0, sub_n[which[s]], 1
            if all(cur[s] >= length[s] for s in range(S)):
                break
            x = np.zeros((B * S, D), np.float32)
            lab = np.zeros(B * S, np.int32)
            mask = np.zeros(B * S, np.float32)
            for t in range(B):
                for s in range(S):
                    r, u = t * S + s, which[s]
                    if cur[s] < length[s]:
                        mask[r] = 1.0
                        lab[r] = sub_p[u][cur[s]][0][0]
                    else:
                        lab[r] = sub_p[u][length[s] - 1][0][0]
                    x[r] = sub_f[u][min(cur[s] + delay, length[s] - 1)]
                    cur[s] += 1
            net.ResetLstmStreams(flags)
            y = net.Propagate(torch.from_numpy(x).to(dev))
            diff = torch.empty_like(y)
            xent.Eval(torch.from_numpy(mask).to(dev), y, diff, labels=torch.from_numpy(lab).to(dev))
            net.Backpropagate(diff)
            steps += 1
            num_done += sum(flags)
    assert steps > 8 and num_done == W * n_utt
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "l.out").GetParams(), net.GetParams())
    assert "Done %d files, 0 with no tgt_mats, 0 with other errors. [TRAINING, NOT-RANDOMIZED" % num_done in err
    assert xent.Report().splitlines()[1] in err


def test_blstm_parallel_tool_matches_api(aslp, dev, tmp_path):
    """aslp-nnet-train-blstm-parallel: groups of num-stream whole utterances, NO frame weights (padded frames count, with an
    empty target), fixed learning rate."""
    (tmp_path / "b.proto").write_text(BLSTM_PROTO)
    tool("aslp-nnet-init", "--seed=82", str(tmp_path / "b.proto"), str(tmp_path / "b.init"))
    rng = np.random.default_rng(24)
    n_utt, D, A, S = 7, 12, 10, 3
    keys = ["q%02d" % i for i in range(n_utt)]
    lens = [int(x) for x in rng.integers(5, 30, n_utt)]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    posts = [[[(int(rng.integers(0, A)), 1.0)] for _ in range(n)] for n in lens]
    (tmp_path / "feats.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "post.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    lr = 0.001
    p = tool("aslp-nnet-train-blstm-parallel", "--learn-rate=%g" % lr, "--momentum=0.9", "--num-stream=%d" % S, "ark:%s" % (tmp_path / "feats.ark"),
             "ark:%s" % (tmp_path / "post.ark"), str(tmp_path / "b.init"), str(tmp_path / "b.out"))
    assert b"Done 7 files, 0 with no tgt_mats, 0 with other errors. [TRAINING, " in p.stderr
    net = aslp.Nnet.Read(tmp_path / "b.init")
    net.SetTrainOptions(learn_rate=lr, momentum=0.9)
    xent = aslp.Xent()
    todo = list(range(n_utt))
    while todo:
        grp, todo = todo[:S], todo[S:]
        n, mx = len(grp), max(lens[i] for i in grp)
        x = np.zeros((n * mx, D), np.float32)
        tgt = np.zeros((n * mx, A), np.float32)
        for s, i in enumerate(grp):
            rows = np.arange(lens[i]) * n + s
            x[rows] = feats[i]
            tgt[rows, [fr[0][0] for fr in posts[i]]] = 1.0
        net.SetSeqLengths([lens[i] for i in grp])
        y = net.Propagate(torch.from_numpy(x).to(dev))
        diff = torch.empty_like(y)
        xent.Eval(torch.ones(n * mx, device=dev), y, diff, targets=torch.from_numpy(tgt).to(dev))
        net.Backpropagate(diff)
    assert np.array_equal(aslp.Nnet.Read(tmp_path / "b.out").GetParams(), net.GetParams())
    assert xent.Report().splitlines()[1] in p.stderr.decode()


def test_forward_mimo_tool(aslp, dev, tmp_path):
    """aslp-nnet-forward-mimo on a two-input / two-output graph net: the LAST output is written, equal to what aslp-nnet-forward
    writes for that branch alone; tables whose keys disagree are an error."""
    rng = np.random.default_rng(41)
    dims = [(10, 24, 7), (14, 16, 5)]
    comps, cid = [], 2
    comps.append(dict(marker="<InputLayer>", dim_in=dims[0][0], dim_out=dims[0][0], id=0, inputs=[-1], offsets=[0]))
    comps.append(dict(marker="<InputLayer>", dim_in=dims[1][0], dim_out=dims[1][0], id=1, inputs=[-1], offsets=[0]))
    for b, (di, h, a) in enumerate(dims):
        W1, b1 = rng.standard_normal((h, di)).astype(np.float32) * 0.3, rng.standard_normal(h).astype(np.float32) * 0.1
        W2, b2 = rng.standard_normal((a, h)).astype(np.float32) * 0.3, rng.standard_normal(a).astype(np.float32) * 0.1
        comps.append(dict(marker="<AffineTransform>", dim_in=di, dim_out=h, id=cid, inputs=[b], offsets=[0], data=nnet_io.affine(W1, b1)))
        comps.append(dict(marker="<Sigmoid>", dim_in=h, dim_out=h, id=cid + 1, inputs=[cid], offsets=[0]))
        comps.append(dict(marker="<AffineTransform>", dim_in=h, dim_out=a, id=cid + 2, inputs=[cid + 1], offsets=[0], data=nnet_io.affine(W2, b2)))
        comps.append(dict(marker="<Softmax>", dim_in=a, dim_out=a, id=cid + 3, inputs=[cid + 2], offsets=[0]))
        comps.append(dict(marker="<OutputLayer>", dim_in=a, dim_out=a, id=cid + 4, inputs=[cid + 3], offsets=[0]))
        cid += 5
        nnet_io.write_simple_nnet(tmp_path / ("b%d.nnet" % b), [("<AffineTransform>", di, h, nnet_io.affine(W1, b1)), ("<Sigmoid>", h, h, b""),
                                                                 ("<AffineTransform>", h, a, nnet_io.affine(W2, b2)), ("<Softmax>", a, a, b"")])
    nnet_io.write_graph_nnet(tmp_path / "mimo.nnet", comps)
    keys = ["m%02d" % i for i in range(5)]
    lens = [int(x) for x in rng.integers(5, 30, len(keys))]
    for b, (di, h, a) in enumerate(dims):
        feats = [rng.standard_normal((n, di)).astype(np.float32) for n in lens]
        (tmp_path / ("f%d.ark" % b)).write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    p = tool("aslp-nnet-forward-mimo", "--apply-log=true", str(tmp_path / "mimo.nnet"), "ark:%s" % (tmp_path / "f0.ark"), "ark:%s" % (tmp_path / "f1.ark"),
             "ark:%s" % (tmp_path / "out.ark"))
    assert b"Nnet num_input 2 num_output 2" in p.stderr and b"Done 5 files" in p.stderr
    tool("aslp-nnet-forward", "--apply-log=true", str(tmp_path / "b1.nnet"), "ark:%s" % (tmp_path / "f1.ark"), "ark:%s" % (tmp_path / "single.ark"))
    assert (tmp_path / "out.ark").read_bytes() == (tmp_path / "single.ark").read_bytes()
    # key order differs between the tables
    ent = kf.parse_bin_archive((tmp_path / "f1.ark").read_bytes(), "matrix")
    (tmp_path / "f1r.ark").write_bytes(kf.archive([(k, kf.matrix_bin(m)) for k, m in ent[::-1]]))
    p = tool("aslp-nnet-forward-mimo", str(tmp_path / "mimo.nnet"), "ark:%s" % (tmp_path / "f0.ark"), "ark:%s" % (tmp_path / "f1r.ark"),
             "ark:%s" % (tmp_path / "x.ark"), ok=False)
    assert p.returncode != 0 and b"Different key from the features" in p.stderr
    p = tool("aslp-nnet-forward-mimo", str(tmp_path / "mimo.nnet"), "ark:%s" % (tmp_path / "f0.ark"), "ark:%s" % (tmp_path / "x.ark"), ok=False)
    assert p.returncode == 1


def test_reference_mains_compiled_unchanged_run_on_this_engine(aslp, oracle, dev, tmp_path):
    """Seam B4 as a source-level drop-in: kaldi-aslp_amd/bin_ref/* are the REFERENCE's own aslp-nnetbin/*.cc, compiled unchanged against
    include/aslp_compat_kaldi.h and linked with this engine (`make -C kaldi-aslp_amd refmains`, development container; the binaries
    travel, the sources do not).  They must do what the engine's own tools do: same model files, same log vocabulary -- and, for the
    training tool, the reference's arithmetic (CPU oracle over the same minibatches)."""
    ref_bin = os.path.join(ROOT, "kaldi-aslp_amd", "bin_ref")
    if not os.path.exists(os.path.join(ref_bin, "aslp-nnet-train-frame")):
        pytest.skip("bin_ref/ not built (needs the reference tree: make -C kaldi-aslp_amd refmains)")

    def ref_tool(name, *args):
        p = subprocess.run([os.path.join(ref_bin, name)] + list(args), capture_output=True, timeout=1800)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        return p

    # init / copy / info: byte-identical files and text
    (tmp_path / "nnet.proto").write_text(PROTO)
    ref_tool("aslp-nnet-init", "--seed=123", str(tmp_path / "nnet.proto"), str(tmp_path / "r.bin"))
    tool("aslp-nnet-init", "--seed=123", str(tmp_path / "nnet.proto"), str(tmp_path / "o.bin"))
    assert (tmp_path / "r.bin").read_bytes() == (tmp_path / "o.bin").read_bytes()
    ref_tool("aslp-nnet-copy", "--binary=false", str(tmp_path / "r.bin"), str(tmp_path / "r.txt"))
    tool("aslp-nnet-copy", "--binary=false", str(tmp_path / "o.bin"), str(tmp_path / "o.txt"))
    assert (tmp_path / "r.txt").read_bytes() == (tmp_path / "o.txt").read_bytes()
    assert ref_tool("aslp-nnet-info", str(tmp_path / "r.bin")).stdout == tool("aslp-nnet-info", str(tmp_path / "o.bin")).stdout

    # train-frame: the reference's loop body (Propagate, LossItf::Eval, Backpropagate: aslp-nnet-train-frame.cc:109-131) against the engine's
    # tool (same steps in the executor's own buffers) and against the oracle
    in_dim, hid, nh, out_dim, mb = 24, 64, 2, 40, 32
    d, path = make_dnn(oracle, tmp_path, in_dim, hid, nh, out_dim, 1, mb, seed=33)
    rng = np.random.default_rng(17)
    keys, feats, posts = write_corpus(tmp_path, rng, 10, in_dim, out_dim)
    lr, seed, rsize = 0.004, 41, 150
    args = ["--learn-rate=%g" % lr, "--momentum=0.5", "--minibatch-size=%d" % mb, "--randomizer-size=%d" % rsize, "--randomizer-seed=%d" % seed,
            "ark:%s" % (tmp_path / "feats.ark"), "ark:%s" % (tmp_path / "post.ark"), str(path)]
    pr = ref_tool("aslp-nnet-train-frame", *args, str(tmp_path / "ref.nnet"))
    po = tool("aslp-nnet-train-frame", *args, str(tmp_path / "own.nnet"))
    got_ref = aslp.Nnet.Read(tmp_path / "ref.nnet").GetParams()
    got_own = aslp.Nnet.Read(tmp_path / "own.nnet").GetParams()
    assert oracle.rel_err(got_ref, got_own) < 1e-6      # (the tool leaves the final Softmax to the loss kernel; same bits or a rounding apart)
    for x, t, _ in minibatches(aslp, feats, posts, mb, seed, rsize):
        oracle.lib.orc_dnn_train_step(d, np.ascontiguousarray(x), np.array([fr[0][0] for fr in t], np.int32), lr, 0.5)
    assert oracle.rel_err(got_ref, oracle_params(oracle, d, 1)) < TOL
    oracle.lib.orc_dnn_destroy(d)
    er, eo = pr.stderr.decode(), po.stderr.decode()
    for word in ("TRAINING STARTED", "[TRAINING, RANDOMIZED,", "AvgLoss:", "FRAME_ACCURACY >>"):
        assert word in er and word in eo
    assert abs(float(scheduler_reads(er, SCHED_LOSS)) - float(scheduler_reads(eo, SCHED_LOSS))) < 1e-4
    # cross-validation through the reference's main
    pc = ref_tool("aslp-nnet-train-frame", "--cross-validate=true", "--minibatch-size=%d" % mb, "ark:%s" % (tmp_path / "feats.ark"),
                  "ark:%s" % (tmp_path / "post.ark"), str(tmp_path / "ref.nnet"))
    assert b"CROSS-VALIDATION STARTED" in pc.stderr and b"AvgLoss:" in pc.stderr


@pytest.mark.skipif(REF_MAINS, reason="this IS the inner run")
def test_every_reference_main_passes_the_tool_tests_of_its_name():
    """All 23 mains of the reference's aslp-nnetbin/ are built unchanged against the engine (kaldi-aslp_amd/bin_ref/).  With
    ASLP_TEST_REF_MAINS=1 the `tool()` helper of this file drives THOSE binaries, so every tool test here (and the MultiTaskLoss one) --
    bit-identical models against the API restatement of each tool's loop, the oracle chains, the log vocabulary the schedulers read --
    is run a second time on the reference's own main() functions.  The three places where the engine's tools differ on purpose are
    spelled out where they are skipped (REF_MAINS): the final report aslp-nnet-train-lstm-streams forms and drops, --skip-width=0, and
    the forward tools' --use-gpu default (the reference's is "no"; this engine has no CPU compute path and says so)."""
    ref_bin = os.path.join(ROOT, "kaldi-aslp_amd", "bin_ref")
    if len([f for f in os.listdir(ref_bin)] if os.path.isdir(ref_bin) else []) < 23:
        pytest.skip("bin_ref/ not built (needs the reference tree: make -C kaldi-aslp_amd refmains)")
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_tools_gpu.py"), os.path.join(here, "test_multitask_gpu.py"), "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=dict(os.environ, ASLP_TEST_REF_MAINS="1"), capture_output=True, timeout=3000)
    tail = p.stdout.decode()[-3000:]
    assert p.returncode == 0, tail
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 24 and "failed" not in tail, tail


def test_train_frame_mimo_shared_trunk_matches_oracle(aslp, oracle, dev, tmp_path):
    """aslp-nnet-train-frame-mimo (aslp-nnetbin/aslp-nnet-train-frame-mimo.cc) on a graph whose two inputs meet in a SHARED trunk and part
    again into two heads: in0 -> Affine -> Sigmoid \\
                                                      (concatenated by column offsets) -> Affine -> Sigmoid -> {Affine -> Softmax -> out0,
                          in1 -> Affine -> Sigmoid /                                                            Affine -> Softmax -> out1}
    The trunk's out-diff is the SUM of the two heads' in-diffs (nnet-nnet.cc:133-144), its in-diff is cut at the column offsets back to the
    two branches (:86-95).  Against the oracle's AffineTransform / Sigmoid / Softmax / Xent chain over the same shuffled minibatches: the
    trained parameters, the gradient applied to every tensor, both losses' reports."""
    rng = np.random.default_rng(23)
    d0, d1, h0, h1, ht, a0, a1, mb = 10, 14, 24, 16, 32, 7, 5, 16
    mk = lambda o, i, s=0.3: (rng.standard_normal((o, i)).astype(np.float32) * s, rng.standard_normal(o).astype(np.float32) * 0.1)
    P = {"b0": mk(h0, d0), "b1": mk(h1, d1), "t": mk(ht, h0 + h1), "o0": mk(a0, ht), "o1": mk(a1, ht)}
    comps = [
        dict(marker="<InputLayer>", dim_in=d0, dim_out=d0, id=0, inputs=[-1], offsets=[0]),
        dict(marker="<InputLayer>", dim_in=d1, dim_out=d1, id=1, inputs=[-1], offsets=[0]),
        dict(marker="<AffineTransform>", dim_in=d0, dim_out=h0, id=2, inputs=[0], offsets=[0], data=nnet_io.affine(*P["b0"])),
        dict(marker="<Sigmoid>", dim_in=h0, dim_out=h0, id=3, inputs=[2], offsets=[0]),
        dict(marker="<AffineTransform>", dim_in=d1, dim_out=h1, id=4, inputs=[1], offsets=[0], data=nnet_io.affine(*P["b1"])),
        dict(marker="<Sigmoid>", dim_in=h1, dim_out=h1, id=5, inputs=[4], offsets=[0]),
        dict(marker="<AffineTransform>", dim_in=h0 + h1, dim_out=ht, id=6, inputs=[3, 5], offsets=[0, h0], data=nnet_io.affine(*P["t"])),   # the shared trunk
        dict(marker="<Sigmoid>", dim_in=ht, dim_out=ht, id=7, inputs=[6], offsets=[0]),
        dict(marker="<AffineTransform>", dim_in=ht, dim_out=a0, id=8, inputs=[7], offsets=[0], data=nnet_io.affine(*P["o0"])),
        dict(marker="<Softmax>", dim_in=a0, dim_out=a0, id=9, inputs=[8], offsets=[0]),
        dict(marker="<OutputLayer>", dim_in=a0, dim_out=a0, id=10, inputs=[9], offsets=[0]),
        dict(marker="<AffineTransform>", dim_in=ht, dim_out=a1, id=11, inputs=[7], offsets=[0], data=nnet_io.affine(*P["o1"])),
        dict(marker="<Softmax>", dim_in=a1, dim_out=a1, id=12, inputs=[11], offsets=[0]),
        dict(marker="<OutputLayer>", dim_in=a1, dim_out=a1, id=13, inputs=[12], offsets=[0]),
    ]
    nnet_io.write_graph_nnet(tmp_path / "trunk.nnet", comps)
    keys = ["t%02d" % i for i in range(10)]
    lens = [int(x) for x in rng.integers(15, 50, len(keys))]
    feats = [[rng.standard_normal((n, d)).astype(np.float32) for n in lens] for d in (d0, d1)]
    posts = [[[[(int(rng.integers(0, a)), 1.0)] for _ in range(n)] for n in lens] for a in (a0, a1)]
    for b in range(2):
        (tmp_path / ("f%d.ark" % b)).write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats[b])]))
        (tmp_path / ("p%d.ark" % b)).write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts[b])]))
    lr, mom, seed, rsize = 0.02, 0.5, 5, 120
    p = tool("aslp-nnet-train-frame-mimo", "--learn-rate=%g" % lr, "--momentum=%g" % mom, "--minibatch-size=%d" % mb, "--randomizer-size=%d" % rsize,
             "--randomizer-seed=%d" % seed, "--objective-function=xent:xent", "ark:%s" % (tmp_path / "f0.ark"), "ark:%s" % (tmp_path / "f1.ark"),
             "ark:%s" % (tmp_path / "p0.ark"), "ark:%s" % (tmp_path / "p1.ark"), str(tmp_path / "trunk.nnet"), str(tmp_path / "trunk.out"))
    err = p.stderr.decode()
    assert "Nnet num_input 2 num_output 2" in err and err.count("FRAME_ACCURACY") == 2

    # the oracle chain; one shuffle mask per cache fill serves all four streams, so the two inputs' minibatches are drawn by the same mask
    A = {k: oracle.Affine(*v) for k, v in P.items()}
    order = ["b0", "b1", "t", "o0", "o1"]   # GetGpuParams / file order (component ids 2, 4, 6, 8, 11)
    st = [dict(frames=0.0, correct=0.0, loss=0.0, entropy=0.0), dict(frames=0.0, correct=0.0, loss=0.0, entropy=0.0)]
    # (one generator over the column-wise concatenation of the two inputs with paired targets: the mask sequence is drawn once per fill)
    cat_feats = [np.concatenate([f0, f1], axis=1) for f0, f1 in zip(feats[0], feats[1])]
    cat_posts = [list(zip(q0, q1)) for q0, q1 in zip(posts[0], posts[1])]
    both = (((np.ascontiguousarray(x[:, :d0]), [q[0] for q in t], None), (np.ascontiguousarray(x[:, d0:]), [q[1] for q in t], None))
            for x, t, _ in minibatches(aslp, cat_feats, cat_posts, mb, seed, rsize))
    sig = lambda v: oracle.unary("orc_sigmoid", v)
    dsig = lambda y, e: oracle.binary("orc_diff_sigmoid", y, e)
    n_mb = 0
    for (x0, t0, _), (x1, t1, _) in both:
        s0, s1 = sig(A["b0"].propagate(x0)), sig(A["b1"].propagate(x1))
        cat = np.concatenate([s0, s1], axis=1)
        tr = sig(A["t"].propagate(cat))
        diffs = []
        for k, (head, t, a) in enumerate((("o0", t0, a0), ("o1", t1, a1))):
            y = oracle.unary("orc_softmax_rows", A[head].propagate(tr))
            tgt = np.zeros((mb, a), np.float32)
            tgt[np.arange(mb), [fr[0][0] for fr in t]] = 1.0
            diff, s = oracle.xent_eval(np.ones(mb, np.float32), y, tgt)
            for key in st[k]:
                st[k][key] += s[key]
            diffs.append(diff)
        d_tr = A["o0"].backpropagate(diffs[0]) + A["o1"].backpropagate(diffs[1])   # link by add
        A["o0"].update(tr, diffs[0], lr, mom)
        A["o1"].update(tr, diffs[1], lr, mom)
        d_t = dsig(tr, d_tr)
        d_cat = A["t"].backpropagate(d_t)
        A["t"].update(cat, d_t, lr, mom)
        e0, e1 = dsig(s0, np.ascontiguousarray(d_cat[:, :h0])), dsig(s1, np.ascontiguousarray(d_cat[:, h0:]))
        A["b0"].update(x0, e0, lr, mom)
        A["b1"].update(x1, e1, lr, mom)
        n_mb += 1
    assert n_mb >= 10
    got = aslp.Nnet.Read(tmp_path / "trunk.out").GetParams()
    want = np.concatenate([np.concatenate([A[k].W.ravel(), A[k].b]) for k in order])
    init = np.concatenate([np.concatenate([P[k][0].ravel(), P[k][1]]) for k in order])
    assert got.shape == want.shape and oracle.rel_err(got, want) < TOL
    off = 0
    for k in order:   # what the run moved each tensor by (a shared-trunk error would sit in "t", "b0", "b1" only)
        n = P[k][0].size + P[k][1].size
        assert oracle.rel_err(got[off:off + n] - init[off:off + n], want[off:off + n] - init[off:off + n]) < 3e-4, k
        off += n
    # both reports: "AvgLoss: <xent> (Xent), ... Frame: <n>" in output order
    losses = [float(x) for x in re.findall(r"AvgLoss: (\S+) \(Xent\)", err)][-2:]
    for k in range(2):
        assert abs(losses[k] - (st[k]["loss"] - st[k]["entropy"]) / st[k]["frames"]) <= 2e-4 * abs(losses[k])
