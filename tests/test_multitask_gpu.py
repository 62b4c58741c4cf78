"""MultiTaskLoss (src/aslp-nnet/nnet-loss.cc:296-407) through the tool that uses it, aslp-nnet-train-simple
--objective-function=multitask,xent,D1,w1,mse,D2,w2 (aslp-nnetbin/aslp-nnet-train-simple.cc:149-160): the trained model, the per-task
statistics and the Report() text against the CPU oracle (oracle/aslp_oracle.c orc_multitask_eval + the oracle's AffineTransform /
Sigmoid chain) stepping through the same minibatches."""
import re

import numpy as np
import pytest

import kaldi_formats as kf
import nnet_io
from test_tools_gpu import minibatches, tool

pytestmark = pytest.mark.gpu
TOL = 1e-4


def g6(x):
    """a double streamed by std::ostream at its default precision"""
    return "%g" % x


def test_train_simple_multitask_matches_oracle(aslp, oracle, dev, tmp_path):
    rng = np.random.default_rng(61)
    D, H, D1, D2, mb = 12, 24, 10, 6, 16
    w1, w2 = 1.0, 0.25
    spec = [("xent", D1, w1), ("mse", D2, w2)]
    W1, b1 = rng.standard_normal((H, D)).astype(np.float32) * 0.3, rng.standard_normal(H).astype(np.float32) * 0.1
    W2, b2 = rng.standard_normal((D1 + D2, H)).astype(np.float32) * 0.3, rng.standard_normal(D1 + D2).astype(np.float32) * 0.1
    nnet_io.write_simple_nnet(tmp_path / "m.nnet", [("<AffineTransform>", D, H, nnet_io.affine(W1, b1)), ("<Sigmoid>", H, H, b""),
                                                     ("<AffineTransform>", H, D1 + D2, nnet_io.affine(W2, b2)),
                                                     ("<Sigmoid>", D1 + D2, D1 + D2, b"")])
    keys = ["mt%02d" % i for i in range(9)]
    lens = [int(x) for x in rng.integers(12, 40, len(keys))]
    feats = [rng.standard_normal((n, D)).astype(np.float32) for n in lens]
    posts, fws = [], []
    for n in lens:
        post = []
        for t in range(n):
            if t % 5 == 3:   # a soft classification target
                a, b = rng.choice(D1, 2, replace=False)
                fr = [(int(a), 0.75), (int(b), 0.25)]
            else:
                fr = [(int(rng.integers(0, D1)), 1.0)]
            fr += [(D1 + c, float(np.float32(rng.uniform(0.05, 0.95)))) for c in range(D2)]   # the regression block's targets
            post.append(fr)
        posts.append(post)
        fws.append(rng.choice(np.array([0.0, 1.0, 1.0, 2.0], np.float32), n))
    (tmp_path / "f.ark").write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
    (tmp_path / "p.ark").write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts)]))
    (tmp_path / "w.ark").write_bytes(kf.archive([(k, kf.vector_bin(x)) for k, x in zip(keys, fws)]))
    lr, mom = 0.02, 0.5
    obj = "multitask,xent,%d,%g,mse,%d,%g" % (D1, w1, D2, w2)
    p = tool("aslp-nnet-train-simple", "--objective-function=" + obj, "--learn-rate=%g" % lr, "--momentum=%g" % mom, "--minibatch-size=%d" % mb,
             "--randomize=false", "--frame-weights=ark:%s" % (tmp_path / "w.ark"), "ark:%s" % (tmp_path / "f.ark"), "ark:%s" % (tmp_path / "p.ark"),
             str(tmp_path / "m.nnet"), str(tmp_path / "m.out"))
    err = p.stderr.decode()
    assert "Done 9 files, 0 with no tgt_mats, 0 with other errors. [TRAINING, NOT-RANDOMIZED" in err

    # the oracle chain over the same minibatches (nnet-nnet.cc:70-154: backward per component with its Update straight behind)
    A1, A2 = oracle.Affine(W1, b1), oracle.Affine(W2, b2)
    xs = dict(frames=0.0, correct=0.0, loss=0.0, entropy=0.0, likelyhood=0.0)
    ms = dict(loss=0.0, frames=0.0)
    n_mb = 0
    for x, t, fw in minibatches(aslp, feats, posts, mb, 0, 32768, shuffle=False, weights=fws):
        tgt = np.zeros((mb, D1 + D2), np.float32)
        for r, fr in enumerate(t):
            for c, v in fr:
                tgt[r, c] = v
        h = oracle.unary("orc_sigmoid", A1.propagate(x))
        y = oracle.unary("orc_sigmoid", A2.propagate(h))
        diff, st = oracle.multitask_eval(spec, fw, y, tgt)
        for k in xs:
            xs[k] += st[0][k]
        for k in ms:
            ms[k] += st[1][k]
        d2 = oracle.binary("orc_diff_sigmoid", y, diff)
        dh = A2.backpropagate(d2)
        A2.update(h, d2, lr, mom)
        d1 = oracle.binary("orc_diff_sigmoid", h, dh)
        A1.update(x, d1, lr, mom)
        n_mb += 1
    assert n_mb >= 10
    got = aslp.Nnet.Read(tmp_path / "m.out").GetParams()
    want = np.concatenate([A1.W.ravel(), A1.b, A2.W.ravel(), A2.b])
    assert got.shape == want.shape
    assert oracle.rel_err(got, want) < TOL
    # the gradient actually applied over the run, per tensor (|lr g| << |W| hides a wrong gradient in the parameters' own error)
    init = np.concatenate([W1.ravel(), b1, W2.ravel(), b2])
    o = 0
    for ten in (W1, b1, W2, b2):
        n = ten.size
        assert oracle.rel_err(got[o:o + n] - init[o:o + n], want[o:o + n] - init[o:o + n]) < 2e-4
        o += n

    # Report(), nnet-loss.cc:370-393: header, one "Loss i, <task report>" per task (each followed by a blank line: the task's own endl and
    # Report's), then the overall line with both vectors streamed element by element (nnet-utils.h:42-45: no brackets, a blank behind each)
    xent_avg = (xs["loss"] - xs["entropy"]) / xs["frames"]
    mse_avg = ms["loss"] / ms["frames"]
    overall = np.float32(w1) * np.float32(xent_avg) + np.float32(w2) * np.float32(mse_avg)
    m = re.search(r"MultiTaskLoss, with 2 parallel loss functions\.\n"
                  r"Loss 1, AvgLoss: (\S+) \(Xent\), Likelyhood: (\S+) Frame: (\S+)\nFRAME_ACCURACY >> (\S+)% <<\n\n"
                  r"Loss 2, AvgLoss: (\S+) \(Mse\), \[RMS (\S+), frames (\S+)\]\n\n"
                  r"Loss \(OVERALL\), AvgLoss: (\S+) \(MultiTaskLoss\), weights (\S+) (\S+) , values (\S+) (\S+) \n", err)
    assert m, err[-1500:]
    v = [float(x) for x in m.groups()]
    close = lambda a, b: abs(a - b) <= 2e-4 * max(abs(b), 1e-3)
    assert close(v[0], xent_avg) and close(v[1], xs["likelyhood"] / xs["frames"]) and v[2] == xs["frames"]
    assert close(v[3], 100.0 * xs["correct"] / xs["frames"])
    assert close(v[4], mse_avg) and close(v[5], np.sqrt(ms["loss"] / ms["frames"] / D2)) and v[6] == ms["frames"]
    assert close(v[7], overall) and m.group(9) == g6(w1) and m.group(10) == g6(w2) and close(v[10], xent_avg) and close(v[11], mse_avg)
    # what the bash scheduler reads from this log (train_scheduler.sh:120: the LAST "AvgLoss:" line, 4th token) is the overall loss
    last = [ln for ln in err.splitlines() if "AvgLoss:" in ln][-1]
    assert last.startswith("Loss (OVERALL),") and close(float(last.split()[3]), overall)

    # cross-validation with the same objective: no update, same report vocabulary
    p = tool("aslp-nnet-train-simple", "--cross-validate=true", "--objective-function=" + obj, "--minibatch-size=%d" % mb,
             "--frame-weights=ark:%s" % (tmp_path / "w.ark"), "ark:%s" % (tmp_path / "f.ark"), "ark:%s" % (tmp_path / "p.ark"), str(tmp_path / "m.out"))
    assert b"[CROSS-VALIDATION, RANDOMIZED" in p.stderr and b"Loss (OVERALL), AvgLoss:" in p.stderr
    # a malformed description: the reference asserts on triplets (nnet-loss.cc:300)
    p = tool("aslp-nnet-train-simple", "--objective-function=multitask,xent,%d" % D1, "ark:%s" % (tmp_path / "f.ark"), "ark:%s" % (tmp_path / "p.ark"),
             str(tmp_path / "m.nnet"), str(tmp_path / "x.out"), ok=False)
    assert p.returncode != 0 and b"Assertion failed" in p.stderr
    # a sum of task widths that is not the network's output width (nnet-loss.cc:348)
    p = tool("aslp-nnet-train-simple", "--objective-function=multitask,xent,%d,1.0,mse,%d,1.0" % (D1, D2 + 1), "--minibatch-size=%d" % mb, "ark:%s" % (tmp_path / "f.ark"),
             "ark:%s" % (tmp_path / "p.ark"), str(tmp_path / "m.nnet"), str(tmp_path / "x.out"), ok=False)
    assert p.returncode != 0 and b"Assertion failed" in p.stderr
