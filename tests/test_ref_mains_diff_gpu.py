"""Differential test of the engine's command-line tools against the REFERENCE's own main() functions (seam B7 / B4).

kaldi-aslp_amd/bin_ref/<tool> is src/aslp-nnetbin/<tool>.cc of the reference, compiled unchanged against include/aslp_compat_kaldi.h and
linked with this engine (`make -C kaldi-aslp_amd refmains`, development container; the binaries travel, the sources do not);
kaldi-aslp_amd/bin/<tool> is the engine's own tool of that name.  Both sit on the same library, so whatever differs between their outputs
is a difference in the TOOL's logic -- option handling, utterance bookkeeping, padding, learning-rate arithmetic, what is logged.  Every
case below runs both on the same inputs with the same flags (option combinations the per-tool tests do not reach: cross-validation,
frame / utterance weights, --drop-len, --skip-width, --frame-limit, --length-tolerance, text models, feature transforms, priors ...)
and wants the written models / archives byte for byte and the log lines the schedulers read (AvgLoss, FRAME_ACCURACY, TOKEN_ACCURACY,
Done ...) word for word, times aside."""
import os
import re
import subprocess

import numpy as np
import pytest

import kaldi_formats as kf
from test_tools_gpu import BLSTM_PROTO, CTC_PROTO, FSMN_PROTO, LC_PROTO, LSTM_PROTO, PROTO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OWN = os.path.join(ROOT, "kaldi-aslp_amd", "bin")
REF = os.path.join(ROOT, "kaldi-aslp_amd", "bin_ref")


def run(bindir, name, args, ok=True):
    p = subprocess.run([os.path.join(bindir, name)] + list(args), capture_output=True, timeout=1800)
    if ok:
        assert p.returncode == 0, "%s/%s %s\n%s" % (os.path.basename(bindir), name, " ".join(args), p.stderr.decode()[-3000:])
    return p


# what differs from run to run or from box to box, not from tool to tool: device selection and memory figures, the profile table, timings
VOLATILE = re.compile(r"free:|Memory used|hipSetDevice|Selected device|active GPU|^-+\[|^-+$|^\S+\s+[0-9.e+-]+s$|Time\[|time elapsed|\t|^\s*$|Total GPU time|profil|Propagate time \S+s"
                      # the engine's warning to a caller that writes through GetGpuParams() pointers without saying so (the reference's workers do;
                      # the engine's own tools announce their writes): expected on one side only
                      r"|parameters are aliased through GetGpuParams by a writer that does not announce")


def log_lines(stderr):
    """EVERY log line of a run -- LOG, WARNING, VLOG, what the schedulers read and what a person reads -- without the
    `LOG (tool:function():file:line)` tag and the echoed command line, with times and rates masked"""
    out = []
    for ln in stderr.decode(errors="replace").splitlines():
        if re.match(r"^/\S*/kaldi-aslp_amd/bin(_ref)?/", ln):
            continue   # (the echoed command line; untagged lines otherwise are the later lines of multi-line reports and stay)
        ln = re.sub(r"^(LOG|WARNING|ERROR|VLOG(\[\d+\])?) \(\S+\)\s*", "", ln)
        if VOLATILE.search(ln):
            continue
        ln = re.sub(r"[-+0-9.e]+ ?min", "<t> min", ln)
        ln = re.sub(r"fps ?[-+0-9.einfa]+", "fps<r>", ln)
        ln = re.sub(r"in [-+0-9.e]+min", "in <t>min", ln)
        ln = re.sub(r"Time\[[^\]]*\]|\[[-+0-9.e]+ ?s\]", "<t>", ln)
        out.append(ln.rstrip())
    return out


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    if not (os.path.isdir(REF) and len(os.listdir(REF)) >= 23):
        pytest.skip("bin_ref/ not built (needs the reference tree: make -C kaldi-aslp_amd refmains)")
    return build_corpus(tmp_path_factory.mktemp("refdiff"))


def build_corpus(d):
    """one set of small tables for every case: 12-dim features of 9 utterances (one without targets, one far too long), posteriors over 10
    classes, frame weights, utterance weights, CTC label sequences over 8 tokens + blank; 20-dim features / 30 classes for the DNN"""
    rng = np.random.default_rng(2026)
    keys = ["u%02d" % i for i in range(9)]
    lens = [23, 7, 41, 12, 30, 5, 18, 64, 27]
    c = {"dir": d, "keys": keys, "lens": lens}
    for tag, dim, ncls in (("s", 12, 10), ("d", 20, 30)):
        feats = [rng.standard_normal((n, dim)).astype(np.float32) for n in lens]
        posts = [[[(int(rng.integers(0, ncls)), 1.0)] for _ in range(n)] for n in lens]
        (d / (tag + "_feats.ark")).write_bytes(kf.archive([(k, kf.matrix_bin(f)) for k, f in zip(keys, feats)]))
        (d / (tag + "_post.ark")).write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts) if k != "u05"]))   # u05: no targets
        posts[2], posts[6] = posts[2][:-2], posts[6][:-9]   # two frames short: inside the default --length-tolerance of 5; nine: outside
        (d / (tag + "_postshort.ark")).write_bytes(kf.archive([(k, kf.posterior_bin(p)) for k, p in zip(keys, posts) if k != "u05"]))
        tgts = [rng.uniform(0.05, 0.95, (n, ncls)).astype(np.float32) for n in lens]   # aslp-nnet-train-mse reads its targets as matrices
        (d / (tag + "_tgt.ark")).write_bytes(kf.archive([(k, kf.matrix_bin(t)) for k, t in zip(keys, tgts)]))   # (read in lock step with the features: same keys)
        tgts[2], tgts[6] = tgts[2][:-2], tgts[6][:-9]
        (d / (tag + "_tgtshort.ark")).write_bytes(kf.archive([(k, kf.matrix_bin(t)) for k, t in zip(keys, tgts)]))
        wts = [rng.choice(np.array([0.0, 1.0, 1.0, 0.5], np.float32), n) for n in lens]
        (d / (tag + "_fw.ark")).write_bytes(kf.archive([(k, kf.vector_bin(w)) for k, w in zip(keys, wts)]))
    (d / "uw.ark").write_bytes(kf.archive([(k, b"\0B" + b"\x04" + np.float32(0.5 + 0.25 * (i % 3)).tobytes()) for i, k in enumerate(keys)]))
    labels = [[int(x) for x in rng.integers(1, 9, max(1, n // 5))] for n in lens]
    (d / "lab.ark").write_bytes(kf.archive([(k, kf.int32vec_bin(l)) for k, l in zip(keys, labels) if k != "u05"]))
    (d / "tokens.txt").write_text("".join("tok%d %d\n" % (i, i) for i in range(9)))
    (d / "counts").write_text("[ " + " ".join(str(int(x)) for x in rng.integers(1, 500, 30)) + " ]\n")
    (d / "counts10").write_text("[ " + " ".join(str(int(x)) for x in rng.integers(1, 500, 10)) + " ]\n")
    protos = {"dnn": PROTO, "lstm": LSTM_PROTO, "blstm": BLSTM_PROTO, "lc": LC_PROTO, "ctc": CTC_PROTO, "fsmn": FSMN_PROTO,
              "uctc": LSTM_PROTO.replace("<OutputDim> 10", "<OutputDim> 9").replace("<InputDim> 10", "<InputDim> 9"),   # one utterance at a time
              "tr": "<NnetProto>\n<AffineTransform> <InputDim> 12 <OutputDim> 12 <BiasMean> 0.0 <BiasRange> 0.1 <ParamStddev> 0.5\n</NnetProto>\n",
              "mse": "<NnetProto>\n<AffineTransform> <InputDim> 12 <OutputDim> 10 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.2\n"
                     "<Sigmoid> <InputDim> 10 <OutputDim> 10\n</NnetProto>\n"}
    for i, (name, text) in enumerate(sorted(protos.items())):
        (d / (name + ".proto")).write_text(text)
        run(OWN, "aslp-nnet-init", ["--seed=%d" % (100 + i), str(d / (name + ".proto")), str(d / (name + ".nnet"))])
    return c


def both(corpus, name, flags, inputs, outputs=("model",), tag=""):
    """runs bin/<name> and bin_ref/<name>; `outputs`: "model" (a file path appended last) or "ark" (an ark: wspecifier appended last)"""
    d = corpus["dir"]
    res = {}
    for side, bindir in (("own", OWN), ("ref", REF)):
        outs = []
        for kind in outputs:
            path = str(d / ("%s%s.%s.%s" % (name, tag, side, kind)))
            if os.path.exists(path):
                os.remove(path)
            outs.append("ark:" + path if kind == "ark" else path)
        extra = ["--use-gpu=yes"] if name.startswith("aslp-nnet-forward") else []   # (the reference's forward tools default to "no")
        p = run(bindir, name, extra + list(flags) + list(inputs) + outs)
        res[side] = (p, [o[4:] if o.startswith("ark:") else o for o in outs])
    (po, fo), (pr, fr) = res["own"], res["ref"]
    for a, b in zip(fo, fr):
        assert os.path.exists(a) == os.path.exists(b), (a, b)
        if os.path.exists(a):
            assert open(a, "rb").read() == open(b, "rb").read(), "%s %s: %s differs from the reference main's" % (name, " ".join(flags), os.path.basename(a))
    return log_lines(po.stderr), log_lines(pr.stderr), po, pr


def tables(corpus, tag, *names):
    d = corpus["dir"]
    return ["ark:%s" % (d / ("%s_%s.ark" % (tag, n))) for n in names]


FRAME_FLAGS = [
    [],
    ["--cross-validate=true"],
    ["--randomize=false", "--minibatch-size=16"],
    ["--momentum=0.9", "--l2-penalty=1e-4", "--l1-penalty=1e-6", "--randomizer-seed=5", "--randomizer-size=64", "--minibatch-size=8"],
    ["--binary=false", "--dropout-retention=1.0", "--report-period=40"],
]


@pytest.mark.parametrize("flags", FRAME_FLAGS, ids=lambda f: " ".join(f) or "defaults")
def test_train_frame(corpus, flags):
    d = corpus["dir"]
    flags = ["--learn-rate=0.01", "--minibatch-size=32"] + flags if not any("minibatch" in f for f in flags) else ["--learn-rate=0.01"] + flags
    cv = "--cross-validate=true" in flags
    lo, lr, _, _ = both(corpus, "aslp-nnet-train-frame", flags, tables(corpus, "d", "feats", "post") + [str(d / "dnn.nnet")], () if cv else ("model",),
                        tag=str(abs(hash(tuple(flags))) % 997))
    assert lo == lr and any("AvgLoss" in x for x in lo)


SIMPLE_FLAGS = [
    [],
    ["--cross-validate=true"],
    ["--frame-weights=ark:{d}/d_fw.ark"],
    ["--utt-weights=ark:{d}/uw.ark", "--randomize=false"],
    ["--frame-weights=ark:{d}/d_fw.ark", "--utt-weights=ark:{d}/uw.ark", "--length-tolerance=1"],
    ["--objective-function=mse", "--learn-rate=0.001"],
    ["--objective-function=multitask,xent,20,1.0,mse,10,0.1", "--learn-rate=0.001"],
]


@pytest.mark.parametrize("tool_name", ["aslp-nnet-train-simple", "aslp-nnet-train-mse"])
@pytest.mark.parametrize("flags", SIMPLE_FLAGS, ids=lambda f: " ".join(x.replace("{d}/", "") for x in f) or "defaults")
def test_train_simple_and_mse(corpus, tool_name, flags):
    d = corpus["dir"]
    if tool_name.endswith("mse") and any("multitask" in f for f in flags):
        pytest.skip("aslp-nnet-train-mse has no multitask objective")
    flags = ["--learn-rate=0.01", "--minibatch-size=16", "--randomizer-size=100"] + [f.format(d=d) for f in flags]
    cv = "--cross-validate=true" in flags
    post = "postshort" if any("utt-weights" in f for f in flags) else "post"   # (some cases on targets that are a few frames short or far too short)
    if tool_name.endswith("mse"):
        post = post.replace("post", "tgt")
    lo, lr, _, _ = both(corpus, tool_name, flags, tables(corpus, "d", "feats", post) + [str(d / "dnn.nnet")], () if cv else ("model",),
                        tag=str(abs(hash(tuple(flags))) % 997))
    assert lo == lr and any("AvgLoss" in x for x in lo)


def test_train_simple_with_a_feature_transform(corpus):
    d = corpus["dir"]
    for flags in (["--feature-transform=%s" % (d / "tr.nnet")], ["--feature-transform=%s" % (d / "tr.nnet"), "--objective-function=mse"]):
        lo, lr, _, _ = both(corpus, "aslp-nnet-train-simple", ["--learn-rate=0.005", "--minibatch-size=16"] + flags,
                            tables(corpus, "s", "feats", "post") + [str(d / "mse.nnet")], tag="tr%d" % len(flags))
        assert lo == lr


def test_train_perutt(corpus):
    d = corpus["dir"]
    for i, flags in enumerate(([], ["--cross-validate=true"], ["--frame-weights=ark:%s" % (d / "s_fw.ark")], ["--drop-len=40"], ["--randomize=false", "--length-tolerance=1"])):
        cv = "--cross-validate=true" in flags
        lo, lr, _, _ = both(corpus, "aslp-nnet-train-perutt", ["--learn-rate=0.5", "--momentum=0.9"] + flags,
                            tables(corpus, "s", "feats", "postshort" if i >= 2 else "post") + [str(d / "fsmn.nnet")], () if cv else ("model",), tag="p%d" % i)
        assert lo == lr and any("AvgLoss" in x for x in lo)


STREAM_FLAGS = [
    ["--num-stream=3", "--batch-size=5", "--targets-delay=2"],
    ["--num-stream=2", "--batch-size=7", "--targets-delay=0", "--cross-validate=true"],
    ["--num-stream=4", "--batch-size=4", "--targets-delay=3", "--report-period=3", "--momentum=0.9"],
    ["--num-stream=3", "--batch-size=6", "--drop-len=40"],
]


@pytest.mark.parametrize("flags", STREAM_FLAGS, ids=lambda f: " ".join(f))
def test_train_lstm_streams(corpus, flags):
    d = corpus["dir"]
    cv = "--cross-validate=true" in flags
    lo, lr, _, _ = both(corpus, "aslp-nnet-train-lstm-streams", ["--learn-rate=0.02"] + flags, tables(corpus, "s", "feats", "post") + [str(d / "lstm.nnet")],
                        () if cv else ("model",), tag=str(abs(hash(tuple(flags))) % 997))
    # (the reference's main forms its final report and drops it, aslp-nnet-train-lstm-streams.cc:221; the engine's tool logs it: one line more)
    final = [x for x in lo if x not in lr]
    assert [x for x in lo if x not in final] == lr and len(final) <= 2


@pytest.mark.parametrize("flags", [f + ["--skip-width=%d" % w] for f in STREAM_FLAGS[:3] for w in (1, 3)], ids=lambda f: " ".join(f))
def test_train_lstm_streams_skip(corpus, flags):
    d = corpus["dir"]
    cv = "--cross-validate=true" in flags
    lo, lr, _, _ = both(corpus, "aslp-nnet-train-lstm-streams-skip", ["--learn-rate=0.02"] + flags, tables(corpus, "s", "feats", "post") + [str(d / "lstm.nnet")],
                        () if cv else ("model",), tag=str(abs(hash(tuple(flags))) % 997))
    assert lo == lr


BLSTM_FLAGS = [
    ["--num-stream=3"],
    ["--num-stream=2", "--cross-validate=true"],
    ["--num-stream=4", "--frame-limit=100"],
    ["--num-stream=3", "--frame-weights=ark:{d}/s_fw.ark", "--momentum=0.9"],
    ["--num-stream=3", "--length-tolerance=1"],
    ["--num-stream=2", "--report-period=2", "--objective-function=xent"],
]


@pytest.mark.parametrize("tool_name", ["aslp-nnet-train-blstm-streams", "aslp-nnet-train-blstm-parallel"])
@pytest.mark.parametrize("flags", BLSTM_FLAGS, ids=lambda f: " ".join(x.replace("{d}/", "") for x in f))
def test_train_blstm_whole_utterances(corpus, tool_name, flags):
    d = corpus["dir"]
    flags = [f.format(d=d) for f in flags]
    cv = "--cross-validate=true" in flags
    post = "postshort" if any("length-tolerance" in f or "frame-weights" in f for f in flags) else "post"
    lo, lr, _, _ = both(corpus, tool_name, ["--learn-rate=0.2" if tool_name.endswith("streams") else "--learn-rate=0.002"] + flags,
                        tables(corpus, "s", "feats", post) + [str(d / "blstm.nnet")], () if cv else ("model",), tag=str(abs(hash(tuple(flags))) % 997))
    assert lo == lr and any("AvgLoss" in x for x in lo)


@pytest.mark.parametrize("flags", [["--skip-width=2", "--num-stream=3"], ["--drop-len=40", "--num-stream=3"], ["--drop-len=25", "--num-stream=2", "--skip-width=3"]],
                         ids=lambda f: " ".join(f))
def test_train_blstm_streams_skip_and_drop(corpus, flags):
    """--drop-len in this tool steps over the utterance AFTER a dropped one as well (feature_reader.Next() in front of `continue`,
    aslp-nnet-train-blstm-streams.cc:165-169): whatever the reference's loop does with that, the engine's does"""
    d = corpus["dir"]
    lo, lr, _, _ = both(corpus, "aslp-nnet-train-blstm-streams", ["--learn-rate=0.2"] + flags, tables(corpus, "s", "feats", "post") + [str(d / "blstm.nnet")],
                        tag=str(abs(hash(tuple(flags))) % 997))
    assert lo == lr


LC_FLAGS = [
    ["--num-stream=3", "--chunk-size=6", "--right-splice=3"],
    ["--num-stream=2", "--chunk-size=8", "--right-splice=4", "--cross-validate=true"],
    ["--num-stream=3", "--chunk-size=6", "--right-splice=3", "--drop-len=40", "--momentum=0.9"],
    ["--num-stream=4", "--chunk-size=5", "--right-splice=2", "--report-period=3"],
]


@pytest.mark.parametrize("flags", LC_FLAGS, ids=lambda f: " ".join(f))
def test_train_blstm_streams_lc(corpus, flags):
    d = corpus["dir"]
    cv = "--cross-validate=true" in flags
    lo, lr, _, _ = both(corpus, "aslp-nnet-train-blstm-streams-lc", ["--learn-rate=0.02"] + flags, tables(corpus, "s", "feats", "post") + [str(d / "lc.nnet")],
                        () if cv else ("model",), tag=str(abs(hash(tuple(flags))) % 997))
    assert lo == lr and any("AvgLoss" in x for x in lo)


CTC_FLAGS = [
    ["--num-stream=3"],
    ["--num-stream=2", "--cross-validate=true"],
    ["--num-stream=4", "--frame-limit=100", "--momentum=0.9"],
    ["--num-stream=3", "--drop-len=40", "--report-step=2"],
    ["--num-stream=3", "--skip-width=2"],
]


@pytest.mark.parametrize("tool_name", ["aslp-nnet-train-ctc-streams", "aslp-nnet-train-warp-ctc-streams"])
@pytest.mark.parametrize("flags", CTC_FLAGS, ids=lambda f: " ".join(f))
def test_train_ctc_streams(corpus, tool_name, flags):
    d = corpus["dir"]
    cv = "--cross-validate=true" in flags
    lo, lr, _, _ = both(corpus, tool_name, ["--learn-rate=0.05"] + flags, ["ark:%s" % (d / "s_feats.ark"), "ark:%s" % (d / "lab.ark"), str(d / "ctc.nnet")],
                        () if cv else ("model",), tag=str(abs(hash(tuple(flags))) % 997))
    assert lo == lr and any("TOKEN_ACCURACY" in x or "Obj" in x for x in lo)


def test_train_ctc_one_utterance_at_a_time(corpus):
    d = corpus["dir"]
    for i, flags in enumerate(([], ["--cross-validate=true"], ["--drop-len=40", "--report-step=2", "--token-symbol-table=%s" % (d / "tokens.txt")])):
        cv = "--cross-validate=true" in flags
        lo, lr, po, pr = both(corpus, "aslp-nnet-train-ctc", ["--learn-rate=0.001", "--momentum=0.9"] + flags,
                              ["ark:%s" % (d / "s_feats.ark"), "ark:%s" % (d / "lab.ark"), str(d / "uctc.nnet")], () if cv else ("model",), tag="c%d" % i)
        assert lo == lr
        if i == 2:   # the hypotheses' token names, one line per utterance
            hyp = lambda p: [ln for ln in p.stderr.decode().splitlines() if re.match(r"u\d\d( tok\d)*\s*$", ln)]
            assert hyp(po) == hyp(pr) and len(hyp(po)) >= 6


FORWARD_FLAGS = [
    ["--apply-log=false"],
    [],
    ["--no-softmax=true", "--apply-log=false"],
    ["--class-frame-counts={d}/counts", "--prior-scale=0.8"],
    ["--no-softmax=true", "--apply-log=false", "--class-frame-counts={d}/counts", "--prior-floor=1e-3"],
    ["--time-shift=2", "--apply-log=false"],
    ["--skip-width=3", "--apply-log=false"],
]


@pytest.mark.parametrize("flags", FORWARD_FLAGS, ids=lambda f: " ".join(x.replace("{d}/", "") for x in f) or "defaults")
def test_forward(corpus, flags):
    d = corpus["dir"]
    flags = [f.format(d=d) for f in flags]
    lo, lr, _, _ = both(corpus, "aslp-nnet-forward", flags, [str(d / "dnn.nnet"), "ark:%s" % (d / "d_feats.ark")], ("ark",), tag=str(abs(hash(tuple(flags))) % 997))
    assert lo == lr


def test_forward_recurrent_and_ctc_options(corpus):
    d = corpus["dir"]
    cases = [("aslp-nnet-forward", ["--apply-log=false"], "lstm"), ("aslp-nnet-forward", ["--apply-log=false", "--feature-transform=%s" % (d / "tr.nnet")], "blstm"),
             ("aslp-nnet-forward", ["--add-softmax=true", "--apply-log=true", "--scale-blank=0.5"], "ctc"),
             ("aslp-nnet-forward-skip", ["--skip-width=2", "--apply-log=false"], "lstm"), ("aslp-nnet-forward-skip", ["--skip-width=3", "--add-softmax=true", "--scale-blank=1.5"], "ctc"),
             ("aslp-nnet-forward-skip", ["--skip-width=1", "--class-frame-counts=%s" % (d / "counts10")], "blstm"),
             ("aslp-nnet-forward-blstm-lc", ["--apply-log=false"], "lc"), ("aslp-nnet-forward-blstm-lc", ["--chunk-size=16", "--right-splice=8"], "lc"),
             ("aslp-nnet-forward-blstm-lc", ["--chunk-size=6", "--right-splice=3", "--class-frame-counts=%s" % (d / "counts10")], "lc")]
    for i, (name, flags, net) in enumerate(cases):
        lo, lr, _, _ = both(corpus, name, flags, [str(d / (net + ".nnet")), "ark:%s" % (d / "s_feats.ark")], ("ark",), tag="f%d" % i)
        assert lo == lr, (name, flags)


def test_model_tools(corpus):
    d = corpus["dir"]
    for i, (name, flags, inputs) in enumerate([
            ("aslp-nnet-copy", ["--binary=false"], [str(d / "dnn.nnet")]),
            ("aslp-nnet-copy", [], [str(d / "lc.nnet")]),
            ("aslp-nnet-copy", ["--binary=false"], [str(d / "lstm.nnet")]),
            ("aslp-nnet-copy", ["--binary=false"], [str(d / "blstm.nnet")]),
            ("aslp-nnet-copy", ["--binary=false"], [str(d / "lc.nnet")]),
            ("aslp-nnet-copy", ["--binary=false"], [str(d / "fsmn.nnet")]),
            ("aslp-nnet-dot", [], [str(d / "lc.nnet")]),
            ("aslp-nnet-convert-to-standard", [], [str(d / "dnn.nnet")]),
            ("aslp-nnet-convert-to-standard", ["--binary=false"], [str(d / "fsmn.nnet")]),
            ("aslp-nnet-init", ["--seed=9"], [str(d / "blstm.proto")]),
            ("aslp-nnet-init", ["--seed=9", "--binary=false"], [str(d / "fsmn.proto")]),
            ("aslp-nnet-dot", [], [str(d / "dnn.nnet")])]):
        both(corpus, name, flags, inputs, tag="m%d" % i)
    # aslp-nnet-insert: the hidden layer of a second net in front of the last updatable component, which is drawn afresh unless told not to
    (d / "base.proto").write_text("<NnetProto>\n<AffineTransform> <InputDim> 10 <OutputDim> 64 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.1\n"
                                  "<Sigmoid> <InputDim> 64 <OutputDim> 64\n<AffineTransform> <InputDim> 64 <OutputDim> 5 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.1\n"
                                  "<Softmax> <InputDim> 5 <OutputDim> 5\n</NnetProto>\n")
    (d / "hid.proto").write_text("<NnetProto>\n<AffineTransform> <InputDim> 64 <OutputDim> 64 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.1\n"
                                 "<Sigmoid> <InputDim> 64 <OutputDim> 64\n</NnetProto>\n")
    run(OWN, "aslp-nnet-init", ["--seed=1", str(d / "base.proto"), str(d / "base.nnet")])
    run(OWN, "aslp-nnet-init", ["--seed=2", str(d / "hid.proto"), str(d / "hid.nnet")])
    both(corpus, "aslp-nnet-insert", ["--randomize-next-component=false"], [str(d / "base.nnet"), str(d / "hid.nnet")], tag="i0")
    # (with the component drawn afresh the two are not comparable byte for byte: the reference's main registers --srand and never uses it -- its
    # normals come from the device generator -- while the engine's tool seeds its host generator with it; what both must say is what they did)
    for bindir in (OWN, REF):
        p = run(bindir, "aslp-nnet-insert", ["--stddev-factor=0.2", "--srand=5", str(d / "base.nnet"), str(d / "hid.nnet"), str(d / "ins.rand.nnet")])
        assert b"Inserted 2 components at position 3" in p.stderr and b"Randomized component index 5 with stddev 0.025" in p.stderr
    both(corpus, "aslp-nnet-insert", ["--insert-at=3", "--randomize-next-component=false", "--binary=false"], [str(d / "base.nnet"), "cat %s |" % (d / "hid.nnet")], tag="i2")
    for net in ("dnn", "lstm", "blstm", "lc", "ctc", "fsmn"):
        assert run(OWN, "aslp-nnet-info", [str(d / (net + ".nnet"))]).stdout == run(REF, "aslp-nnet-info", [str(d / (net + ".nnet"))]).stdout


# ---- src/aslp-parallelbin/: the workers and the parameter server on the engine's sync layer -------------------------------------------
WORKER_FLAGS = [
    ["--worker-type=bsp", "--sync-period=64"],
    ["--worker-type=bmuf", "--sync-period=100", "--bmuf-learn-rate=0.9", "--bmuf-momentum=0.5"],
    ["--worker-type=sod", "--sync-period=64", "--solver=adam", "--adam-lr=0.0005"],
    ["--worker-type=sod", "--sync-period=50", "--solver=momentum", "--lr=0.5", "--sgd-momentum=0.8"],
]


@pytest.mark.parametrize("flags", WORKER_FLAGS, ids=lambda f: " ".join(f))
def test_workers_alone_in_their_group(corpus, flags):
    """one rank: every protocol's arithmetic on this rank's own model (BSP: the identity; BMUF: block momentum on w - w_g; SOD: a solver
    stepping the model by its own delta), the sync schedule, and the loop around them -- including the step more the reference's frame
    worker takes at the end of its data (tests/test_parallel_gpu.py)"""
    d = corpus["dir"]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    for name, net, extra in (("aslp-nnet-train-frame-worker", "dnn", ["--minibatch-size=16", "--randomizer-size=100", "--learn-rate=0.01"]),
                             ("aslp-nnet-train-lstm-stream-worker", "lstm", ["--num-stream=3", "--batch-size=5", "--targets-delay=2", "--learn-rate=0.02"]),
                             ("aslp-nnet-train-lc-blstm-streams-worker", "lc", ["--num-stream=3", "--chunk-size=6", "--right-splice=3", "--learn-rate=0.02"])):
        if name != "aslp-nnet-train-frame-worker" and "--solver=momentum" in flags:
            continue   # (every protocol on the frame worker; BSP, BMUF and one SOD solver on the stream workers: the suite's time)
        if name.startswith("aslp-nnet-train-lc") and "--worker-type=sod" in flags:
            continue   # (no SOD worker in that tool: aslp-nnet-train-lc-blstm-streams-worker.cc:17-21)
        outs = {}
        for side, bindir in (("own", OWN), ("ref", REF)):
            out = str(d / ("%s.%s.%s.model" % (name, abs(hash(tuple(flags))) % 997, side)))
            p = subprocess.run([os.path.join(bindir, name)] + flags + extra + tables(corpus, "d" if net == "dnn" else "s", "feats", "post") + [str(d / (net + ".nnet")), out],
                               capture_output=True, timeout=1800, env=env)
            assert p.returncode == 0, (side, name, p.stderr.decode()[-2000:])
            outs[side] = (open(out, "rb").read(), log_lines(p.stderr))
        assert outs["own"][0] == outs["ref"][0], (name, flags)
        lo, lr = outs["own"][1], outs["ref"][1]
        if name != "aslp-nnet-train-frame-worker":   # (the stream workers' final report: formed and dropped by the reference's mains, :265 / :428, logged by the engine's tools -- as aslp-nnet-train-lstm-streams)
            final = [x for x in lo if x not in lr]
            lo = [x for x in lo if x not in final]
            assert len(final) <= 2
        assert lo == lr, (name, flags)


def test_two_bsp_ranks_and_a_server_alone(corpus):
    """two OS processes on one device (shared-memory transport): the reference's worker main finds its rank, the world size and the rendezvous
    file in the environment, as it would under mpirun; the engine's tool takes them as flags.  Rank 0's model byte for byte.  And the
    parameter server with nobody to serve."""
    import secrets
    d = corpus["dir"]
    common = ["--worker-type=bsp", "--sync-period=64", "--minibatch-size=16", "--randomizer-size=100", "--learn-rate=0.01"] + tables(corpus, "d", "feats", "post") + [str(d / "dnn.nnet")]
    models = {}
    for side, bindir in (("own", OWN), ("ref", REF)):
        comm = str(d / ("comm." + side))
        token = secrets.token_hex(6)
        procs = []
        for r in range(2):
            env = dict(os.environ, ASLP_COMM_TOKEN=token, ASLP_COMM_TRANSPORT="shm")
            out = str(d / ("bsp2.%s.%d.model" % (side, r)))
            if side == "own":
                argv = [os.path.join(bindir, "aslp-nnet-train-frame-worker"), "--rank=%d" % r, "--num-workers=2", "--comm-file=" + comm, "--gpu-id=0"] + common + [out]
            else:
                argv = [os.path.join(bindir, "aslp-nnet-train-frame-worker"), "--gpu-id=0"] + common + [out]
                env.update(RANK=str(r), WORLD_SIZE="2", ASLP_COMM_FILE=comm)
            procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
        try:
            for r, p in enumerate(procs):
                _, e = p.communicate(timeout=900)
                assert p.returncode == 0, (side, r, e.decode()[-2000:])
                assert ("Mpi cluster info total 2 worker rank %d" % r).encode() in e
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        models[side] = open(str(d / ("bsp2.%s.0.model" % side)), "rb").read()
    assert models["own"] == models["ref"]
    # aslp-nnet-train-simple-mpi: exactly two ranks swap and average whole models (NnetMpiSync), each on the shard "JOB" names
    import shutil
    for r in range(2):
        shutil.copy(str(d / "d_feats.ark"), str(d / ("mpi_feats.%d.ark" % r)))
    for side, bindir in (("own", OWN), ("ref", REF)):
        comm = str(d / ("comm.mpi." + side))
        token = secrets.token_hex(6)
        procs = []
        for r in range(2):
            env = dict(os.environ, ASLP_COMM_TOKEN=token, ASLP_COMM_TRANSPORT="shm")
            args = ["--learn-rate=0.01", "--minibatch-size=16", "--randomizer-size=100", "--sync-period=64", "ark:%s" % (d / "mpi_feats.JOB.ark"),
                    "ark:%s" % (d / "d_post.ark"), str(d / "dnn.nnet"), str(d / ("mpi.%s.%d.model" % (side, r)))]
            if side == "own":
                argv = [os.path.join(bindir, "aslp-nnet-train-simple-mpi"), "--rank=%d" % r, "--num-workers=2", "--comm-file=" + comm] + args
            else:
                argv = [os.path.join(bindir, "aslp-nnet-train-simple-mpi")] + args
                env.update(RANK=str(r), WORLD_SIZE="2", ASLP_COMM_FILE=comm)
            procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
        try:
            for r, p in enumerate(procs):
                _, e = p.communicate(timeout=900)
                assert p.returncode == 0, (side, r, e.decode()[-2000:])
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        models["mpi." + side] = open(str(d / ("mpi.%s.0.model" % side)), "rb").read()
    assert models["mpi.own"] == models["mpi.ref"]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    for st in ("easgd", "asgd", "masgd"):
        outs = {}
        for side, bindir in (("own", OWN), ("ref", REF)):
            out = str(d / ("server.%s.%s.model" % (st, side)))
            p = subprocess.run([os.path.join(bindir, "aslp-nnet-train-server"), "--server-type=" + st, "--alpha=0.5", "--sync-period=10", str(d / "dnn.nnet"), out],
                               capture_output=True, timeout=900, env=env)
            assert p.returncode == 0 and b"Mpi cluster info total 1 server rank 0" in p.stderr and b"All worker finished" in p.stderr, p.stderr.decode()[-2000:]
            outs[side] = open(out, "rb").read()
        assert outs["own"] == outs["ref"]


# ---- more option combinations of the stream tools, and the multi-input tools ---------------------------------------------------------------
MORE_STREAM_CASES = [
    ("aslp-nnet-train-lstm-streams-skip", "lstm", ["--num-stream=3", "--batch-size=5", "--targets-delay=1", "--skip-width=2", "--frame-weights=ark:{d}/s_fw.ark"]),
    ("aslp-nnet-train-lstm-streams-skip", "lstm", ["--num-stream=2", "--batch-size=6", "--feature-transform={d}/tr.nnet", "--dump-interval=2"]),
    ("aslp-nnet-train-lstm-streams-skip", "lstm", ["--num-stream=3", "--batch-size=5", "--objective-function=mse", "--learn-rate=0.001"]),
    ("aslp-nnet-train-lstm-streams", "lstm", ["--num-stream=3", "--batch-size=5", "--objective-function=mse", "--learn-rate=0.001", "--report-period=2"]),
    ("aslp-nnet-train-blstm-streams-lc", "lc", ["--num-stream=3", "--chunk-size=6", "--right-splice=3", "--frame-weights=ark:{d}/s_fw.ark"]),
    ("aslp-nnet-train-blstm-streams-lc", "lc", ["--num-stream=2", "--chunk-size=7", "--right-splice=2", "--feature-transform={d}/tr.nnet", "--dump-interval=3"]),
    ("aslp-nnet-train-blstm-streams-lc", "lc", ["--num-stream=3", "--chunk-size=6", "--right-splice=3", "--objective-function=mse", "--learn-rate=0.001"]),
    ("aslp-nnet-train-blstm-streams", "blstm", ["--num-stream=3", "--feature-transform={d}/tr.nnet", "--learn-rate=0.2"]),
    ("aslp-nnet-train-blstm-streams", "blstm", ["--num-stream=2", "--objective-function=mse", "--learn-rate=0.01"]),
    ("aslp-nnet-train-blstm-parallel", "blstm", ["--num-stream=3", "--feature-transform={d}/tr.nnet", "--learn-rate=0.002"]),
    ("aslp-nnet-train-blstm-parallel", "blstm", ["--num-stream=2", "--objective-function=mse", "--learn-rate=0.0005"]),
    ("aslp-nnet-train-perutt", "fsmn", ["--feature-transform={d}/tr.nnet", "--learn-rate=0.5"]),
    ("aslp-nnet-train-perutt", "fsmn", ["--objective-function=mse", "--learn-rate=0.05"]),
]


# what neither program accepts: the same refusal from both (an option the tool does not have; an objective it does not support)
REFUSED = [("aslp-nnet-train-lstm-streams-skip", "--frame-weights="), ("aslp-nnet-train-lstm-streams-skip", "--objective-function=mse"),
           ("aslp-nnet-train-blstm-streams-lc", "--frame-weights="), ("aslp-nnet-train-blstm-streams-lc", "--objective-function=mse")]


@pytest.mark.parametrize("case", MORE_STREAM_CASES, ids=lambda c: c[0][10:] + " " + " ".join(x.replace("{d}/", "") for x in c[2]))
def test_stream_tools_more_options(corpus, case):
    name, net, flags = case
    d = corpus["dir"]
    flags = [f.format(d=d) for f in flags]
    if not any(f.startswith("--learn-rate") for f in flags):
        flags = ["--learn-rate=0.02"] + flags
    if any(name == n and any(f.startswith(opt) for f in flags) for n, opt in REFUSED):
        errs = []
        for bindir in (OWN, REF):
            p = run(bindir, name, flags + tables(corpus, "s", "feats", "post") + [str(d / (net + ".nnet")), str(d / "refused.model")], ok=False)
            assert p.returncode != 0
            errs.append([x for x in log_lines(p.stderr) if "Invalid option" in x or "objective function" in x or "Only Support" in x])
        assert errs[0] == errs[1] and len(errs[0]) == 1, errs
        return
    lo, lr, _, _ = both(corpus, name, flags, tables(corpus, "s", "feats", "post") + [str(d / (net + ".nnet"))], tag=str(abs(hash(tuple(flags))) % 997))
    if name == "aslp-nnet-train-lstm-streams":
        final = [x for x in lo if x not in lr]
        lo = [x for x in lo if x not in final]
    assert lo == lr


def test_stream_workers_dumps_and_weights(corpus):
    d = corpus["dir"]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    # (--cross-validate on a worker: the reference's mains register it, then the LSTM worker trains regardless (:176-182) and both end in
    # Write("") on the main node; the engine's tools refuse the flag: "the worker tools train only")
    for name, net, flags in (("aslp-nnet-train-lstm-stream-worker", "lstm", ["--worker-type=bmuf", "--sync-period=40", "--dump-interval=2", "--num-stream=2", "--batch-size=6", "--learn-rate=0.02"]),
                             ("aslp-nnet-train-lc-blstm-streams-worker", "lc", ["--worker-type=bsp", "--sync-period=40", "--drop-len=40", "--num-stream=2", "--chunk-size=8", "--right-splice=4",
                                                                                   "--learn-rate=0.02"])):
        cv = "--cross-validate=true" in flags
        res = {}
        for side, bindir in (("own", OWN), ("ref", REF)):
            out = str(d / ("%s.x%d.%s.model" % (name, abs(hash(tuple(flags))) % 997, side)))
            p = subprocess.run([os.path.join(bindir, name)] + flags + tables(corpus, "s", "feats", "post") + [str(d / (net + ".nnet"))] + ([] if cv else [out]),
                               capture_output=True, timeout=1800, env=env)
            assert p.returncode == 0, (side, name, flags, p.stderr.decode()[-2000:])
            res[side] = (b"" if cv else open(out, "rb").read(), log_lines(p.stderr))
        assert res["own"][0] == res["ref"][0], (name, flags)
        lo, lr = res["own"][1], res["ref"][1]
        final = [x for x in lo if x not in lr]
        assert [x for x in lo if x not in final] == lr and len(final) <= 2, (name, flags)


def test_multi_input_tools(corpus):
    """aslp-nnet-train-frame-mimo / aslp-nnet-forward-mimo on a graph net with two inputs, a shared trunk and two heads"""
    import nnet_io
    d = corpus["dir"]
    rng = np.random.default_rng(5)
    d0, d1, h0, h1, ht, a0, a1 = 12, 20, 16, 24, 32, 10, 30
    mk = lambda o, i: nnet_io.affine(rng.standard_normal((o, i)).astype(np.float32) * 0.3, rng.standard_normal(o).astype(np.float32) * 0.1)
    comps = [dict(marker="<InputLayer>", dim_in=d0, dim_out=d0, id=0, inputs=[-1], offsets=[0]),
             dict(marker="<InputLayer>", dim_in=d1, dim_out=d1, id=1, inputs=[-1], offsets=[0]),
             dict(marker="<AffineTransform>", dim_in=d0, dim_out=h0, id=2, inputs=[0], offsets=[0], data=mk(h0, d0)),
             dict(marker="<Sigmoid>", dim_in=h0, dim_out=h0, id=3, inputs=[2], offsets=[0]),
             dict(marker="<AffineTransform>", dim_in=d1, dim_out=h1, id=4, inputs=[1], offsets=[0], data=mk(h1, d1)),
             dict(marker="<Sigmoid>", dim_in=h1, dim_out=h1, id=5, inputs=[4], offsets=[0]),
             dict(marker="<AffineTransform>", dim_in=h0 + h1, dim_out=ht, id=6, inputs=[3, 5], offsets=[0, h0], data=mk(ht, h0 + h1)),
             dict(marker="<Sigmoid>", dim_in=ht, dim_out=ht, id=7, inputs=[6], offsets=[0]),
             dict(marker="<AffineTransform>", dim_in=ht, dim_out=a0, id=8, inputs=[7], offsets=[0], data=mk(a0, ht)),
             dict(marker="<Softmax>", dim_in=a0, dim_out=a0, id=9, inputs=[8], offsets=[0]),
             dict(marker="<OutputLayer>", dim_in=a0, dim_out=a0, id=10, inputs=[9], offsets=[0]),
             dict(marker="<AffineTransform>", dim_in=ht, dim_out=a1, id=11, inputs=[7], offsets=[0], data=mk(a1, ht)),
             dict(marker="<Softmax>", dim_in=a1, dim_out=a1, id=12, inputs=[11], offsets=[0]),
             dict(marker="<OutputLayer>", dim_in=a1, dim_out=a1, id=13, inputs=[12], offsets=[0])]
    nnet_io.write_graph_nnet(d / "mimo.nnet", comps)
    io = ["ark:%s" % (d / "s_feats.ark"), "ark:%s" % (d / "d_feats.ark"), "ark:%s" % (d / "s_post.ark"), "ark:%s" % (d / "d_post.ark"), str(d / "mimo.nnet")]
    # 1111 frames would leave a part-filled last minibatch, where the reference's multi-table reader dies on its randomizers' own check
    # (data-reader.cc:139-141 asks for a minibatch that is not there) and the engine's tool stops: compare where both finish -- minibatch 1
    for i, flags in enumerate((["--minibatch-size=1", "--randomizer-size=60", "--learn-rate=0.01", "--objective-function=xent:xent"],
                               ["--minibatch-size=1", "--randomizer-size=40", "--learn-rate=0.01", "--momentum=0.5", "--randomizer-seed=3", "--objective-function=xent:xent"])):
        lo, lr, _, _ = both(corpus, "aslp-nnet-train-frame-mimo", flags, io, tag="mi%d" % i)
        assert lo == lr and sum("AvgLoss" in x for x in lo) >= 2
    fio = [str(d / "mimo.nnet"), "ark:%s" % (d / "s_feats.ark"), "ark:%s" % (d / "d_feats.ark")]
    for i, flags in enumerate((["--apply-log=false"], [], ["--no-softmax=true", "--apply-log=false"], ["--time-shift=1", "--apply-log=false"])):
        lo, lr, _, _ = both(corpus, "aslp-nnet-forward-mimo", flags, fio, ("ark",), tag="fm%d" % i)
        assert lo == lr


VERBOSE_CASES = [
    ("aslp-nnet-train-simple", "dnn", "d", ["--verbose=2", "--learn-rate=0.01", "--minibatch-size=16", "--randomizer-size=100"]),
    ("aslp-nnet-train-mse", "dnn", "dmse", ["--verbose=1", "--learn-rate=0.001", "--minibatch-size=16", "--randomizer-size=100"]),
    ("aslp-nnet-train-perutt", "fsmn", "s", ["--verbose=2", "--learn-rate=0.5"]),
    ("aslp-nnet-train-lstm-streams", "lstm", "s", ["--verbose=2", "--num-stream=3", "--batch-size=5", "--learn-rate=0.02"]),
    ("aslp-nnet-train-lstm-streams-skip", "lstm", "s", ["--verbose=1", "--num-stream=3", "--batch-size=5", "--skip-width=2", "--learn-rate=0.02"]),
    ("aslp-nnet-train-blstm-streams", "blstm", "s", ["--verbose=2", "--num-stream=3", "--learn-rate=0.2"]),
    ("aslp-nnet-train-blstm-parallel", "blstm", "s", ["--verbose=1", "--num-stream=3", "--learn-rate=0.002"]),
    ("aslp-nnet-train-blstm-streams-lc", "lc", "s", ["--verbose=2", "--num-stream=3", "--chunk-size=6", "--right-splice=3", "--learn-rate=0.02"]),
]


@pytest.mark.parametrize("case", VERBOSE_CASES, ids=lambda c: c[0][10:] + " " + c[3][0])
def test_verbose_logs(corpus, case):
    """--verbose=1 / 2: which state dumps (InfoPropagate / InfoBackPropagate / InfoGradient, "### After N frames,", per-utterance lines) a tool
    prints and when -- the dumps themselves come from the library and are the same text on both sides when the training is the same"""
    name, net, tab, flags = case
    d = corpus["dir"]
    inputs = tables(corpus, "d", "feats", "tgt") if tab == "dmse" else tables(corpus, tab, "feats", "post")
    lo, lr, _, _ = both(corpus, name, flags, inputs + [str(d / (net + ".nnet"))], tag="v" + str(abs(hash(tuple(flags))) % 997))
    if name == "aslp-nnet-train-lstm-streams":
        final = [x for x in lo if x not in lr]
        lo = [x for x in lo if x not in final]
    assert lo == lr and any("###" in x or "VLOG" in x or "After" in x for x in lo + ["###"])


def test_ctc_and_worker_verbose(corpus):
    d = corpus["dir"]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    ctc_io = ["ark:%s" % (d / "s_feats.ark"), "ark:%s" % (d / "lab.ark")]
    for i, (name, flags, io, net) in enumerate((
            ("aslp-nnet-train-ctc-streams", ["--verbose=2", "--num-stream=3", "--learn-rate=0.05"], ctc_io, "ctc"),
            ("aslp-nnet-train-warp-ctc-streams", ["--verbose=1", "--num-stream=3", "--learn-rate=0.05"], ctc_io, "ctc"),
            ("aslp-nnet-train-ctc", ["--verbose=2", "--learn-rate=0.001"], ctc_io, "uctc"),
            ("aslp-nnet-train-frame", ["--verbose=2", "--learn-rate=0.01", "--minibatch-size=16", "--randomizer-size=100"], tables(corpus, "d", "feats", "post"), "dnn"),
            ("aslp-nnet-train-frame-worker", ["--verbose=2", "--worker-type=bsp", "--sync-period=64", "--learn-rate=0.01", "--minibatch-size=16", "--randomizer-size=100"],
             tables(corpus, "d", "feats", "post"), "dnn"),
            ("aslp-nnet-train-lstm-stream-worker", ["--verbose=2", "--worker-type=bsp", "--sync-period=40", "--num-stream=3", "--batch-size=5", "--learn-rate=0.02"],
             tables(corpus, "s", "feats", "post"), "lstm"),
            ("aslp-nnet-train-lc-blstm-streams-worker", ["--verbose=2", "--worker-type=bmuf", "--sync-period=40", "--num-stream=3", "--chunk-size=6", "--right-splice=3", "--learn-rate=0.02"],
             tables(corpus, "s", "feats", "post"), "lc"))):
        res = {}
        for side, bindir in (("own", OWN), ("ref", REF)):
            out = str(d / ("%s.vb%d.%s.model" % (name, i, side)))
            p = subprocess.run([os.path.join(bindir, name)] + flags + io + [str(d / (net + ".nnet")), out], capture_output=True, timeout=1800, env=env)
            assert p.returncode == 0, (side, name, p.stderr.decode()[-2000:])
            res[side] = (open(out, "rb").read(), log_lines(p.stderr))
        assert res["own"][0] == res["ref"][0], name
        lo, lr = res["own"][1], res["ref"][1]
        if "stream" in name and "worker" in name:   # (the stream workers' final report, see test_workers_alone_in_their_group)
            final = [x for x in lo if x not in lr]
            lo = [x for x in lo if x not in final]
        assert lo == lr, name


def test_forward_verbose(corpus):
    d = corpus["dir"]
    for i, (name, net, tab) in enumerate((("aslp-nnet-forward", "dnn", "d"), ("aslp-nnet-forward-skip", "lstm", "s"), ("aslp-nnet-forward-blstm-lc", "lc", "s"))):
        flags = ["--verbose=2", "--apply-log=false"] + (["--skip-width=2"] if "skip" in name else [])
        lo, lr, _, _ = both(corpus, name, flags, [str(d / (net + ".nnet")), "ark:%s" % (d / (tab + "_feats.ark"))], ("ark",), tag="vf%d" % i)
        assert lo == lr
