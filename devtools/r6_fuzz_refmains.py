"""Exploratory fuzz of the engine's tools against the reference's own mains (kaldi-aslp_amd/bin_ref/): random option combinations per tool on the
small tables of tests/test_ref_mains_diff_gpu.py, models byte for byte and whole logs line for line.  Prints every mismatch with the command
that produced it (a mismatch becomes a regression case in tests/test_ref_mains_diff_gpu.py).   python3 devtools/r6_fuzz_refmains.py [seed] [cases per tool]"""
import os
import random
import subprocess
import sys
import tempfile
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_ref_mains_diff_gpu as T   # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
per_tool = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rnd = random.Random(seed)


corpus = T.build_corpus(pathlib.Path(tempfile.mkdtemp(prefix="refdiff_fuzz")))
d = corpus["dir"]


def pick(opts):
    out = []
    for name, values in opts:
        if rnd.random() < 0.55:
            v = rnd.choice(values)
            out.append("--%s=%s" % (name, v))
    return out


TRN = [("learn-rate", ["0.01", "0.002", "0.05"]), ("momentum", ["0", "0.5", "0.9"]), ("l2-penalty", ["0", "1e-4"]), ("l1-penalty", ["0", "1e-6"])]
RND = [("minibatch-size", ["8", "16", "32", "50"]), ("randomizer-size", ["40", "100", "300", "32768"]), ("randomizer-seed", ["1", "777", "12345"])]
COMMON = [("binary", ["true", "false"]), ("report-period", ["1", "3", "50", "200"]), ("verbose", ["0", "1", "2"])]
SEQ = [("num-stream", ["1", "2", "3", "5"]), ("batch-size", ["3", "5", "8", "20"]), ("targets-delay", ["0", "1", "3", "5"]), ("drop-len", ["0", "25", "40", "100"])]
TOOLS = {
    "aslp-nnet-train-frame": ("dnn", "d", "post", TRN + RND + COMMON + [("cross-validate", ["false", "true"]), ("randomize", ["true", "false"]), ("dropout-retention", ["0", "1.0"])]),
    "aslp-nnet-train-simple": ("dnn", "d", "post", TRN + RND + COMMON + [("cross-validate", ["false", "true"]), ("randomize", ["true", "false"]), ("length-tolerance", ["0", "1", "5"]),
                                                                        ("frame-weights", ["ark:%s/d_fw.ark" % d]), ("utt-weights", ["ark:%s/uw.ark" % d]),
                                                                        ("objective-function", ["xent", "mse"])]),
    "aslp-nnet-train-mse": ("dnn", "d", "tgt", TRN + RND + COMMON + [("cross-validate", ["false", "true"]), ("randomize", ["true", "false"]), ("length-tolerance", ["0", "1", "5"]),
                                                                    ("frame-weights", ["ark:%s/d_fw.ark" % d]), ("utt-weights", ["ark:%s/uw.ark" % d])]),
    "aslp-nnet-train-perutt": ("fsmn", "s", "post", TRN + COMMON + [("cross-validate", ["false", "true"]), ("length-tolerance", ["0", "1", "5"]), ("drop-len", ["0", "25", "40"]),
                                                                   ("frame-weights", ["ark:%s/s_fw.ark" % d]), ("feature-transform", [str(d / "tr.nnet")]), ("objective-function", ["xent", "mse"])]),
    "aslp-nnet-train-lstm-streams": ("lstm", "s", "post", TRN + COMMON + SEQ + [("cross-validate", ["false", "true"]), ("objective-function", ["xent", "mse"])]),
    "aslp-nnet-train-lstm-streams-skip": ("lstm", "s", "post", TRN + COMMON + SEQ + [("cross-validate", ["false", "true"]), ("skip-width", ["1", "2", "3"]), ("dump-interval", ["0", "2"]),
                                                                                    ("feature-transform", [str(d / "tr.nnet")]), ("length-tolerance", ["0", "5"])]),
    "aslp-nnet-train-blstm-streams": ("blstm", "s", "post", TRN + COMMON + [("num-stream", ["1", "2", "3", "5"]), ("frame-limit", ["50", "100", "100000"]), ("drop-len", ["0", "25", "40"]),
                                                                           ("skip-width", ["1", "2", "3"]), ("cross-validate", ["false", "true"]), ("length-tolerance", ["0", "1", "5"]),
                                                                           ("frame-weights", ["ark:%s/s_fw.ark" % d]), ("feature-transform", [str(d / "tr.nnet")]), ("objective-function", ["xent", "mse"])]),
    "aslp-nnet-train-blstm-parallel": ("blstm", "s", "post", TRN + COMMON + [("num-stream", ["1", "2", "3", "5"]), ("frame-limit", ["50", "100", "100000"]), ("drop-len", ["0", "25", "40"]),
                                                                            ("cross-validate", ["false", "true"]), ("length-tolerance", ["0", "1", "5"]),
                                                                            ("frame-weights", ["ark:%s/s_fw.ark" % d]), ("feature-transform", [str(d / "tr.nnet")]), ("objective-function", ["xent", "mse"])]),
    "aslp-nnet-train-blstm-streams-lc": ("lc", "s", "post", TRN + COMMON + [("num-stream", ["1", "2", "3", "5"]), ("chunk-size", ["4", "6", "10", "64"]), ("right-splice", ["0", "2", "5", "16"]),
                                                                           ("drop-len", ["0", "25", "40"]), ("cross-validate", ["false", "true"]), ("dump-interval", ["0", "3"]),
                                                                           ("feature-transform", [str(d / "tr.nnet")])]),
    "aslp-nnet-train-ctc-streams": ("ctc", "s", "lab", TRN + COMMON + [("num-stream", ["1", "2", "3", "5"]), ("frame-limit", ["50", "100", "100000"]), ("drop-len", ["0", "25", "40"]),
                                                                      ("skip-width", ["0", "1", "2"]), ("cross-validate", ["false", "true"]), ("report-step", ["1", "2", "100"])]),
    "aslp-nnet-train-warp-ctc-streams": ("ctc", "s", "lab", TRN + COMMON + [("num-stream", ["1", "2", "3", "5"]), ("frame-limit", ["50", "100", "100000"]), ("drop-len", ["0", "25", "40"]),
                                                                           ("skip-width", ["0", "1", "2"]), ("cross-validate", ["false", "true"]), ("report-step", ["1", "2", "100"])]),
    "aslp-nnet-train-ctc": ("uctc", "s", "lab", TRN + COMMON + [("drop-len", ["0", "25", "40"]), ("cross-validate", ["false", "true"]), ("report-step", ["1", "2", "100"]),
                                                               ("token-symbol-table", [str(d / "tokens.txt")])]),
    "aslp-nnet-train-frame-worker": ("dnn", "d", "post", TRN + RND + COMMON + [("worker-type", ["bsp", "bmuf", "sod"]), ("sync-period", ["30", "64", "200", "25600"]),
                                                                             ("bmuf-learn-rate", ["1.0", "0.8"]), ("bmuf-momentum", ["0.0", "0.9"]), ("solver", ["sgd", "momentum", "adagrad", "rmsprop", "adadelta", "adam"]),
                                                                             ("lr", ["0.01", "0.5"]), ("dropout-retention", ["0", "1.0"])]),
    "aslp-nnet-train-lstm-stream-worker": ("lstm", "s", "post", TRN + COMMON + SEQ + [("worker-type", ["bsp", "bmuf", "sod"]), ("sync-period", ["30", "64", "25600"]), ("dump-interval", ["0", "2"]),
                                                                                     ("solver", ["sgd", "adam"]), ("objective-function", ["xent", "mse"])]),
    "aslp-nnet-train-lc-blstm-streams-worker": ("lc", "s", "post", TRN + COMMON + [("num-stream", ["1", "2", "3"]), ("chunk-size", ["4", "6", "10"]), ("right-splice", ["0", "2", "5"]),
                                                                                  ("drop-len", ["0", "25", "40"]), ("worker-type", ["bsp", "bmuf"]), ("sync-period", ["30", "64", "25600"]),
                                                                                  ("dump-interval", ["0", "3"]), ("feature-transform", [str(d / "tr.nnet")])]),
    "aslp-nnet-forward": ("dnn", "d", None, [("apply-log", ["true", "false"]), ("no-softmax", ["true", "false"]), ("class-frame-counts", [str(d / "counts")]), ("prior-scale", ["1.0", "0.5"]),
                                             ("prior-floor", ["1e-10", "1e-3"]), ("time-shift", ["0", "1", "3"]), ("skip-width", ["0", "1", "2", "4"]), ("verbose", ["0", "2"]),
                                             ("add-softmax", ["false", "true"]), ("scale-blank", ["1.0", "0.5"])]),
    "aslp-nnet-forward-skip": ("lstm", "s", None, [("apply-log", ["true", "false"]), ("no-softmax", ["true", "false"]), ("class-frame-counts", [str(d / "counts10")]), ("time-shift", ["0", "1"]),
                                                   ("skip-width", ["1", "2", "3"]), ("add-softmax", ["false", "true"]), ("scale-blank", ["1.0", "2.0"])]),
    "aslp-nnet-forward-blstm-lc": ("lc", "s", None, [("apply-log", ["true", "false"]), ("no-softmax", ["true", "false"]), ("class-frame-counts", [str(d / "counts10")]),
                                                     ("chunk-size", ["4", "10", "40", "64"]), ("right-splice", ["0", "3", "16"])]),
}
bad = 0
total = 0
for name, (net, tab, tgt, opts) in TOOLS.items():
    for k in range(per_tool):
        flags = pick(opts)
        cv = "--cross-validate=true" in flags
        fwd = name.startswith("aslp-nnet-forward")
        if fwd:
            inputs = [str(d / (net + ".nnet")), "ark:%s" % (d / (tab + "_feats.ark"))]
            flags = ["--use-gpu=yes"] + flags
        else:
            tgt_spec = "ark:%s" % (d / "lab.ark") if tgt == "lab" else "ark:%s" % (d / ("%s_%s.ark" % (tab, tgt)))
            inputs = ["ark:%s" % (d / (tab + "_feats.ark")), tgt_spec, str(d / (net + ".nnet"))]
        res = {}
        for side, bindir in (("own", T.OWN), ("ref", T.REF)):
            out = str(d / ("fz.%s.%d.%s" % (name, k, side)))
            if os.path.exists(out):
                os.remove(out)
            outs = [] if cv else (["ark:" + out] if fwd else [out])
            try:
                p = subprocess.run([os.path.join(bindir, name)] + flags + inputs + outs, capture_output=True, timeout=90, env=dict(os.environ, RANK="0", WORLD_SIZE="1"))
            except subprocess.TimeoutExpired:
                print("HANG (90 s) %s/%s %s" % (os.path.basename(bindir), name, " ".join(flags)), flush=True)
                res[side] = (-999, None, [])
                continue
            data = open(out, "rb").read() if os.path.exists(out) else None
            res[side] = (p.returncode, data, T.log_lines(p.stderr))
        total += 1
        (ro, do, lo), (rr, dr, lr) = res["own"], res["ref"]
        same_rc = (ro == 0) == (rr == 0)
        if name in ("aslp-nnet-train-lstm-streams", "aslp-nnet-train-lstm-stream-worker", "aslp-nnet-train-lc-blstm-streams-worker") and ro == 0:   # (its final report: logged by the engine's tool, formed and dropped by the reference's main)
            extra = [x for x in lo if x not in lr]
            if len(extra) <= 2 and all("AvgLoss" in x or "FRAME_ACCURACY" in x or x == "" for x in extra):
                lo = [x for x in lo if x not in extra]
        if name == "aslp-nnet-forward-skip" and not any(f.startswith("--skip-width") for f in flags):
            continue   # (--skip-width=0, the default: the reference's main writes empty matrices, the engine's tool refuses)
        if not same_rc or do != dr or (ro == 0 and lo != lr):
            bad += 1
            print("MISMATCH %s %s  rc %s/%s  model %s  log %s" % (name, " ".join(flags), ro, rr, "same" if do == dr else "DIFFERS", "same" if lo == lr else "DIFFERS"), flush=True)
            if lo != lr:
                for a, b in zip(lo + ["<end>"] * 3, lr + ["<end>"] * 3):
                    if a != b:
                        print("    own: %s\n    ref: %s" % (a[:200], b[:200]))
                        break
print("fuzz seed %d: %d cases, %d mismatches" % (seed, total, bad))
