"""orc_multitask_eval (oracle/aslp_oracle.c; MultiTaskLoss::Eval, nnet-loss.cc:341-368) against the formulas written out in float64:
task i sees only its own column block of the output and of the dense targets, its diff is scaled by the task weight, the statistics are
those of Xent::Eval (nnet-loss.cc:63-122) / Mse::Eval (:205-236) on that block."""
import numpy as np


def test_multitask_blocks_match_float64_formulas(oracle):
    rng = np.random.default_rng(5)
    rows, spec = 23, [("xent", 7, 1.0), ("mse", 5, 0.25), ("xent", 4, 0.5)]
    width = sum(d for _, d, _ in spec)
    y = rng.uniform(0.02, 0.98, (rows, width)).astype(np.float32)
    tgt = np.zeros((rows, width), np.float32)
    off = 0
    for kind, d, _ in spec:
        for r in range(rows):
            if kind == "xent":
                if r % 4 == 1:
                    a, b = rng.choice(d, 2, replace=False)
                    tgt[r, off + a], tgt[r, off + b] = 0.75, 0.25
                else:
                    tgt[r, off + rng.integers(0, d)] = 1.0
            else:
                tgt[r, off:off + d] = rng.uniform(-1, 1, d)
        off += d
    fw = rng.choice(np.array([0.0, 1.0, 2.0], np.float32), rows)
    diff, st = oracle.multitask_eval(spec, fw, y, tgt)
    y64, t64, w64 = y.astype(np.float64), tgt.astype(np.float64), fw.astype(np.float64)
    off = 0
    for i, (kind, d, wt) in enumerate(spec):
        yb, tb = y64[:, off:off + d], t64[:, off:off + d]
        if kind == "xent":
            w = w64 * tb.sum(1)
            want = (yb - tb) * w[:, None] * wt
            assert abs(st[i]["frames"] - w.sum()) < 1e-5
            assert abs(st[i]["loss"] + (np.log(yb + 1e-20) * tb * w[:, None]).sum()) < 1e-4
            assert abs(st[i]["entropy"] + (np.log(tb + 1e-20) * tb * w[:, None]).sum()) < 1e-4
            assert abs(st[i]["likelyhood"] - (yb * tb * w[:, None]).sum()) < 1e-4
            assert abs(st[i]["correct"] - (w * (yb.argmax(1) == tb.argmax(1))).sum()) < 1e-5
        else:
            d0 = (yb - tb) * w64[:, None]
            want = d0 * wt
            assert abs(st[i]["loss"] - 0.5 * (d0 * d0 * w64[:, None]).sum()) < 1e-4
            assert st[i]["frames"] == float(int(w64.sum()))
        assert np.max(np.abs(diff[:, off:off + d] - want)) < 1e-6
        off += d
