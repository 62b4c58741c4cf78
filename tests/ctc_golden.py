"""Reader for tests/golden/ctc_*.bin (written by oracle/gen_ctc_golden.cpp from the reference)."""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["small", "inf", "grad_a20_t50_l15", "grad_a5_t10_l5_mb65", "ragged", "a128_t200"]


def load(name):
    raw = open(os.path.join(GOLDEN, "ctc_%s.bin" % name), "rb").read()
    A, mb, maxT, nl = np.frombuffer(raw, "<i4", 4)
    o = 16
    def take(dtype, n):
        nonlocal o
        a = np.frombuffer(raw, dtype, n, o).copy()
        o += a.nbytes
        return a
    d = dict(A=int(A), mb=int(mb), maxT=int(maxT))
    d["input_lengths"] = take("<i4", mb)
    d["label_lengths"] = take("<i4", mb)
    d["flat_labels"] = take("<i4", nl)
    d["acts"] = take("<f4", maxT * mb * A)
    d["costs"] = take("<f4", mb)
    d["grads"] = take("<f4", maxT * mb * A)
    assert o == len(raw)
    return d
