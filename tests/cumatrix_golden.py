"""Loader of tests/golden/cumatrix_ops.bin: outputs of the REFERENCE's own CuMatrix CPU branch (generator
oracle/gen_cumatrix_golden.cpp, built by `make -C oracle ref` from the reference sources where they lie)."""
import os
import struct

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cumatrix_ops.bin")


def load():
    b = open(PATH, "rb").read()
    out, p = {}, 0
    while p < len(b):
        name = b[p:p + 32].split(b"\0")[0].decode()
        r, c, kind = struct.unpack("<iii", b[p + 32:p + 44])
        p += 44
        a = np.frombuffer(b[p:p + 4 * r * c], np.int32 if kind else np.float32).reshape(r, c).copy()
        p += 4 * r * c
        out[name] = a[0] if kind else a
    return out
