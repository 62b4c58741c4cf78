"""Independent Python writers / readers of the Kaldi stream formats, written from the format description only
(src/base/io-funcs-inl.h, matrix/kaldi-matrix.cc:1201-1330, compressed-matrix.cc, hmm/posterior.cc:29-60,
util/kaldi-holder-inl.h:191-288, util/kaldi-table-inl.h:358-403).  TEST INFRASTRUCTURE: the C++ table layer is
checked against files produced here and this module parses what the C++ layer writes."""
import struct

import numpy as np


def w_int32(v):
    return b"\x04" + struct.pack("<i", int(v))


def w_float(v):
    return b"\x04" + struct.pack("<f", float(v))


def matrix_bin(m, double=False):
    m = np.ascontiguousarray(m, dtype=np.float64 if double else np.float32)
    return b"\0B" + (b"DM " if double else b"FM ") + w_int32(m.shape[0]) + w_int32(m.shape[1]) + m.tobytes()


def vector_bin(v, double=False):
    v = np.ascontiguousarray(v, dtype=np.float64 if double else np.float32)
    return b"\0B" + (b"DV " if double else b"FV ") + w_int32(v.shape[0]) + v.tobytes()


def matrix_txt(m):
    if m.shape[1] == 0:
        return b" [ ]\n"
    s = " ["
    for r in m:
        s += "\n  " + "".join("%.9g " % x for x in r)
    return (s + "]\n").encode()


def posterior_bin(post):
    out = b"\0B" + w_int32(len(post))
    for frame in post:
        out += w_int32(len(frame))
        for i, p in frame:
            out += w_int32(i) + w_float(p)
    return out


def posterior_txt(post):
    return ("".join("[ " + "".join("%d %.9g " % (i, p) for i, p in fr) + "] " for fr in post) + "\n").encode()


def int32vec_bin(v):
    return b"\0B" + w_int32(len(v)) + b"".join(w_int32(x) for x in v)


def int32vec_txt(v):
    return ("".join("%d " % x for x in v) + "\n").encode()


def compressed_cm(min_value, rng, headers, data):
    """format 1 ("CM"): headers uint16 [cols, 4], data uint8 [cols, rows] (column-major bytes)"""
    cols, rows = data.shape
    return (b"\0BCM " + struct.pack("<ffii", min_value, rng, rows, cols) + np.ascontiguousarray(headers, np.uint16).tobytes() +
            np.ascontiguousarray(data, np.uint8).tobytes())


def compressed_cm2(min_value, rng, data):
    rows, cols = data.shape
    return b"\0BCM2 " + struct.pack("<ffii", min_value, rng, rows, cols) + np.ascontiguousarray(data, np.uint16).tobytes()


def decode_cm(min_value, rng, headers, data):
    f32 = np.float32
    u16 = lambda v: f32(f32(min_value) + f32(f32(f32(rng) * f32(1.52590218966964e-05)) * f32(v)))
    cols, rows = data.shape
    out = np.zeros((rows, cols), np.float32)
    for c in range(cols):
        p0, p25, p75, p100 = (u16(int(h)) for h in headers[c])
        for r in range(rows):
            v = int(data[c, r])
            if v <= 64:
                f = np.float64(p0) + np.float64(f32(f32(p25 - p0) * f32(v))) * (1 / 64.0)
            elif v <= 192:
                f = np.float64(p25) + np.float64(f32(f32(p75 - p25) * f32(v - 64))) * (1 / 128.0)
            else:
                f = np.float64(p75) + np.float64(f32(f32(p100 - p75) * f32(v - 192))) * (1 / 63.0)
            out[r, c] = f32(f)
    return out


def decode_cm2(min_value, rng, data):
    f32 = np.float32
    return (f32(min_value) + (f32(rng) * f32(1.52590218966964e-05)).astype(f32) * data.astype(f32)).astype(f32)


def archive(entries):
    """entries: list of (key, object-bytes)"""
    return b"".join(k.encode() + b" " + o for k, o in entries)


# ---- parsers of what the C++ layer writes -------------------------------------------------------------------
class _Cur:
    def __init__(self, b):
        self.b, self.p = b, 0

    def take(self, n):
        out = self.b[self.p:self.p + n]
        assert len(out) == n, "truncated"
        self.p += n
        return out

    def token(self):
        e = self.b.index(b" ", self.p)
        t = self.b[self.p:e]
        self.p = e + 1
        return t.decode()

    def int32(self):
        assert self.take(1) == b"\x04"
        return struct.unpack("<i", self.take(4))[0]

    def float32(self):
        assert self.take(1) == b"\x04"
        return struct.unpack("<f", self.take(4))[0]


def parse_bin_archive(b, kind):
    """-> list of (key, object) for kind in matrix|vector|posterior|int32-vector"""
    c, out = _Cur(b), []
    while c.p < len(b):
        key = c.token()
        assert c.take(2) == b"\0B", key
        if kind == "matrix":
            assert c.token() == "FM"
            r, k = c.int32(), c.int32()
            out.append((key, np.frombuffer(c.take(4 * r * k), np.float32).reshape(r, k)))
        elif kind == "vector":
            assert c.token() == "FV"
            n = c.int32()
            out.append((key, np.frombuffer(c.take(4 * n), np.float32)))
        elif kind == "posterior":
            post = []
            for _ in range(c.int32()):
                post.append([(c.int32(), c.float32()) for _ in range(c.int32())])
            out.append((key, post))
        else:
            out.append((key, [c.int32() for _ in range(c.int32())]))
    return out
