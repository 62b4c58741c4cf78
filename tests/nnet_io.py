"""Writes nnet model files in the reference's BINARY format from numpy arrays, so tests can hand
identical weights to the HIP engine and to the oracle.  Format: src/aslp-nnet/nnet-component.cc:
328-342 (component header), base/io-funcs-inl.h (basic types), matrix/kaldi-matrix.cc:1201-1223."""
import struct

import numpy as np


def tok(s): return s.encode() + b" "
def i32(v): return b"\x04" + struct.pack("<i", int(v))
def f32(v): return b"\x04" + struct.pack("<f", float(v))
def f64(v): return b"\x08" + struct.pack("<d", float(v))
def ivec(v): return b"\x04" + struct.pack("<i", len(v)) + np.asarray(v, "<i4").tobytes()
def fmat(m):
    m = np.ascontiguousarray(m, "<f4"); return tok("FM") + i32(m.shape[0]) + i32(m.shape[1]) + m.tobytes()
def fvec(v):
    v = np.ascontiguousarray(v, "<f4"); return tok("FV") + i32(v.shape[0]) + v.tobytes()
def dvec(v):
    v = np.ascontiguousarray(v, "<f8"); return tok("DV") + i32(v.shape[0]) + v.tobytes()


def header(marker, dim_out, dim_in, cid, inputs, offsets, name=None):
    b = tok(marker) + i32(dim_out) + i32(dim_in)
    if name:
        b += tok("<Name>") + tok(name)
    return b + i32(cid) + ivec(inputs) + ivec(offsets)


def affine(W, b, lr_coef=1.0, bias_lr_coef=1.0, max_norm=0.0):
    return (tok("<LearnRateCoef>") + f32(lr_coef) + tok("<BiasLearnRateCoef>") + f32(bias_lr_coef) +
            tok("<MaxNorm>") + f32(max_norm) + fmat(W) + fvec(b))


def batchnorm(shift, scale, num_acc=0.0, acc_means=None, acc_vars=None):
    d = len(shift)
    return (tok("<NumAccFrames>") + f64(num_acc) + dvec(acc_means if acc_means is not None else np.zeros(d)) +
            dvec(acc_vars if acc_vars is not None else np.zeros(d)) + fvec(shift) + fvec(scale))


def write_simple_nnet(path, layers):
    """layers: list of (marker, dim_in, dim_out, data_bytes). Adds InputLayer/OutputLayer with the ids and
    mono-inputs Nnet::AutoComplete assigns (nnet-nnet.cc:541-568)."""
    out = b"\x00B" + tok("<Nnet>")
    din = layers[0][1]
    out += header("<InputLayer>", din, din, 0, [-1], [0])
    for i, (marker, di, do, data) in enumerate(layers):
        out += header(marker, do, di, i + 1, [i], [0]) + data
    dlast = layers[-1][2]
    out += header("<OutputLayer>", dlast, dlast, len(layers) + 1, [len(layers)], [0])
    out += tok("</Nnet>")
    with open(path, "wb") as f:
        f.write(out)


def write_graph_nnet(path, comps):
    """comps: list of dicts(marker, dim_in, dim_out, id, inputs, offsets, data, name=None), in id order."""
    out = b"\x00B" + tok("<Nnet>")
    for c in comps:
        out += header(c["marker"], c["dim_out"], c["dim_in"], c["id"], c["inputs"], c["offsets"], c.get("name")) + c.get("data", b"")
    out += tok("</Nnet>")
    with open(path, "wb") as f:
        f.write(out)


def tensors(ts):
    """matrices as FM, vectors as FV, in the given order"""
    return b"".join(fmat(t) if t.ndim == 2 else fvec(t) for t in ts)


def lstm(dirs, clip, cell_dim=None):
    """LSTM-family payload: [<CellDim> n] <ClipGradient> f, then each direction's tensors
    (e.g. nnet-blstm-projected-streams-lc.h:242-271)"""
    b = b""
    if cell_dim is not None:
        b += tok("<CellDim>") + i32(cell_dim)
    b += tok("<ClipGradient>") + f32(clip)
    for d in dirs:
        b += tensors(d.tensors())
    return b


def gru(g, clip):
    return tok("<ClipGradient>") + f32(clip) + tensors(g.tensors())


def rowconv(w):
    return tok("<FutureContext>") + i32(w.shape[1] - 1) + fmat(w)


def fsmn(coef, past, future, lr_coef=1.0):
    return tok("<PastContext>") + i32(past) + tok("<FutureContext>") + i32(future) + tok("<LearnRateCoef>") + f32(lr_coef) + fmat(coef)
