"""Loader of tests/golden/cumatrix_ops.bin: outputs of the REFERENCE's own CuMatrix CPU branch (generator
oracle/gen_cumatrix_golden.cpp, built by `make -C oracle ref` from the reference sources where they lie)."""
import os
import struct

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cumatrix_ops.bin")


BLAS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cumatrix_blas_ops.bin")


def load_blas():
    """tests/golden/cumatrix_blas_ops.bin: the same CPU branch linked against the OpenBLAS inside the image's scipy wheel (generator
    oracle/gen_cumatrix_blas_golden.cpp): BLAS-backed operations and the op sequences of AffineTransform, BatchNormalization and
    LstmProjectedStreams issued against the reference's library.  kind 2 = float64; one-row records are vectors."""
    return {k: (v[0] if v.ndim == 2 and v.shape[0] == 1 else v) for k, v in load(BLAS_PATH).items()}


FULLWIDTH_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lstm_fullwidth.bin")
DIGEST_STRIDE = 61


def load_fullwidth():
    """tests/golden/lstm_fullwidth.bin (`oracle/_ref/gen_cumatrix_blas_golden <file> lstm_fullwidth`): two training steps of one projected-LSTM layer
    at BASELINE cfg3's widths on the reference's library, as a DIGEST -- a matrix record `name` holds every 61st element of the row-major
    image, `name#` (float64) = {sum, sum of squares, element count}; initial parameters, inputs and out-diffs are not stored: `lcfull_rng` is
    the generator state in front of them (oracle_lib.GoldenRng replays it).  Returns (records, generator state)."""
    g = {k: (v[0] if v.ndim == 2 and v.shape[0] == 1 else v) for k, v in load(FULLWIDTH_PATH).items()}
    rng = g["lcfull_rng"].astype(np.int64)
    state = (int(rng[0]) & 0xFFFFFFFF) | ((int(rng[1]) & 0xFFFFFFFF) << 32)
    return {k[7:]: v for k, v in g.items() if k.startswith("lcfull_")}, state


DIR_STRIDE = 257


def load_fullwidth_directions():
    """the records behind `lcdir_rng` in the same file: the two directions of BLstmProjectedStreamsLC at cfg3's widths -- `lcff` forward in time from a
    carried state, `lcfb` backward in time from zero (the small `lcf` / `lcb` records of cumatrix_blas_ops.bin at full width), digest stride 257.
    Returns ({tag: records}, generator state in front of `lcff`)."""
    g = {k: (v[0] if v.ndim == 2 and v.shape[0] == 1 else v) for k, v in load(FULLWIDTH_PATH).items()}
    rng = g["lcdir_rng"].astype(np.int64)
    state = (int(rng[0]) & 0xFFFFFFFF) | ((int(rng[1]) & 0xFFFFFFFF) << 32)
    return {tag: {k[len(tag) + 1:]: v for k, v in g.items() if k.startswith(tag + "_")} for tag in ("lcff", "lcfb")}, state


def digest_of(a, stride=DIGEST_STRIDE):
    """what the generators' digest mode keeps of a tensor: (every stride-th element, [sum, sum of squares, count])"""
    f = np.asarray(a, np.float32).ravel()
    return f[::stride], np.array([f.astype(np.float64).sum(), (f.astype(np.float64) ** 2).sum(), f.size])


TEMPORAL_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "temporal_fullsize.bin")


def load_temporal_fullsize(rng_factory):
    """tests/golden/temporal_fullsize.bin (`oracle/_ref/gen_cumatrix_blas_golden <file> temporal_fullsize`): RowConvolution (512 wide, FutureContext
    20, T = 800, S = 32 ragged streams) and CompactFsmn (512 wide, 30 + 30 taps, T = 800) on the reference's library, digest stride 257.
    rng_factory(state) -> oracle_lib.GoldenRng.  Returns (rc records, fsmn records, replayed tensors) in the generator's order: 31 length
    draws, then w / in / od of RowConvolution, then coef / in / od of CompactFsmn."""
    g = {k: (v[0] if v.ndim == 2 and v.shape[0] == 1 else v) for k, v in load(TEMPORAL_PATH).items()}
    st = g["tmp_rng"].astype(np.int64)
    rng = rng_factory((int(st[0]) & 0xFFFFFFFF) | ((int(st[1]) & 0xFFFFFFFF) << 32))
    u = rng.fill((31,), 0.0, 1.0)
    lens = np.concatenate([[800], 400 + (u * np.float32(401.0)).astype(np.int32) % 401]).astype(np.int32)
    rc = {k[7:]: v for k, v in g.items() if k.startswith("rcfull_")}
    assert np.array_equal(rc["lens"], lens)
    tens = dict(lens=lens, rc_w=rng.fill((512, 21), -0.8, 0.8), rc_in=rng.fill((800 * 32, 512), -1.5, 1.5), rc_od=rng.fill((800 * 32, 512), -1.0, 1.0),
                fsmn_coef=rng.fill((61, 512), -0.5, 0.5), fsmn_in=rng.fill((800, 512), -1.5, 1.5), fsmn_od=rng.fill((800, 512), -1.0, 1.0))
    # behind them: GruStreams 512 -> 512, S = 32, T = 60 (`grufull`): W_x [3H x D], W_h [2H x H], W_g [H x H] in [-0.03, 0.03), bias, in, od
    tens.update(gru_Wx=rng.fill((1536, 512), -0.03, 0.03), gru_Wh=rng.fill((1024, 512), -0.03, 0.03), gru_Wg=rng.fill((512, 512), -0.03, 0.03),
                gru_bias=rng.fill((1536,), -0.3, 0.3), gru_in=rng.fill((60 * 32, 512), -1.5, 1.5), gru_od=rng.fill((60 * 32, 512), -1.0, 1.0))
    tens["gru"] = {k[8:]: v for k, v in g.items() if k.startswith("grufull_")}
    return rc, {k[9:]: v for k, v in g.items() if k.startswith("fsmnfull_")}, tens


CFG2_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dnn_cfg2_fullsize.bin")


def load_cfg2_fullsize():
    """tests/golden/dnn_cfg2_fullsize.bin (`oracle/_ref/ref_dnn_bench golden <file>`): two training steps of BASELINE cfg2 itself -- 440 -> 5 x 2048 +
    BatchNormalization + Sigmoid -> 3000, minibatch 1024, learn rate 0.008 -- issued on the reference's library in the order of Nnet::Propagate /
    Backpropagate (oracle/ref_dnn_bench.cpp), as a digest (stride 257).  Returns (records without the prefix, generator state, stride)."""
    g = {k: (v[0] if v.ndim == 2 and v.shape[0] == 1 else v) for k, v in load(CFG2_PATH).items()}
    rng = g["cfg2_rng"].astype(np.int64)
    state = (int(rng[0]) & 0xFFFFFFFF) | ((int(rng[1]) & 0xFFFFFFFF) << 32)
    return {k[5:]: v for k, v in g.items() if k.startswith("cfg2_")}, state, int(g["cfg2_stride"][0])


def replay_cfg2(rng):
    """the tensors of that fixture in the generator's order: six weight matrices uniform in [-0.07, 0.07), then per step a minibatch uniform in
    [-1.7, 1.7) and one label per frame ((int)(u * 3000) % 3000 in float32 arithmetic, as ref_dnn_bench.cpp draws it)"""
    IN, HID, NH, OUT, MB = 440, 2048, 5, 3000, 1024
    W = [rng.fill((OUT if l == NH else HID, IN if l == 0 else HID), -0.07, 0.07) for l in range(NH + 1)]
    batches = []
    for _ in range(2):
        x = rng.fill((MB, IN), -1.7, 1.7)
        u = rng.fill((MB,), 0.0, 1.0)
        batches.append((x, ((u * np.float32(OUT)).astype(np.int32) % OUT).astype(np.int32)))
    return W, batches


COMPONENT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "component_ops.bin")


def load_components():
    """tests/golden/component_ops.bin: the op sequences of LinearTransform, ConvolutionalComponent, MaxPoolingComponent, LengthNormComponent,
    Pnorm / Maxout issued on the reference's CuMatrix library (generator oracle/gen_component_golden.cpp)."""
    return {k: (v[0] if v.ndim == 2 and v.shape[0] == 1 else v) for k, v in load(COMPONENT_PATH).items()}


def load_known_answers():
    """tests/golden/component_known_answers.json: the vectors of the reference's own nnet-component-test.cc (oracle/gen_component_known_answers.py)"""
    import json
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "component_known_answers.json")))
    out = {}
    for name, rec in doc["tests"].items():
        out[name] = {"component": rec.get("component"), "format": rec.get("component_format"),
                     "matrices": {k: np.asarray(v, np.float32) for k, v in rec["matrices"].items()}}
    return out


def load(path=PATH):
    b = open(path, "rb").read()
    out, p = {}, 0
    while p < len(b):
        name = b[p:p + 32].split(b"\0")[0].decode()
        r, c, kind = struct.unpack("<iii", b[p + 32:p + 44])
        p += 44
        esz = 8 if kind == 2 else 4
        a = np.frombuffer(b[p:p + esz * r * c], (np.float32, np.int32, np.float64)[kind]).reshape(r, c).copy()
        p += esz * r * c
        out[name] = a[0] if kind else a
    return out
