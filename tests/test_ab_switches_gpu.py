"""The A/B switches of the recurrent kernels (DESIGN "A/B switches") select between code paths that must agree: a child process
per setting (the switches are read once per process) trains the same small latency-controlled BLSTM for two chunks and reports a
digest of output, input diff and parameters.  The wave-local collection and the pinned read-ahead change WHEN operands are fetched,
not what is multiplied: bit-identical to the workgroup-wide / scheduler-ordered path.  The per-timestep kernels sum the recurrent
product in another order: equal within the fp32 bar."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, json, hashlib, os
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import aslp_import
aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device("cuda:0")
S, chunk, T, D, C = int(os.environ.get("AB_S", "16")), int(os.environ.get("AB_CHUNK", "6")), int(os.environ.get("AB_T", "9")), int(os.environ.get("AB_D", "24")), 128
layer = os.environ.get("AB_MARKER", "<BLstmProjectedStreamsLC>") + " <InputDim> %%d <OutputDim> 128 <CellDim> %%d <ParamScale> %%s <ClipGradient> %%s\n"
proto = "<NnetProto>\n" + "".join(layer %% (D if l == 0 else 128, C, os.environ.get("AB_PSCALE", "0.05"), os.environ.get("AB_CLIP", "5.0"))
                                  for l in range(int(os.environ.get("AB_LAYERS", "1")))) + "</NnetProto>\n"
net = aslp.Nnet.Init(proto, seed=5)
net.SetTrainOptions(learn_rate=float(os.environ.get("AB_LR", "1e-3")), momentum=0.9)
net.SetChunkSize(chunk)
g = torch.Generator(device="cpu"); g.manual_seed(3)
outs = []
for step in range(2):
    x = torch.randn(T * S, D, generator=g).to(dev)
    od = (torch.randn(T * S, 128, generator=g) * float(os.environ.get("AB_ODSCALE", "0.1"))).to(dev)
    net.ResetLstmStreams([1] * S if step == 0 else [0] * S)
    out = net.Propagate(x).cpu().numpy()
    idf = net.Backpropagate(od, want_in_diff=True).cpu().numpy()
    outs += [out, idf]
outs.append(np.asarray(net.GetParams(), np.float32))
np.save(sys.argv[1], np.concatenate([o.ravel() for o in outs]))
'''


def run(tmp_path, name, **env):
    out = str(tmp_path / (name + ".npy"))
    e = dict(os.environ, **env)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, out], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return np.load(out)


def test_recurrent_ab_switches_agree(tmp_path):
    base = run(tmp_path, "default")                       # chains of 8 streams, gate non-linearities on v_exp_f32 / v_rcp_f32, products on v_mfma_f32_16x16x32_f16
    assert np.isfinite(base).all()
    for name, env in (("wg_collect", {"ASLP_LSTM_WAVE_COLLECT": "0"}), ("own_order", {"ASLP_LSTM_READ_AHEAD": "0"}),
                      ("both_off", {"ASLP_LSTM_WAVE_COLLECT": "0", "ASLP_LSTM_READ_AHEAD": "0"})):
        other = run(tmp_path, name, **env)
        assert np.array_equal(base, other), name            # same operands, same order of products: same bits

    def close(a, b, tol=1e-5):
        return np.linalg.norm(a - b) / np.linalg.norm(b) < tol and np.abs(a - b).max() / max(1.0, np.abs(b).max()) < 10 * tol

    # the recurrent products on the fp32 matrix instruction (lstm_seq_fwd / lstm_seq_bwd) instead of two-piece fp16 operands: same values to
    # fp32 rounding, and that family's own switches still agree bit for bit
    f32 = run(tmp_path, "fp32_mfma", ASLP_LSTM_SPLIT_F16="0")
    assert close(f32, base)
    for name, env in (("f32_wg_collect", {"ASLP_LSTM_WAVE_COLLECT": "0"}), ("f32_own_order", {"ASLP_LSTM_READ_AHEAD": "0"})):
        assert np.array_equal(f32, run(tmp_path, name, ASLP_LSTM_SPLIT_F16="0", **env)), name
    exact = run(tmp_path, "exact_act", ASLP_LSTM_FAST_ACT="0")      # correctly rounded expf / division (the reference's CPU bits)
    assert close(exact, base)
    step = run(tmp_path, "per_timestep", ASLP_LSTM_PERSISTENT="0")
    assert close(step, base)


def test_one_launch_conversion_changes_no_bit(tmp_path):
    """ASLP_COPY_PLANES=0 sends every fp32 -> fp16-planes conversion through a maximum launch and a conversion launch again; the default
    makes maxima and planes of up to eight matrices in ONE launch whose workgroups find each matrix' maximum among themselves
    (nn_fused.hip copy_planes_coop: the layer's six weight matrices and its input, the gate diffs, ...).  Same maxima, same planes, so the
    same products: a layer wide enough for the planes path (T S = 128 rows, 64-wide input, C 128, R 64) must come out bit for bit."""
    sizes = {"AB_S": "16", "AB_T": "8", "AB_CHUNK": "6", "AB_D": "64"}
    one = run(tmp_path, "one_launch", **sizes)
    assert np.isfinite(one).all()
    assert np.array_equal(one, run(tmp_path, "two_launches", ASLP_COPY_PLANES="0", **sizes))
    # ... and the planes path is what ran: products on the fp32 instruction give other bits
    assert not np.array_equal(one, run(tmp_path, "no_planes", ASLP_LSTM_PLANES="0", **sizes))


def test_d_r_beside_the_lower_recurrence_changes_no_bit(tmp_path):
    """In a stack of LC-BLSTM layers the upper layer's d_r product (which only its W_rm gradient reads) is issued with the gradients, beside the
    recurrence of the layer below, instead of in front of the layer's in-diff on the main stream (ASLP_LSTM_DR_ASIDE=0: as before).  Same
    products on another stream: output, input diff and parameters of two training steps must not change by a bit -- which also says that
    nothing the main stream does meanwhile touches what that product reads or writes."""
    sizes = {"AB_S": "16", "AB_T": "8", "AB_CHUNK": "6", "AB_D": "64", "AB_LAYERS": "3"}
    aside = run(tmp_path, "aside", **sizes)
    assert np.isfinite(aside).all()
    for k in range(3):
        assert np.array_equal(aside, run(tmp_path, "main_%d" % k, ASLP_LSTM_DR_ASIDE="0", **sizes)), k
    assert np.array_equal(aside, run(tmp_path, "aside_again", **sizes))
    # ... and in a stack of unidirectional projected layers (d_r = out_diff + dGATES(next) W_r by the single-direction product)
    uni = dict(sizes, AB_MARKER="<LstmProjectedStreams>")
    a = run(tmp_path, "uni_aside", **uni)
    assert np.isfinite(a).all() and np.array_equal(a, run(tmp_path, "uni_main", ASLP_LSTM_DR_ASIDE="0", **uni))


@pytest.mark.parametrize("pscale,odscale,lr", [("0.3", "1e-9", "1e-3"), ("0.001", "1e5", "1e-12"), ("0.05", "1e-20", "1e-3"), ("0.2", "30.0", "1e-6")])
def test_split_f16_products_keep_fp32_accuracy_at_any_magnitude(tmp_path, pscale, odscale, lr):
    """The recurrent products carry each fp32 operand as two fp16 pieces behind power-of-two scales (weights per column, gate diffs per
    stream and timestep: csrc/rnn_persistent.hip lstm_seq_fwd_h / lstm_seq_bwd_h), so nothing depends on the operands' magnitude: large
    weights with vanishing gradients, tiny weights with huge gradients (no clipping), gradients near the bottom of fp32's range -- output,
    input diff and the parameters after two updates agree with the fp32-instruction kernels like two fp32 summation orders do.  (Weights
    large enough to saturate the gates are left out: that recurrence is chaotic and any two fp32 summation orders part ways on it; the
    column scale itself is exercised over 10 orders of magnitude by devtools/micro/f16_split.hip.)"""
    env = dict(AB_PSCALE=pscale, AB_ODSCALE=odscale, AB_LR=lr, AB_CLIP="0.0")
    split = run(tmp_path, "split", **env)
    f32 = run(tmp_path, "f32", ASLP_LSTM_SPLIT_F16="0", **env)
    assert np.isfinite(split).all() and np.isfinite(f32).all()
    n = 2 * (9 * 16 * 128 + 9 * 16 * 24)        # outputs and input diffs of the two chunks, then the parameters
    for name, a, b in (("out/in_diff", split[:n], f32[:n]), ("params", split[n:], f32[n:])):
        # per block of equal kind: input diffs of the 1e-9 case are ~1e-9 themselves, so compare blockwise relative to each block's own size
        for lo, hi in ((0, 9 * 16 * 128), (9 * 16 * 128, 9 * 16 * (128 + 24))) if name != "params" else ((0, len(a)),):
            x, y = a[lo:hi].astype(np.float64), b[lo:hi].astype(np.float64)
            assert np.linalg.norm(x - y) <= 1e-5 * np.linalg.norm(y) + 1e-30, (name, lo, np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300))


def test_lstm_layer_products_from_prepared_planes(tmp_path):
    """ASLP_LSTM_PLANES=3 (the default): the bidirectional layer keeps fp16 planes of its products' operands (layer input, weights, m, dGATES, d_r,
    r: made by multi-matrix conversion launches, shifted row ranges as windows) and every batched pair product reads them; 1 / 2: only the
    forward / pre-recurrence ones; 4: all but the weight gradients issued beside a recurrence; 0: the pairs on the fp32 instruction.  T S, D, C
    and R are multiples of 64 here so that every product is served.  Same values to fp32 rounding."""
    shape = {"AB_S": "32", "AB_CHUNK": "5", "AB_T": "8", "AB_D": "64"}
    base = run(tmp_path, "planes_off", ASLP_LSTM_PLANES="0", **shape)
    assert np.isfinite(base).all()
    for level in ("1", "2", "3", "4"):
        other = run(tmp_path, "planes_" + level, ASLP_LSTM_PLANES=level, **shape)
        assert np.linalg.norm(other - base) / np.linalg.norm(base) < 1e-5, level
        assert np.abs(other - base).max() / max(1.0, np.abs(base).max()) < 1e-4, level
    # the maxima of the gate diffs from the backward recurrence itself and the known bound of m (|o tanh c| < 1) instead of maximum passes:
    # at most another power-of-two scale of the same planes
    passes = run(tmp_path, "max_passes", ASLP_LSTM_KNOWN_BOUNDS="0", **shape)
    default = run(tmp_path, "default", **shape)
    assert np.linalg.norm(passes - default) / np.linalg.norm(default) < 1e-6
    assert np.abs(passes - default).max() / max(1.0, np.abs(default).max()) < 1e-5
    # the buffers' preparation for the persistent kernels as a launch of its own instead of riding along with the conversion launch behind it
    # (split16.h SeqFillJob): the same stores, the same bits
    assert np.array_equal(run(tmp_path, "fill_alone", ASLP_LSTM_FILL_ALONG="0", **shape), default)
