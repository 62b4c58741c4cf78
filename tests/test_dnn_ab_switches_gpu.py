"""A/B switches of the feed-forward step path (DESIGN "A/B switches"): a child process per setting (the switches are read once per
process) trains the same sigmoid DNN for a few steps and hands back loss and parameters.  Switches that change WHEN something runs
(ASLP_LATE_JOIN) must give the same bits; switches that change who writes the operand planes or how a product is tiled give the same
values to fp32 rounding."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, os
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import aslp_import
aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device("cuda:0")
bn, mb, hidden, out = os.environ["AB_BN"] == "1", int(os.environ["AB_MB"]), 1024, 3000
proto, d = "<NnetProto>\n", 440
for _ in range(3):
    proto += "<AffineTransform> <InputDim> %%d <OutputDim> %%d <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.05\n" %% (d, hidden)
    if bn: proto += "<BatchNormalization> <InputDim> %%d <OutputDim> %%d\n" %% (hidden, hidden)
    proto += "<Sigmoid> <InputDim> %%d <OutputDim> %%d\n" %% (hidden, hidden)
    d = hidden
proto += ("<AffineTransform> <InputDim> %%d <OutputDim> %%d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.05\n<Softmax> <InputDim> %%d <OutputDim> %%d\n</NnetProto>\n" %% (d, out, out, out))
net = aslp.Nnet.Init(proto, seed=3)
net.SetTrainOptions(learn_rate=2e-3, momentum=0.9)
xe = aslp.Xent()
g = torch.Generator(device="cpu"); g.manual_seed(5)
for step in range(6):
    x = torch.randn(mb, 440, generator=g).to(dev)
    lab = torch.randint(0, out, (mb,), generator=g, dtype=torch.int32).to(dev)
    net.TrainStepXent(xe, x, lab)
st = xe.GetStats()
res = np.concatenate([[(st["loss"] - st["entropy"]) / st["frames"]], np.asarray(net.GetParams(), np.float64)])
np.save(sys.argv[1], res)
'''


def run(tmp_path, name, **env):
    out = str(tmp_path / (name + ".npy"))
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, out], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return np.load(out)


def close(a, b, tol=2e-6):
    return abs(a[0] - b[0]) < tol * abs(b[0]) and np.linalg.norm(a[1:] - b[1:]) < tol * np.linalg.norm(b[1:])


def test_batchnorm_net_switches(tmp_path):
    base = run(tmp_path, "default", AB_BN="1", AB_MB="1024")
    assert np.isfinite(base).all()
    # the wait for the side-stream weight updates at the end of the backward pass instead of inside the next forward pass: same work, same bits
    assert np.array_equal(base, run(tmp_path, "join_at_end", AB_BN="1", AB_MB="1024", ASLP_LATE_JOIN="0"))
    # round 5's launch-count work changes who makes an operand's planes and when a sum is taken, never a value: in-diff's planes by a
    # conversion launch behind bn_backward_coop instead of by its own workgroups; copy, maximum pass and conversion pass of the network input
    # as three launches; a device declared shared (both of those stand down); every switch together
    for name, env in (("bn_diff_convert", {"ASLP_BN_DIFF_PLANES": "0"}), ("three_launch_input", {"ASLP_COPY_PLANES": "0"}),
                      ("shared_device", {"ASLP_DEVICE_SHARED": "1"}),
                      ("all_off", {"ASLP_BN_DIFF_PLANES": "0", "ASLP_COPY_PLANES": "0", "ASLP_LATE_JOIN": "0"})):
        assert np.array_equal(base, run(tmp_path, name, AB_BN="1", AB_MB="1024", **env)), name
    # Softmax / Xent rows one column per access: the rows' sums are formed in another order
    assert close(run(tmp_path, "scalar_rows", AB_BN="1", AB_MB="1024", ASLP_SOFTMAX_VEC="0"), base, 2e-5)
    # the weights' planes converted in every step / no weight updates beside the backward pass / the fp32 instruction
    assert close(run(tmp_path, "w_convert", AB_BN="1", AB_MB="1024", ASLP_KEEP_WEIGHT_PLANES="0"), base)
    assert close(run(tmp_path, "fp32", AB_BN="1", AB_MB="1024", ASLP_GEMM_SPLIT_F16="0"), base)


def test_minibatch_256_switches(tmp_path):
    base = run(tmp_path, "default", AB_BN="0", AB_MB="256")
    assert np.isfinite(base).all()
    assert np.array_equal(base, run(tmp_path, "join_at_end", AB_BN="0", AB_MB="256", ASLP_LATE_JOIN="0"))
    # 256-row products as K chunks + a second launch instead of whole on 32 x 64 tiles: another summation order
    assert close(run(tmp_path, "k_chunks", AB_BN="0", AB_MB="256", ASLP_GEMM_S16_SMALL="0"), base)
    assert close(run(tmp_path, "fp32", AB_BN="0", AB_MB="256", ASLP_GEMM_SPLIT_F16="0"), base)
