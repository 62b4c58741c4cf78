"""GPU parity of RowConvolution and CompactFsmn (csrc/temporal.hip + nnet/nnet-temporal.h) against
oracle/aslp_oracle_temporal.c through the C ABI: output, input diff, updated taps over consecutive
batches, ragged lengths, sizes that are not multiples of the tile shapes."""
import numpy as np
import pytest
import torch

import nnet_io

pytestmark = pytest.mark.gpu
TOL = 1e-4


# K + 1 taps select the instantiation of the streaming kernels (temporal.hip rowconv_fwd_stream / rowconv_bwd_fused <ring length, exact>):
# K = 2 -> <4>, 3 and 5 -> <8>, 12 and 15 -> <16>, 20 (cfg5's FutureContext, at its full size D=512 / T=800 / S=32) -> <21, exact>, 31 -> <32>;
# an odd width (33) and more than 32 taps (40, 63, 64, 100, 130: the reference has no limit on FutureContext) run on the round-1 kernels,
# whose tap gradients go in groups of 64 taps
@pytest.mark.parametrize("dims", [(4, 2, 6, 3), (70, 3, 21, 4), (256, 2, 50, 8), (33, 5, 9, 2), (64, 12, 40, 4), (96, 15, 33, 5),
                                  (512, 20, 800, 32), (40, 31, 64, 3), (48, 40, 90, 3), (130, 63, 70, 2), (72, 64, 150, 3), (40, 100, 120, 2),
                                  (36, 130, 90, 2)])
def test_rowconv_train_steps_match_oracle(aslp, oracle, dev, tmp_path, dims):
    D, K, T, S = dims
    rng = np.random.default_rng(5)
    m = oracle.RowConv(D, K, rng)
    path = tmp_path / "rc.nnet"
    nnet_io.write_simple_nnet(path, [("<RowConvolution>", D, D, nnet_io.rowconv(m.w))])
    net = aslp.Nnet.Read(path)
    lr, mmt = 0.01, 0.9
    net.SetTrainOptions(learn_rate=lr, momentum=mmt)
    assert oracle.rel_err(net.GetParams(), m.w.ravel()) == 0.0
    for step in range(2 if T * S * D > 1 << 22 else 3):
        lens = rng.integers(1, T + 1, S).astype(np.int32)
        lens[0] = T
        x = rng.standard_normal((T * S, D)).astype(np.float32)
        od = rng.standard_normal((T * S, D)).astype(np.float32)
        net.SetSeqLengths(lens)
        out_ref = m.propagate(x, T, S, lens)
        idf_ref = m.backpropagate(od, T, S, lens)
        m.update(lr, mmt)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, out_ref) < TOL, ("out", step)
        idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf, idf_ref) < TOL, ("in_diff", step)
        assert oracle.rel_err(net.GetParams(), m.w.ravel()) < TOL, ("params", step)
        assert np.all(out.reshape(T, S, D)[lens[1]:, 1] == 0)


# (40, 120, 120, 50): 241 taps do not fit the LDS tiles of fsmn_filter_lds / fsmn_backward_fused -- the round-1 kernels serve them;
# (512, 30, 30, 1000): 32 chunks of frames, (64, 30, 30, 200): 7, (16, 40, 40, 5): one partial tile
@pytest.mark.parametrize("dims", [(5, 3, 2, 11), (64, 30, 30, 200), (130, 20, 10, 77), (512, 30, 30, 1000), (16, 40, 40, 5), (40, 120, 120, 50)])
def test_fsmn_train_steps_match_oracle(aslp, oracle, dev, tmp_path, dims):
    D, P, F, T = dims
    rng = np.random.default_rng(6)
    m = oracle.Fsmn(D, P, F, rng, scale=0.1)
    path = tmp_path / "fsmn.nnet"
    nnet_io.write_simple_nnet(path, [("<CompactFsmn>", D, D, nnet_io.fsmn(m.coef, P, F, lr_coef=0.5))])
    net = aslp.Nnet.Read(path)
    lr = 0.01
    net.SetTrainOptions(learn_rate=lr, momentum=0.9)  # momentum is ignored by this component
    for step in range(2):
        Tn = T if step == 0 else max(1, T - 3)
        x = rng.standard_normal((Tn, D)).astype(np.float32)
        od = rng.standard_normal((Tn, D)).astype(np.float32)
        out_ref = m.propagate(x)
        idf_ref = m.backpropagate(x, od, 0.0)
        m.update(lr * 0.5)
        out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert oracle.rel_err(out, out_ref) < TOL, ("out", step)
        idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
        assert oracle.rel_err(idf, idf_ref) < TOL, ("in_diff", step)
        assert oracle.rel_err(net.GetParams(), m.coef.ravel()) < TOL, ("params", step)


def test_fsmn_rejects_more_than_max_frames(aslp, oracle, dev, tmp_path):
    rng = np.random.default_rng(1)
    m = oracle.Fsmn(8, 2, 2, rng)
    path = tmp_path / "fsmn.nnet"
    nnet_io.write_simple_nnet(path, [("<CompactFsmn>", 8, 8, nnet_io.fsmn(m.coef, 2, 2))])
    net = aslp.Nnet.Read(path)
    with pytest.raises(RuntimeError):  # KALDI_ASSERT(T <= max_frames_), cfsmn.h:175
        net.Propagate(torch.zeros(3001, 8, device=dev))


def test_temporal_init_and_io(aslp, dev, tmp_path):
    proto = """<NnetProto>
<CompactFsmn> <InputDim> 12 <OutputDim> 12 <PastContext> 4 <FutureContext> 3 <LearnRateCoef> 0.5 <ClipGradient> 1.0
<RowConvolution> <InputDim> 12 <OutputDim> 12 <FutureContext> 2
</NnetProto>
"""
    net = aslp.Nnet.Init(proto, seed=777)
    assert net.NumParams() == 8 * 12 + 12 * 3
    p = tmp_path / "t.nnet"
    net.Write(p, binary=True)
    net2 = aslp.Nnet.Read(p)
    assert np.array_equal(net.GetParams(), net2.GetParams())
    p2 = tmp_path / "t.txt"
    net.Write(p2, binary=False)
    assert np.allclose(aslp.Nnet.Read(p2).GetParams(), net.GetParams(), rtol=1e-6)
    with pytest.raises(RuntimeError):
        aslp.Nnet.Init("<NnetProto>\n<RowConvolution> <InputDim> 12 <OutputDim> 8 <FutureContext> 2\n</NnetProto>\n")
