"""MatrixRandomizer / RandomizerMask (kaldi-aslp_amd/nnet/nnet-randomizer.h) on the GPU: a
line-by-line mirror of the reference's own unit test (src/aslp-nnet/nnet-randomizer-test.cc:
76-124, 1111 rows, capacity 1000, minibatch 100) plus bit-exact row gathers under a real shuffle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_unit_test_matrix_randomizer(aslp, dev):
    rng = np.random.default_rng(0)
    m = rng.standard_normal((1111, 10)).astype(np.float32)
    m2 = torch.from_numpy(m).to(dev)
    r = aslp.MatrixRandomizer(randomizer_size=1000, minibatch_size=100)
    r.AddData(m2)
    assert r.IsFull()
    r.Randomize(np.arange(1111, dtype=np.int32))  # no shuffling
    i = 0
    while not r.Done():
        assert np.array_equal(r.Value().cpu().numpy(), m[i * 100:(i + 1) * 100])
        r.Next()
        i += 1
    assert i == 11
    # filling for the 2nd time: the last 11 rows move to the front
    assert not r.IsFull()
    r.AddData(m2)
    assert r.IsFull()
    assert r.NumFrames() == 11 + 1111
    assert np.array_equal(r.Value().cpu().numpy()[:11], m[1100:1111])
    assert not r.Done()
    while not r.Done():
        r.Value()
        r.Next()
        i += 1
    assert i == 22


def test_unit_test_randomizer_mask(aslp, dev):
    m = aslp.randomizer_mask(5, seed=777)
    assert len(m) == 5 and int(m.sum()) == 4 + 3 + 2 + 1 + 0


@pytest.mark.parametrize("rows,cols", [(1111, 10), (4096, 440), (300, 33)])
def test_shuffled_minibatches_are_exact_row_gathers(aslp, oracle, dev, rows, cols):
    rng = np.random.default_rng(rows)
    m = rng.standard_normal((rows, cols)).astype(np.float32)
    mb = 64
    r = aslp.MatrixRandomizer(randomizer_size=rows - 1, minibatch_size=mb)
    r.AddData(torch.from_numpy(m).to(dev))
    mask = aslp.randomizer_mask(rows, seed=777)
    assert sorted(mask.tolist()) == list(range(rows))
    r.Randomize(mask)
    want = np.empty_like(m)
    oracle.lib.orc_randomize(want, cols, m, cols, cols, mask, rows)  # cu::Randomize, cu-math.cc:80-127
    i = 0
    while not r.Done():
        assert np.array_equal(r.Value().cpu().numpy(), want[i * mb:(i + 1) * mb])  # bit-exact
        r.Next()
        i += 1
    assert i == rows // mb
    # asking for a minibatch that is not there is an error (KALDI_ASSERT, nnet-randomizer.cc:94)
    with pytest.raises(RuntimeError):
        r.Value()
