import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aslp_import, nnet_io, oracle_lib as oracle
from test_rnn_gpu import build, oracle_step, FAMILY
import pathlib, tempfile
aslp = aslp_import.load()
dev = torch.device("cuda:0")
marker = sys.argv[1] if len(sys.argv) > 1 else "<LstmProjectedStreams>"
D, Cc, R, T, S = [int(v) for v in (sys.argv[2:7] if len(sys.argv) > 6 else (5, 8, 4, 6, 3))]
tmp = pathlib.Path(tempfile.mkdtemp())
clip, lr, mmt = 0.5, 0.01, 0.9
dirs, grads, out_dim, path = build(oracle, tmp, marker, D, Cc, R, clip, seed=1)
net = aslp.Nnet.Read(path)
net.SetTrainOptions(learn_rate=lr, momentum=mmt)
rng = np.random.default_rng(7)
bidir, proj, cifg, lc, _ = FAMILY[marker]
carried = (not bidir) or lc
chunk = T - 2 if lc else 0
if lc: net.SetChunkSize(chunk)
state = np.zeros((S, dirs[0].width), np.float32) if carried else None
flat = lambda: np.concatenate([d.flat() for d in dirs])
for step in range(4):
    x = rng.standard_normal((T * S, D)).astype(np.float32)
    od = rng.standard_normal((T * S, out_dim)).astype(np.float32)
    lens = None
    if carried:
        flags = [1] * S if step == 0 else [int(v) for v in rng.integers(0, 2, S)]
        net.ResetLstmStreams(flags)
        for s, fl in enumerate(flags):
            if fl: state[s] = 0
    else:
        lens = rng.integers(1, T + 1, S).astype(np.int32); lens[0] = T
        net.SetSeqLengths(lens)
    out_ref, idf_ref, state = oracle_step(oracle, marker, dirs, grads, x, od, T, S, state, lens, chunk, lr, mmt, clip)
    out = net.Propagate(torch.from_numpy(x).to(dev)).cpu().numpy()
    idf = net.Backpropagate(torch.from_numpy(od).to(dev), want_in_diff=True).cpu().numpy()
    print(step, "out %.2e in_diff %.2e params %.2e" % (oracle.rel_err(out, out_ref), oracle.rel_err(idf, idf_ref), oracle.rel_err(net.GetParams(), flat())),
          "max|out-ref| %.2e" % np.abs(out - out_ref).max(), flush=True)
    o3 = np.abs(out - out_ref).reshape(T, S, -1).max(2)
    print("   per (t,s) max err:\n", np.array2string(o3, precision=1))
