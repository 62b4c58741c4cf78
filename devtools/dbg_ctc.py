import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import aslp_import, oracle_lib as oracle
from test_oracle_ctc_cpu import orc_ctc
aslp = aslp_import.load(); dev = torch.device("cuda:0")
for (A, mb, maxT, seed) in [(128, 8, 200, 5), (128, 8, 400, 5), (128, 8, 800, 5), (30, 8, 800, 6)]:
    rng = np.random.default_rng(seed)
    in_len = rng.integers(1, maxT + 1, mb).astype(np.int32); in_len[0] = maxT
    labels = []
    for t in in_len:
        L = int(rng.integers(0, max(1, t // 2) + 1)); lab = rng.integers(1, A, L)
        labels.append([int(v) for v in lab])
    acts = (rng.standard_normal((maxT * mb, A)) * 2).astype(np.float32)
    flat = np.array([v for l in labels for v in l], np.int32); lab_len = np.array([len(l) for l in labels], np.int32)
    rcost, rgrad = orc_ctc(oracle, acts.reshape(-1).copy(), flat, lab_len, in_len, A, mb)
    costs, grads = aslp.ops.ctc_loss(torch.from_numpy(acts).to(dev), labels, in_len)
    g = grads.cpu().numpy().reshape(-1)
    fin = np.isfinite(rcost) & (rcost > 0)
    print(A, mb, maxT, "cost rel %.2e" % np.max(np.abs(costs[fin] - rcost[fin]) / rcost[fin]), "grad rel-frob %.2e" % oracle.rel_err(g, rgrad),
          "max abs %.2e" % np.abs(g - rgrad).max())
