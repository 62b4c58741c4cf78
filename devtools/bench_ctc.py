"""Times the HIP CTC loss: 32 utterances x <= 800 frames x alphabet 128 (SURVEY 8d CTC variant)."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
A, mb, maxT = 128, 32, 800
rng = np.random.default_rng(0)
in_len = rng.integers(200, maxT + 1, mb).astype(np.int32); in_len[0] = maxT
labels = [[int(v) for v in rng.integers(1, A, int(t) // 4)] for t in in_len]
acts = torch.from_numpy(rng.random((maxT * mb, A)).astype(np.float32)).to(dev)
for _ in range(3): aslp.ops.ctc_loss(acts, labels, in_len)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): aslp.ops.ctc_loss(acts, labels, in_len)
torch.cuda.synchronize()
print("ctc_loss 32 x <=800 x 128: %.3f ms per call" % ((time.perf_counter() - t0) * 1e3 / n))
