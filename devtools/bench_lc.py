"""LC-BLSTM (BASELINE.json cfg3) step timing: 4 x BLstmProjectedStreamsLC (C=512, R=256), in 40,
chunk 40 + right context 20 (T=60), S streams, Affine 512->A + Softmax + Xent.  Reports valid
frames/s (chunk*S per step, like the reference's fps) and the per-kernel breakdown via hip events.
Usage: python devtools/bench_lc.py [S] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import aslp_import  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CHUNK, RIGHT, A = 40, 20, 3000
T = CHUNK + RIGHT

aslp = aslp_import.load()
aslp.ops.use_torch_stream()
lines = ["<NnetProto>"]
d = 40
for _ in range(4):
    lines.append("<BLstmProjectedStreamsLC> <InputDim> %d <OutputDim> 512 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0" % d)
    d = 512
lines.append("<AffineTransform> <InputDim> 512 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04" % A)
lines.append("<Softmax> <InputDim> %d <OutputDim> %d" % (A, A))
lines.append("</NnetProto>")
net = aslp.Nnet.Init("\n".join(lines) + "\n", seed=777)
net.SetTrainOptions(learn_rate=1e-5, momentum=0.9)
net.SetChunkSize(CHUNK)
if os.environ.get('NO_OVERLAP') == '1': net.SetUpdateOverlap(False)
if os.environ.get('NO_FUSION') == '1': net.SetLayerFusion(False)
xent = aslp.Xent()
dev = torch.device("cuda:0")
x = torch.randn(T * S, 40, device=dev)
labels = torch.randint(0, A, (T * S,), device=dev, dtype=torch.int32)
fw = torch.ones(T * S, device=dev)
fw.view(T, S)[CHUNK:] = 0  # right-context frames carry no loss (frame mask of the LC tool)


def step(i):
    net.ResetLstmStreams([1] * S if i == 0 else [0] * S)
    net.TrainStepXent(xent, x, labels, fw)


for i in range(3):
    step(i)
torch.cuda.synchronize()
aslp.lib.aslp_lstm_seq_polls(1)
t0 = time.perf_counter()
for i in range(STEPS):
    step(i + 3)
torch.cuda.synchronize()
el = time.perf_counter() - t0
if os.environ.get("SEQ_TIMING") == "1":
    import ctypes as C
    for mode, name in ((1, "forward"), (2, "backward")):
        aslp.lib.aslp_lstm_seq_timing(mode, None)
        for i in range(5):
            step(i + 100)
        buf = (C.c_ulonglong * 8)()
        aslp.lib.aslp_lstm_seq_timing(0, buf)
        n = max(1, buf[0])
        names = ("collect m(t-1)", "barrier", "product", "barrier", "epilogue") if mode == 1 else ("product+publish", "sum shares", "collect shares", "barrier", "gate diffs")
        vals = [buf[k] * 0.01 / n for k in (1, 3, 2, 4, 5)]
        print("%s kernel, workgroup 0 wave 0, us per timestep: %s   (L2-local launches %d of %d; workgroup 0 resident %.1f us per launch)"
              % (name, "  ".join("%s %.2f" % (nm, v) for nm, v in zip(names, vals)), buf[6], n // T, buf[7] * 0.01 / max(1, n // T)))
if os.environ.get("SEQ_RESIDENCY") == "1":   # when do the workgroups of a persistent launch arrive and leave?
    import ctypes as C
    aslp.lib.aslp_lstm_seq_residency.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for mode, name in ((3, "forward"), (4, "backward")):
        aslp.lib.aslp_lstm_seq_timing(mode, None)
        step(300)
        aslp.lib.aslp_lstm_seq_timing(0, None)
        # the step's 8 persistent launches: forward layers 0..3, then backward layers 3..0 -> the traced kind sits at these distances
        for back, layer in (((7, 0), (6, 1), (5, 2), (4, 3)) if mode == 3 else ((3, 3), (2, 2), (1, 1), (0, 0))):
            buf = (C.c_ulonglong * 512)()
            aslp.lib.aslp_lstm_seq_residency(buf, 256, back)
            ent = [buf[2 * i] for i in range(256)]
            ext = [buf[2 * i + 1] for i in range(256)]
            e0 = min(ent)
            print("%s layer %d: last entry %.2f us; exit first %.1f / median %.1f / last %.1f us; last exit per chain: %s"
                  % (name, layer, (max(ent) - e0) * 0.01, (min(ext) - e0) * 0.01, (sorted(ext)[128] - e0) * 0.01, (max(ext) - e0) * 0.01,
                     " ".join("%.1f" % ((max(ext[c::8]) - e0) * 0.01) for c in range(8))))
print("hand-off re-polls per step (all waves): %.0f" % (aslp.lib.aslp_lstm_seq_polls(1) / STEPS))
if os.environ.get("GEMM_PROFILE") == "1":
    import ctypes as C
    aslp.lib.aslp_gemm_profile(1)
    aslp.lib.aslp_gemm_profile_reset()
    for i in range(5):
        step(i + 200)
    torch.cuda.synchronize()
    aslp.lib.aslp_gemm_profile(0)
    aslp.lib.aslp_gemm_profile_dump()
    for vi, name in enumerate(("NT", "NN", "TN", "TT")):
        fl, ms = C.c_double(), C.c_double()
        n = aslp.lib.aslp_gemm_profile_get(vi, C.byref(fl), C.byref(ms))
        if n:
            print("GEMM %s: %d launches/step, %.1f GF/step, %.3f ms/step, %.1f TFLOP/s" % (name, n / 5, fl.value / 5e9, ms.value / 5, fl.value / ms.value / 1e9))
print("S=%d  ms/step %.3f  valid frames/s %.0f  rows/s %.0f  xent/frame %.4f" % (
    S, el * 1e3 / STEPS, CHUNK * S * STEPS / el, T * S * STEPS / el,
    (xent.GetStats()["loss"] - xent.GetStats()["entropy"]) / max(1.0, xent.GetStats()["frames"])))
