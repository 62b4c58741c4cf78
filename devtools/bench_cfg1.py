"""BASELINE.json configs[0] on the GPU: the 5x2048 sigmoid DNN WITHOUT BatchNorm, 440 -> 3000, minibatch 256, lr 0.008, no
momentum (run_dnn.sh) -- the configuration the reference's CPU path is quoted on (2.26 k frames/s).
Usage: python devtools/bench_cfg1.py [minibatch] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import aslp_import  # noqa: E402

MB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 500
aslp = aslp_import.load()
aslp.ops.use_torch_stream()
lines, d = ["<NnetProto>"], 440
for _ in range(5):
    lines.append("<AffineTransform> <InputDim> %d <OutputDim> 2048 <BiasMean> -2.0 <BiasRange> 4.0 <ParamStddev> 0.04" % d)
    lines.append("<Sigmoid> <InputDim> 2048 <OutputDim> 2048")
    d = 2048
lines += ["<AffineTransform> <InputDim> 2048 <OutputDim> 3000 <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04",
          "<Softmax> <InputDim> 3000 <OutputDim> 3000", "</NnetProto>"]
net = aslp.Nnet.Init("\n".join(lines) + "\n", seed=777)
net.SetTrainOptions(learn_rate=0.008, momentum=0.0)
xent = aslp.Xent()
dev = torch.device("cuda:0")
x = torch.randn(MB, 440, device=dev)
lab = torch.randint(0, 3000, (MB,), device=dev, dtype=torch.int32)
for _ in range(50):
    net.TrainStepXent(xent, x, lab)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(STEPS):
    net.TrainStepXent(xent, x, lab)
torch.cuda.synchronize()
dt = (time.time() - t0) / STEPS
print("cfg1 minibatch %d: %.3f ms/step, %.0f frames/s, %.1f TFLOP/s algorithmic" % (MB, dt * 1e3, MB / dt, MB / dt * 141131776 / 1e12))
if os.environ.get("GEMM_PROFILE") == "1":
    import ctypes as C
    net.SetUpdateOverlap(False)
    aslp.lib.aslp_gemm_profile(1)
    aslp.lib.aslp_gemm_profile_reset()
    for _ in range(20):
        net.TrainStepXent(xent, x, lab)
    torch.cuda.synchronize()
    aslp.lib.aslp_gemm_profile(0)
    aslp.lib.aslp_gemm_profile_dump()
    for vi, name in enumerate(("NT", "NN", "TN", "TT")):
        fl, ms = C.c_double(), C.c_double()
        n = aslp.lib.aslp_gemm_profile_get(vi, C.byref(fl), C.byref(ms))
        if n:
            print("GEMM %s: %d launches/step, %.1f GF/step, %.3f ms/step, %.1f TFLOP/s" % (name, n / 20, fl.value / 20e9, ms.value / 20, fl.value / ms.value / 1e9))
