"""A stack of unidirectional projected LSTM layers (4 x LstmProjectedStreams C 512, R 256, S streams, T frames + AffineTransform + Softmax + Xent):
train-step time through the engine.  Usage: python devtools/bench_lstm_stack.py [S] [T] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aslp_import
aslp = aslp_import.load(); aslp.ops.use_torch_stream()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
A, d = 3000, 40
lines = ["<NnetProto>"]
for _ in range(4):
    lines.append("<LstmProjectedStreams> <InputDim> %d <OutputDim> 256 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0" % d)
    d = 256
lines += ["<AffineTransform> <InputDim> 256 <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04" % A, "<Softmax> <InputDim> %d <OutputDim> %d" % (A, A), "</NnetProto>"]
net = aslp.Nnet.Init("\n".join(lines) + "\n", seed=1)
net.SetTrainOptions(learn_rate=1e-5, momentum=0.9)
xent = aslp.Xent()
x = torch.randn(T * S, 40, device=dev)
lab = torch.randint(0, A, (T * S,), device=dev, dtype=torch.int32)
def step(i):
    net.ResetLstmStreams([1] * S if i == 0 else [0] * S)
    net.TrainStepXent(xent, x, lab)
for i in range(20): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N): step(i + 20)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / N
print("4 x LstmProjectedStreams S=%d T=%d: %.3f ms/step, %.0f k rows/s" % (S, T, el * 1e3, T * S / el / 1e3))
