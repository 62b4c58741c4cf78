"""One-layer recurrent components (SURVEY 8d cfg5 swaps) at S streams x T frames: train-step time through the engine.
Usage: python devtools/bench_rnn.py [S] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import aslp_import
aslp = aslp_import.load(); aslp.ops.use_torch_stream()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
A = 128
cases = [("GruStreams", "<GruStreams> <InputDim> 512 <OutputDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0", 512),
         ("LstmProjectedStreams", "<LstmProjectedStreams> <InputDim> 512 <OutputDim> 256 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0", 256),
         ("LstmCifgProjectedStreams", "<LstmCifgProjectedStreams> <InputDim> 512 <OutputDim> 256 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0", 256),
         ("BLstmProjectedStreams", "<BLstmProjectedStreams> <InputDim> 512 <OutputDim> 512 <CellDim> 512 <ParamScale> 0.02 <ClipGradient> 5.0", 512),
         ("Lstm", "<Lstm> <InputDim> 512 <OutputDim> 512 <ParamScale> 0.01 <ClipGradient> 5.0", 512),
         ("BLstm", "<BLstm> <InputDim> 512 <OutputDim> 1024 <ParamScale> 0.01 <ClipGradient> 5.0", 1024)]
for name, line, od in cases:
    proto = "<NnetProto>\n%s\n<AffineTransform> <InputDim> %d <OutputDim> %d <BiasMean> 0.0 <BiasRange> 0.0 <ParamStddev> 0.04\n<Softmax> <InputDim> %d <OutputDim> %d\n</NnetProto>\n" % (line, od, A, A, A)
    net = aslp.Nnet.Init(proto, seed=1)
    net.SetTrainOptions(learn_rate=1e-5, momentum=0.9)
    xent = aslp.Xent()
    x = torch.randn(T * S, 512, device=dev)
    lab = torch.randint(0, A, (T * S,), device=dev, dtype=torch.int32)
    net.SetSeqLengths([T] * S)
    def step(i):
        net.ResetLstmStreams([1] * S if i == 0 else [0] * S)
        net.TrainStepXent(xent, x, lab)
    for i in range(3): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for i in range(n): step(i + 3)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
    print("%-26s S=%d T=%d: %.3f ms/step, %.1f us per timestep (fwd+bwd), %.0f k rows/s" % (name, S, T, el * 1e3, el * 1e6 / T, T * S / el / 1e3))
