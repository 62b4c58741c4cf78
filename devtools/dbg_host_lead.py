"""How far ahead of the GPU does the host get while it issues cfg2 training steps?  (host time to ISSUE n steps vs GPU time to run them)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, aslp_import
aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device("cuda:0")
net = aslp.Nnet.Init(bench.proto(), seed=777); net.SetTrainOptions(learn_rate=0.008, momentum=0.0)
xent = aslp.Xent()
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn(1024, 440, device=dev, generator=g); lab = torch.randint(0, 3000, (1024,), device=dev, generator=g, dtype=torch.int32)
for _ in range(300): net.TrainStepXent(xent, x, lab)
torch.cuda.synchronize()
for n in (5, 10, 20, 40, 80, 160, 320):
    t0 = time.perf_counter()
    for _ in range(n): net.TrainStepXent(xent, x, lab)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n %4d: host issued in %7.2f ms, GPU done after %7.2f ms (%.3f ms/step)  -> host ahead by %.1f ms when it finished issuing" % (n, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / n, (t2 - t1) * 1e3))
