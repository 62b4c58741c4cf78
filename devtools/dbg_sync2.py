import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mode = sys.argv[1]
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from parallel_model import BspWorker
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
def init_pg():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
if mode != "none" and "late" not in mode:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
net = aslp.Nnet.Init(bench.proto(), seed=777)
net.SetTrainOptions(learn_rate=1e-5)
if "nooverlap" in mode: net.SetUpdateOverlap(False)
xent = aslp.Xent()
if "late" in mode: init_pg()
x = torch.randn(1024, 440, device=dev); lab = torch.randint(0, 3000, (1024,), device=dev, dtype=torch.int32)
w = BspWorker(net) if "worker" in mode else None
if "sync" in mode: w.Synchronize(100)
if "allreduce" in mode:
    t = torch.ones(10, device=dev); dist.all_reduce(t)
for _ in range(5): net.TrainStepXent(xent, x, lab)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): net.TrainStepXent(xent, x, lab)
torch.cuda.synchronize()
print(mode, "%.3f ms/step" % ((time.perf_counter() - t0) / 50 * 1e3))
