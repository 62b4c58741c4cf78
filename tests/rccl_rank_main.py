"""One rank of the multi-GPU RCCL tests (tests/test_rccl_multigpu_gpu.py starts N of these, one per GPU, with RANK /
WORLD_SIZE / ASLP_COMM_FILE / ASLP_COMM_TOKEN in the environment).  Drives the product's own RcclComm and workers
(libaslp_parallel.so) and prints one JSON object with what this rank observed; the test compares it with closed forms.

Scenarios:
  reduce-barrier   src/aslp-parallel/reduce-barrier-test.cc:15-31: rank r issues r + 4 all-reduces of 1, then all-reduces 0
                   until the global sum is 0 (ranks finish at different times and drain), then a barrier
  bsp | bmuf       bsp-worker.cc:33-65 / bmuf-worker.cc:37-68 on raw device buffers, ranks running out of data one
                   after the other (rank r has steps - r rounds), like aslp-parallel-selftest does with threads
  easgd            easgd-worker.cc:37-80 + easgd-server.cc:37-86: rank 0 serves, workers take turns in a fixed order
  p2p              ncclSend / ncclRecv ring + pairwise exchange
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    kind, dim, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import numpy as np
    import torch
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
    import aslp_import
    aslp = aslp_import.load()
    aslp.ops.use_torch_stream()
    from kaldi_aslp_amd import native_parallel as npar
    # RcclComm, or ShmComm when ASLP_COMM_TRANSPORT=shm (ranks as separate processes on ONE device)
    comm = npar.ProcessComm(os.environ.get("ASLP_COMM_FILE"), rank=rank, num_nodes=world, token=os.environ.get("ASLP_COMM_TOKEN"))   # the library default (900 s: a cold box can take minutes to page RCCL in)
    out = {"rank": comm.Rank(), "world": comm.NumNodes()}
    f32 = np.float32

    def init_model():
        size = dim + dim // 2 + 1
        h = (f32(rank + 1) + f32(0.01) * np.arange(size, dtype=f32)).astype(f32)
        t = torch.from_numpy(h).to(dev)
        return t, [t[:dim], t[dim:dim + dim // 2], t[dim + dim // 2:]]   # three tensors, like a small model

    if kind == "reduce-barrier":
        sums = []
        for _ in range(rank + 4):
            sums.append(comm.AllReduceHostInt([1])[0])
        drain = []
        while True:
            s = comm.AllReduceHostInt([0])[0]
            drain.append(s)
            if s == 0:
                break
        comm.Barrier()
        # the same on device memory: ncclAllReduce(float) and (double) in place
        t = torch.full((1000,), float(rank + 1), device=dev)
        comm.AllReduce(t)
        d = torch.full((77,), float(rank + 1) * 0.5, device=dev, dtype=torch.float64)
        comm.AllReduce(d)
        torch.cuda.synchronize()
        out.update(sums=sums, drain=drain, dev_f32=float(t[0].item()), dev_f32_all_equal=bool((t == t[0]).all().item()),
                   dev_f64=float(d[0].item()))
    elif kind in ("bsp", "bmuf"):
        t, parts = init_model()
        w = npar.BspWorker(comm) if kind == "bsp" else npar.BmufWorker(comm, float(sys.argv[4]), float(sys.argv[5]))
        w.InitParam(parts)
        trace = []
        mine = steps - rank
        for s in range(mine):
            t += f32(0.5 * (rank + 1) + 0.25 * s)
            more = w.Synchronize(100 * (rank + 1) + s)
            torch.cuda.synchronize()
            trace.append(t.cpu().numpy().tolist())
            assert more
        w.Stop()
        torch.cuda.synchronize()
        out.update(trace=trace, final=t.cpu().numpy().tolist())
    elif kind == "easgd":
        alpha = float(sys.argv[4])
        t, parts = init_model()
        if rank == 0:
            npar.ServerRun(comm, "easgd", parts, p0=alpha)
            torch.cuda.synchronize()
            out.update(final=t.cpu().numpy().tolist())
        else:
            w = npar.EasgdWorker(comm, alpha)
            w.InitParam(parts)
            trace = []
            # fixed arrival order at the server: worker 1, 2, ..., 1, 2, ... -- each waits for its turn on a host-side all-reduce
            # among the WORKERS?  RCCL has one communicator here, so the turn is taken through the filesystem instead
            turn_file = os.environ["ASLP_COMM_FILE"] + ".turn"
            nworkers = world - 1
            for s in range(steps):
                my_turn = s * nworkers + (rank - 1)
                import time
                t0 = time.time()
                while True:
                    try:
                        cur = int(open(turn_file).read() or "0")
                    except (OSError, ValueError):
                        cur = 0
                    if cur == my_turn:
                        break
                    if time.time() - t0 > 120:
                        raise RuntimeError("turn %d never came (at %d)" % (my_turn, cur))
                    time.sleep(0.002)
                t += f32(0.5 * rank + 0.25 * s)
                w.Synchronize(10)
                torch.cuda.synchronize()
                trace.append(t.cpu().numpy().tolist())
                tmp = turn_file + ".%d" % rank
                with open(tmp, "w") as f:
                    f.write(str(my_turn + 1))
                os.replace(tmp, turn_file)
            w.Stop()
            out.update(trace=trace)
    elif kind == "p2p":
        a = torch.full((dim,), float(rank + 1), device=dev)
        b = torch.zeros(dim, device=dev)
        nxt, prv = (rank + 1) % world, (rank - 1) % world
        if rank % 2 == 0:
            comm.Send(nxt, a)
            comm.Recv(prv, b)
        else:
            comm.Recv(prv, b)
            comm.Send(nxt, a)
        torch.cuda.synchronize()
        ring = float(b[0].item())
        peer = rank ^ 1
        c = torch.zeros(dim, device=dev)
        if peer < world:
            comm.Exchange(peer, a, c)
            torch.cuda.synchronize()
        out.update(ring=ring, ring_all_equal=bool((b == b[0]).all().item()), exchanged=float(c[0].item()) if peer < world else None)
    else:
        raise SystemExit("unknown scenario " + kind)
    comm.Barrier()
    comm.close()
    print("RCCL_RANK_RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
