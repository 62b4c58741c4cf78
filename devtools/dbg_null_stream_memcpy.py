"""Does a blocking null-stream hipMemcpy wait for work on a hipStreamNonBlocking stream on this runtime?  (tests/test_b6_raw_writer_gpu.py)"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aslp_import
aslp = aslp_import.load()
f = aslp.lib.hipMemcpy
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev) * 0.01
B = torch.zeros(1 << 20, device=dev)
s = torch.cuda.Stream()
torch.cuda.synchronize()
for trial in range(3):
    B.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        x = a
        for _ in range(40):
            x = x @ a          # ~40 x 1 ms
        B.fill_(1.0)
    host = np.empty(1 << 20, np.float32)
    t0 = time.perf_counter()
    rc = f(host.ctypes.data, B.data_ptr(), 4 << 20, 2)
    dt = time.perf_counter() - t0
    print("trial", trial, "rc", rc, "hipMemcpy took %.2f ms, saw ones: %d of %d" % (dt * 1e3, int(host.sum()), host.size))
    torch.cuda.synchronize()
