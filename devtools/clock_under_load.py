"""Runs the NN layer GEMM (1024x2048x2048) back to back for a few seconds while sampling rocm-smi's sclk / power from a
side thread: tells whether the fp32 MFMA peak (157.3 TF at 2.4 GHz) is reachable at the clock the chip actually holds."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
dev = torch.device('cuda:0')
M, N, K = 1024, 2048, 2048
A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); C = torch.empty(M, N, device=dev)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l])
        except Exception as e:  # noqa
            samples.append([repr(e)])
        time.sleep(0.3)
mode = sys.argv[1] if len(sys.argv) > 1 else "aslp"
fn = (lambda: aslp.ops.sgemm(0, 0, 1.0, A, B, 0.0, C)) if mode == "aslp" else (lambda: torch.mm(A, B, out=C))
for _ in range(10): fn()
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 4.0:
    for _ in range(200): fn()
    torch.cuda.synchronize(); n += 200
el = time.perf_counter() - t0
stop = True; th.join()
print("%s: %d GEMMs in %.2f s: %.1f us each, %.1f TFLOP/s" % (mode, n, el, el / n * 1e6, 2.0 * M * N * K * n / el / 1e12))
for s in samples[:12]: print("  ", " | ".join(s))
