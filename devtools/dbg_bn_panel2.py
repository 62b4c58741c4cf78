"""BN backward with the Sigmoid folded in vs Sigmoid backward first: dshift / dscale / in_diff must agree bit for bit"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import aslp_import
aslp = aslp_import.load(); aslp.ops.use_torch_stream()
from importlib import import_module
L = import_module("kaldi-aslp_amd._lib") if False else None
lib = aslp.ops.lib; ptr = aslp.ops.ptr; dim = aslp.ops.dim
import ctypes as C
from importlib import util
MD = type(dim(torch.zeros(1, 1)))
lib.aslp_bn_backward_act.restype = None
lib.aslp_bn_backward_act.argtypes = [C.c_void_p, MD, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
for rows, cols in [(256, 128), (1024, 2048)]:
    g = torch.Generator(device="cpu").manual_seed(1)
    dy = torch.randn(rows, cols, generator=g).to(dev)
    xhat = torch.randn(rows, cols, generator=g).to(dev)
    y = torch.sigmoid(torch.randn(rows, cols, generator=g)).to(dev)
    scale = (torch.rand(cols, generator=g) + 0.5).to(dev)
    inv = (torch.rand(cols, generator=g) + 0.5).to(dev)
    res = []
    for fused in (True, False):
        xh = xhat.clone(); ds = torch.zeros(cols, device=dev); dsh = torch.zeros(cols, device=dev); ind = torch.empty(rows, cols, device=dev)
        if fused:
            lib.aslp_bn_backward_act(None, dim(dy), ptr(dy), dim(dy).stride, ptr(xh), dim(xh).stride, ptr(scale), None, ptr(inv), ptr(ds), ptr(dsh), 0.0,
                                     ptr(ind), dim(ind).stride, ptr(y), dim(y).stride)
        else:
            d = torch.empty_like(dy)
            aslp.ops.diff_sigmoid(d, y, dy)
            lib.aslp_bn_backward_act(None, dim(d), ptr(d), dim(d).stride, ptr(xh), dim(xh).stride, ptr(scale), None, ptr(inv), ptr(ds), ptr(dsh), 0.0,
                                     ptr(ind), dim(ind).stride, None, 0)
        aslp.ops.check_error(); torch.cuda.synchronize()
        res.append((ds.cpu().numpy(), dsh.cpu().numpy(), ind.cpu().numpy(), xh.cpu().numpy()))
    for name, a, b in zip(("dscale", "dshift", "in_diff", "D"), res[0], res[1]):
        print(rows, cols, name, "equal" if np.array_equal(a, b) else "DIFFER max %.3g" % np.abs(a - b).max())
