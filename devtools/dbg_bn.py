import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import aslp_import; aslp = aslp_import.load(); aslp.ops.use_torch_stream()
import oracle_lib as o
dev = torch.device('cuda:0')
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for rows, cols in [(129, 260), (129, 256), (128, 260), (130,260), (129,264)]:
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((rows, cols)) * 2 + 0.5).astype(np.float32)
    dy = rng.standard_normal((rows, cols)).astype(np.float32)
    bn = o.Bn(cols)
    bn.scale[:] = rng.uniform(0.5, 1.5, cols); bn.shift[:] = rng.standard_normal(cols)
    bn.dscale[:] = rng.standard_normal(cols); bn.dshift[:] = rng.standard_normal(cols)
    dsc0, dsh0 = bn.dscale.copy(), bn.dshift.copy()
    ref_out = bn.propagate(x)
    xd = T(x); out, xhat = torch.empty_like(xd), torch.empty_like(xd)
    scale, shift = T(bn.scale), T(bn.shift)
    mean, inv = torch.empty(cols, device=dev), torch.empty(cols, device=dev)
    aslp.ops.bn_forward(xd, out, xhat, scale, shift, mean, inv)
    ref_idf = bn.backpropagate(x, dy, 0.9)
    dsc, dsh = T(dsc0), T(dsh0); idf = torch.empty_like(xd)
    aslp.ops.bn_backward(xd, T(dy), xhat, scale, mean, inv, dsc, dsh, 0.9, idf)
    e = np.abs(idf.cpu().numpy() - ref_idf)
    print(rows, cols, 'rel', o.rel_err(idf.cpu().numpy(), ref_idf), 'max at', np.unravel_index(e.argmax(), e.shape), e.max(),
          'dsc', o.rel_err(dsc.cpu().numpy(), bn.dscale), 'dsh', o.rel_err(dsh.cpu().numpy(), bn.dshift))
    bad = np.where(e.max(0) > 1e-3)[0]
    print('  bad cols', bad[:20], len(bad))
