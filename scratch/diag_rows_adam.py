import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from openrec_amd import runtime as rt, _ffi
from oracle import numpy_oracle as orc
torch.cuda.init(); dev = torch.device("cuda", 0)
ctx = rt.default_context(); lib = ctx._lib
for rows, D, n, K, use_gather in ((40000, 4, 600, 15, False), (40000, 128, 600, 15, True), (500, 16, 600, 15, True), (50, 64, 600, 40, True), (200000, 32, 3000, 60, True), (7, 8, 900, 25, False)):
    rng = np.random.default_rng(1)
    W0 = rng.uniform(-.05, .05, (rows, D)).astype(np.float32)
    t = rt.Table(rows, D, ctx).write(W0)
    opt = rt.Optimizer.adam(0.002, ctx=ctx)
    W = W0.astype(np.float64); oo = orc.AdamTFSparse(0.002)
    ids_all = rng.integers(0, rows, (K, n)).astype(np.int32); ids_all[:, :5] = -1
    g_all = rng.normal(0, 1e-3, (K, n, D)).astype(np.float32)
    out = torch.zeros((n, D), device=dev)
    for s in range(K):
        ids = torch.from_numpy(ids_all[s]).to(dev); g = torch.from_numpy(g_all[s]).to(dev)
        torch.cuda.synchronize()
        if use_gather:
            _ffi.check(lib.orx_gather_rows(ctx._h, t._h, None, ids.data_ptr(), n, out.data_ptr(), D))
            ctx.synchronize()
            m = ids_all[s] >= 0
            ge = np.abs(out.cpu().numpy()[m] - W[ids_all[s][m]]).max() / 0.05
            if ge > 1e-5: print("  gather err at step", s, ge)
        opt.advance([t])
        _ffi.check(lib.orx_apply_rows(ctx._h, opt._h, t._h, None, ids.data_ptr(), n, g.data_ptr(), D))
        ctx.synchronize()
        m = ids_all[s] >= 0
        oo.begin_step(); oo.apply(W, ids_all[s][m], g_all[s][m].astype(np.float64), key="W")
    got = t.read()
    e = np.abs(got - W).max(axis=1) / np.abs(W).max()
    print("rows", rows, "D", D, "gather", use_gather, "max err", e.max(), "bad rows", int((e > 5e-5).sum()))
