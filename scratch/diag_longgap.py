import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
from oracle.dlrm_oracle import DLRMOracle
rng = np.random.default_rng(9)
ln_emb = [6000, 5]
cfg = dict(m_spa=16, ln_emb=ln_emb, ln_bot=[8, 16], ln_top=[16, 1], dense_dim=4, reference_compat=False)
B, K = 8, 500
dense = rng.uniform(0, 2, (K, B, 4)).astype(np.float32)
sparse = np.stack([rng.integers(0, n, (K, B)) for n in ln_emb], 2).astype(np.int32)
sparse[::7, 0, 0] = 11; sparse[3, 1, 0] = 4242; sparse[K - 2, 1, 0] = 4242
label = (rng.uniform(size=(K, B)) < 0.3).astype(np.float32)
o = DLRMOracle(dtype=np.float64, seed=4, **cfg)
oo = orc.AdamTFSparse(0.01, 0.9, 0.999, 1e-7)
m = rt.DLRMModel(**cfg)
m.param("emb").write(np.concatenate(o.emb).astype(np.float32))
for nm, layers in (("bot", o.bot), ("top", o.top)):
    for l, (W, b) in enumerate(layers):
        m.param(nm + "_w", l).write(W.astype(np.float32)); m.param(nm + "_b", l).write(b.astype(np.float32).reshape(1, -1))
opt = rt.Optimizer.adam(0.01, 0.9, 0.999, 1e-7)
m.step(opt, dense.reshape(-1, 4), sparse.reshape(-1, 2), label.reshape(-1), K=K)
for s in range(K): o.step(dense[s], sparse[s], label[s], oo)
vslot = opt.slot(m.param("emb"), 1); mslot = opt.slot(m.param("emb"), 0)
wv = np.concatenate([oo.v[("emb", f)] for f in range(2)]); wm = np.concatenate([oo.m[("emb", f)] for f in range(2)])
live = np.abs(wv) > 1e-30
rel = np.where(live, np.abs(vslot - wv) / np.maximum(np.abs(wv), 1e-300), 0)
rowerr = rel.max(axis=1)
refs = {}
for s in range(K):
    for r in sparse[s, :, 0]: refs.setdefault(int(r), []).append(s)
bad = np.argsort(-rowerr)[:8]
for r in bad:
    print("row", r, "v relerr", rowerr[r], "refs", refs.get(int(r)), "ratio dev/oracle", (vslot[r] / np.maximum(wv[r], 1e-300))[:3], "implied extra steps", np.log((vslot[r] / np.maximum(wv[r], 1e-300))[0]) / np.log(0.999))
print("rows with relerr>1e-3:", int((rowerr > 1e-3).sum()), "of live rows", int(live.any(axis=1).sum()))
