import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
from oracle.dlrm_oracle import DLRMOracle
m_spa, beta2 = 4, 0.999
rng = np.random.default_rng(5)
ln_emb = [3, 40, 30000, 700, 9000, 20]
off = np.concatenate([[0], np.cumsum(ln_emb)])
cfg = dict(m_spa=m_spa, ln_emb=ln_emb, ln_bot=[16, m_spa], ln_top=[64, 32, 1], dense_dim=13, reference_compat=False)
B, K = 96, 15
dense = np.log1p(rng.integers(0, 100, (K, B, 13))).astype(np.float32)
sparse = np.stack([rng.integers(0, n, (K, B)) for n in ln_emb], 2).astype(np.int32)
label = (rng.uniform(size=(K, B)) < 0.25).astype(np.float32)
o = DLRMOracle(dtype=np.float64, seed=2, **cfg)
oo = orc.AdamTFSparse(0.002, 0.9, beta2, 1e-7)
start = [np.concatenate(o.emb).astype(np.float32)] + [(W.astype(np.float32), b.astype(np.float32)) for W, b in o.bot + o.top]
for s in range(K): o.step(dense[s], sparse[s], label[s], oo)
want = np.concatenate(o.emb)
cnt = np.zeros(off[-1], int); lastref = np.full(off[-1], -1); dupstep = np.zeros(off[-1], int)
for s in range(K):
    for f in range(len(ln_emb)):
        r = off[f] + sparse[s, :, f]
        u, c = np.unique(r, return_counts=True)
        cnt[u] += 1; lastref[u] = s; dupstep[u[c > 1]] += 1
for Ksplit in ((15,),):
    m = rt.DLRMModel(**cfg)
    m.param("emb").write(start[0])
    for nm, n0, c_ in (("bot", 1, len(o.bot)), ("top", 1 + len(o.bot), len(o.top))):
        for l in range(c_):
            m.param(nm + "_w", l).write(start[n0 + l][0]); m.param(nm + "_b", l).write(start[n0 + l][1].reshape(1, -1))
    opt = rt.Optimizer.adam(0.002, 0.9, beta2, 1e-7)
    s0 = 0
    for kk in Ksplit:
        m.step(opt, dense[s0:s0 + kk].reshape(-1, 13), sparse[s0:s0 + kk].reshape(-1, len(ln_emb)), label[s0:s0 + kk].reshape(-1), K=kk); s0 += kk
    emb = m.param("emb").read()
    e = np.abs(emb - want).max(axis=1) / np.abs(want).max()
    bad = e > 5e-5
    print("split", Ksplit[:2], "bad rows", bad.sum(), "of touched", (cnt > 0).sum(), "bad&untouched", (bad & (cnt == 0)).sum())
    for c in range(1, 5): print("  steps-referenced", c, "rows", (cnt == c).sum(), "bad", (bad & (cnt == c)).sum())
    pass
    print("  dup-in-step rows", (dupstep > 0).sum(), "bad", (bad & (dupstep > 0)).sum())
