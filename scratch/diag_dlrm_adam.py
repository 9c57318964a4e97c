import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
from oracle.dlrm_oracle import DLRMOracle
def rel(a, b): return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))
for m_spa, beta2 in ((4, 0.999), (128, 0.999), (32, 0.95)):
    rng = np.random.default_rng(5)
    ln_emb = [3, 40, 30000, 700, 9000, 20]
    cfg = dict(m_spa=m_spa, ln_emb=ln_emb, ln_bot=[16, m_spa], ln_top=[64, 32, 1], dense_dim=13, reference_compat=False)
    B, K = 96, 40
    dense = np.log1p(rng.integers(0, 100, (K, B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, (K, B)) for n in ln_emb], 2).astype(np.int32)
    label = (rng.uniform(size=(K, B)) < 0.25).astype(np.float32)
    o = DLRMOracle(dtype=np.float64, seed=2, **cfg)
    oo = orc.AdamTFSparse(0.002, 0.9, beta2, 1e-7)
    start = [np.concatenate(o.emb).astype(np.float32)] + [(W.astype(np.float32), b.astype(np.float32)) for W, b in o.bot + o.top]
    refs = {}
    ref_loss = []
    for s in range(K):
        ref_loss.append(o.step(dense[s], sparse[s], label[s], oo))
        if s in (14, 24, K - 1):
            refs[s] = (o.inference(dense[0], sparse[0]).copy(), np.concatenate(o.emb).copy())
    for form in ("lazy", "dense"):
        os.environ.pop("ORX_ADAM_DENSE", None)
        if form == "dense": os.environ["ORX_ADAM_DENSE"] = "1"
        m = rt.DLRMModel(**cfg)
        m.param("emb").write(start[0])
        for nm, n0, cnt in (("bot", 1, len(o.bot)), ("top", 1 + len(o.bot), len(o.top))):
            for l in range(cnt):
                m.param(nm + "_w", l).write(start[n0 + l][0]); m.param(nm + "_b", l).write(start[n0 + l][1].reshape(1, -1))
        opt = rt.Optimizer.adam(0.002, 0.9, beta2, 1e-7)
        loss = []
        for lo, hi in ((0, 15), (15, 25), (25, K)):
            loss += list(m.step(opt, dense[lo:hi].reshape(-1, 13), sparse[lo:hi].reshape(-1, len(ln_emb)), label[lo:hi].reshape(-1), K=hi - lo))
            pred = m.inference(dense[0], sparse[0]); emb = m.param("emb").read()
            e = np.abs(emb - refs[hi - 1][1]).max(axis=1) / np.abs(refs[hi - 1][1]).max()
            print(m_spa, beta2, form, "after", hi, "pred err", rel(pred, refs[hi - 1][0]), "emb err", e.max(), "rows>5e-5", int((e > 5e-5).sum()),
                  "loss err", float(np.abs(np.array(loss) - np.array(ref_loss[:hi])).max() / np.abs(ref_loss).max()))
