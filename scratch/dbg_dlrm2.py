import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
from oracle.dlrm_oracle import DLRMOracle
name, compat, f16 = sys.argv[1], bool(int(sys.argv[2])), bool(int(sys.argv[3]))
rng = np.random.default_rng(11)
ln_emb = [50, 300, 7, 1000, 33]
cfg = {"narrow": dict(m_spa=16, ln_bot=[64, 16], ln_top=[128, 64, 1], B=700),
       "wide": dict(m_spa=32, ln_bot=[512, 256, 32], ln_top=[1024, 512, 256, 1], B=2304),
       "ragged": dict(m_spa=24, ln_bot=[100, 36, 24], ln_top=[136, 100, 40, 1], B=517)}[name]
B = cfg.pop("B"); cfg.update(ln_emb=ln_emb, dense_dim=13)
o = DLRMOracle(dtype=np.float32, seed=5, reference_compat=compat, **cfg)
m = rt.DLRMModel(reference_compat=compat, fp16_mlp=f16, **cfg)
m.param("emb").write(np.concatenate(o.emb))
for nm, layers in (("bot", o.bot), ("top", o.top)):
    for l, (W, b) in enumerate(layers):
        b[:] = rng.normal(size=b.shape).astype(np.float32) * 0.1
        m.param(nm + "_w", l).write(W); m.param(nm + "_b", l).write(b.reshape(1, -1))
opt, oo = rt.Optimizer.sgd(0.02), orc.SGD(0.02)
for step in range(3):
    dense = np.log1p(rng.integers(0, 100, (B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, B) for n in ln_emb], 1).astype(np.int32)
    label = (rng.uniform(size=B) < 0.3).astype(np.float32)
    before = [(nm, l, o.__dict__[nm][l][0].copy(), o.__dict__[nm][l][1].copy()) for nm in ("bot", "top") for l in range(len(o.__dict__[nm]))]
    l16 = m.step(opt, dense, sparse, label)[0]; l32 = o.step(dense, sparse, label, oo)
    out = [f"step {step} loss rel {abs(l16-l32)/abs(l32):.1e}"]
    for nm, l, W0, b0 in before:
        W1, b1 = o.__dict__[nm][l]
        dW, db = m.param(nm + "_w", l).read() - W0, m.param(nm + "_b", l).read().reshape(-1) - b0.reshape(-1)
        rW, rb = W1 - W0, (b1 - b0).reshape(-1)
        out.append(f"{nm}{l} W {np.abs(dW - rW).max() / np.abs(rW).max():.3f} b {np.abs(db - rb).max() / (np.abs(rb).max() + 1e-30):.3f}")
        m.param(nm + "_w", l).write(W1); m.param(nm + "_b", l).write(b1.reshape(1, -1))
    m.param("emb").write(np.concatenate(o.emb))
    print(" | ".join(out))
