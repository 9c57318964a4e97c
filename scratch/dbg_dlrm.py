import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
from oracle.dlrm_oracle import DLRMOracle
rng = np.random.default_rng(4)
ln_emb = [50, 300, 7, 1000]
cfg = dict(m_spa=32, ln_emb=ln_emb, ln_bot=[96, 32], ln_top=[200, 72, 1], dense_dim=13)
o = DLRMOracle(dtype=np.float32, seed=5, reference_compat=False, **cfg)
m = rt.DLRMModel(reference_compat=False, fp16_mlp=True, **cfg)
m.param("emb").write(np.concatenate(o.emb))
for nm, layers in (("bot", o.bot), ("top", o.top)):
    for l, (W, b) in enumerate(layers):
        b[:] = rng.normal(size=b.shape).astype(np.float32) * 0.1
        m.param(nm + "_w", l).write(W); m.param(nm + "_b", l).write(b.reshape(1, -1))
B = 333
dense = np.log1p(rng.integers(0, 100, (B, 13))).astype(np.float32)
sparse = np.stack([rng.integers(0, n, B) for n in ln_emb], 1).astype(np.int32)
label = (rng.uniform(size=B) < 0.3).astype(np.float32)
p16, p32 = m.inference(dense, sparse), o.inference(dense, sparse)
print("pred diff", np.abs(p16 - p32).max())
p16b = m.inference(dense, sparse)
print("pred repeat diff", np.abs(p16 - p16b).max())
l16 = m.step(rt.Optimizer.sgd(0.1), dense, sparse, label)[0]
l32 = o.step(dense, sparse, label, orc.SGD(0.1))
print("loss", l16, l32, "mse from p16", np.mean((p16 - label) ** 2), "from p32", np.mean((p32 - label) ** 2))
for key, dev, ref in ((("top", 0), m.param("top_w", 0).read(), o.top[0][0]), (("top", 1), m.param("top_w", 1).read(), o.top[1][0]), (("top", 2), m.param("top_w", 2).read(), o.top[2][0]),
                      (("bot", 0), m.param("bot_w", 0).read(), o.bot[0][0]), (("bot", 1), m.param("bot_w", 1).read(), o.bot[1][0])):
    print(key, "max abs diff", np.abs(dev - ref).max(), "max |ref|", np.abs(ref).max())
for nm, layers in (("bot", o.bot), ("top", o.top)):
    for l, (W, b) in enumerate(layers):
        print(nm, l, "bias diff", np.abs(m.param(nm + "_b", l).read().reshape(-1) - b).max())
