import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openrec_amd.tf2.compat import tf, optimizers
from openrec_amd.tf2.recommenders import DLRM
rng = np.random.default_rng(0)
counts = [int(x) for x in rng.integers(3, 5000, 26)]
for B, optname in ((1024, "SGD"), (8192, "SGD"), (1024, "Adam")):      # dlrm_criteo.py:31 trains with Adam
    m = DLRM(m_spa=4, ln_emb=counts, ln_bot=[8, 4], ln_top=[128, 64, 1], reference_compat=False)
    opt = optimizers.SGD(0.01) if optname == "SGD" else optimizers.Adam()
    dense = np.log1p(rng.integers(0, 100, (B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, c, B) for c in counts], 1).astype(np.int32)
    label = (rng.random(B) < 0.3).astype(np.float32)
    def step():
        with tf.GradientTape() as tape:
            l = m(dense, sparse, label)
        opt.apply_gradients(zip(tape.gradient(l, m.trainable_variables), m.trainable_variables))
        return l
    for _ in range(70): step()          # past the first full queue: buffers sized for K = 32
    float(step())
    t0 = time.perf_counter()
    for _ in range(100): l = step()
    float(l); dt = (time.perf_counter() - t0) / 100
    print(f"DLRM drop-in train_step B={B} {optname}: {dt*1e6:.0f} us/step = {B/dt/1e6:.2f} M samples/s")
from openrec_amd import runtime as rt
for B in (1024,):
    m = rt.DLRMModel(m_spa=4, ln_emb=counts, ln_bot=[8, 4], ln_top=[128, 64, 1], dense_dim=13, reference_compat=False)
    opt = rt.Optimizer.sgd(0.01)
    K = 200
    dense = np.log1p(rng.integers(0, 100, (K * B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, c, K * B) for c in counts], 1).astype(np.int32)
    label = (rng.random(K * B) < 0.3).astype(np.float32)
    m.step(opt, dense, sparse, label, K=K); m.ctx.synchronize()
    t0 = time.perf_counter(); m.step(opt, dense, sparse, label, K=K); m.ctx.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"DLRM K-step call (host batches) B={B}: {dt*1e6:.0f} us/step")
