import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openrec_amd import runtime as rt
counts = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = rt.DLRMModel(m_spa=128, ln_emb=counts, ln_bot=[512, 256, 128], ln_top=[1024, 1024, 512, 256, 1], dense_dim=13, reference_compat=False, fp16_mlp=(len(sys.argv) > 2 and sys.argv[2] == "fp16"))
ctx = m.ctx
rng = np.random.default_rng(0)
K = 5
dense = np.log1p(rng.integers(0, 100, (K * B, 13))).astype(np.float32)
sparse = np.stack([rng.integers(0, n, K * B) for n in counts], 1).astype(np.int32)
label = (rng.uniform(size=K * B) < 0.25).astype(np.float32)
opt = rt.Optimizer.sgd(0.01)
m.step(opt, dense, sparse, label, K=K); ctx.synchronize()
t0 = time.perf_counter(); loss = m.step(opt, dense, sparse, label, K=K); ctx.synchronize(); dt = time.perf_counter() - t0
print("DLRM step (%s MLP):" % ("fp16" if len(sys.argv) > 2 and sys.argv[2] == "fp16" else "fp32") + " %.2f ms/step (B=%d, incl. H2D staging), %.0f samples/s, loss %s" % (dt / K * 1e3, B, K * B / dt, loss[:3]))
