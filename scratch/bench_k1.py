import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from openrec_amd import runtime as rt
dev = torch.device("cuda", 0)
for B, NU, NI in ((65536, 1_000_000, 1_000_000), (1000, 5551, 16980)):
    U = rt.Table(NU, 64).init_uniform(seed=0); V = rt.Table(NI, 64).init_uniform(seed=1); b = rt.Table(NI, 1).init_uniform(seed=2)
    opt = rt.Optimizer.sgd(0.05)
    n = 200
    uid = torch.randint(0, NU, (n, B), device=dev, dtype=torch.int32); pid = torch.randint(0, NI, (n, B), device=dev, dtype=torch.int32)
    nid = torch.randint(0, NI, (n, B), device=dev, dtype=torch.int32)
    hu, hp, hn = uid.cpu().numpy(), pid.cpu().numpy(), nid.cpu().numpy()
    for label, ids, wl in (("device ids, no loss", (uid, pid, nid), False), ("device ids, loss", (uid, pid, nid), True), ("host ids, loss", (hu, hp, hn), True)):
        for s in range(10):
            rt.pairwise_step("bpr", opt, U, V, b, ids[0][s], ids[1][s], ids[2][s], K=1, B=B, want_loss=wl)
        U.ctx.synchronize(); t0 = time.perf_counter()
        for s in range(10, n):
            rt.pairwise_step("bpr", opt, U, V, b, ids[0][s], ids[1][s], ids[2][s], K=1, B=B, want_loss=wl)
        U.ctx.synchronize(); dt = (time.perf_counter() - t0) / (n - 10)
        print(f"B={B} K=1 per call, {label}: {dt*1e6:.1f} us/step = {B/dt/1e6:.1f} M triplets/s")
# the drop-in API (the reference example trains with Adam, tf2_examples/bpr_citeulike.py:31)
from openrec_amd.tf2 import compat as tf
from openrec_amd.tf2.recommenders import BPR
for optname in ("SGD", "SGD", "Adam"):
  for B, NU, NI in ((65536, 1_000_000, 1_000_000), (1000, 5551, 16980)):
    m = BPR(dim_user_embed=64, dim_item_embed=64, total_users=NU, total_items=NI)
    optimizer = tf.keras.optimizers.SGD(0.05) if optname == "SGD" else tf.keras.optimizers.Adam()
    def train_step(u, p, n):
        with tf.GradientTape() as tape:
            loss, l2 = m(u, p, n)
        grads = tape.gradient((loss, l2), m.trainable_variables)
        optimizer.apply_gradients(zip(grads, m.trainable_variables))
        return loss
    rng = np.random.default_rng(0)
    u = rng.integers(0, NU, (100, B)).astype(np.int32); p = rng.integers(0, NI, (100, B)).astype(np.int32); nn = rng.integers(0, NI, (100, B)).astype(np.int32)
    for s in range(36): train_step(u[s], p[s], nn[s])
    t0 = time.perf_counter()
    for s in range(36, 100): l = train_step(u[s], p[s], nn[s])
    float(l); dt = (time.perf_counter() - t0) / 64
    print(f"B={B} drop-in train_step ({optname}; tape + apply_gradients, host ids): {dt*1e6:.1f} us/step = {B/dt/1e6:.1f} M triplets/s")
