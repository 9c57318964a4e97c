import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
D = 128; K, B = 4, 1500
for fb in ["0", "2", "1", "4"]:
    os.environ["ORX_FORCE_FALLBACK"] = fb
    rng = np.random.default_rng(21)
    U = rng.uniform(-.05, .05, (900, D)).astype(np.float32) * 30; V = rng.uniform(-.05, .05, (1100, D)).astype(np.float32) * 30
    b = rng.uniform(-.05, .05, (1100, 1)).astype(np.float32)
    rng = np.random.default_rng(5)
    uid = rng.integers(0, 900, (K, B)).astype(np.int32); pid = rng.integers(0, 1100, (K, B)).astype(np.int32)
    nid = rng.integers(0, 1100, (K, B)).astype(np.int32)
    nid[:, :40] = pid[:, :40]
    tU = rt.Table(900, D).write(U); tV = rt.Table(1100, D).write(V); tb = rt.Table(1100, 1).write(b)
    loss, l2 = rt.pairwise_step("ucml", rt.Optimizer.sgd(0.01), tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=True)
    oo = orc.SGD(lr=0.01)
    for s in range(K):
        lr, _ = orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=True)
        print(fb, s, loss[s], lr, abs(loss[s] - lr) / abs(lr))
    eU = np.abs(tU.read() - U).max(1); eV = np.abs(tV.read() - V).max(1)
    print(fb, "bad U rows", (eU > 1e-5).sum(), "bad V rows", (eV > 1e-5).sum(), eU.max(), eV.max())
    bad = np.nonzero(eV > 1e-5)[0][:5]
    for r in bad:
        print("  V row", r, "in p", (pid[K-1] == r).sum(), "in n", (nid[K-1] == r).sum(), "norm got", np.linalg.norm(tV.read()[r]), "want", np.linalg.norm(V[r]))
