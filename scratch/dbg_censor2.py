import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
for D in (16, 64, 128):
  for K in (2,):
    B = 1500
    rng = np.random.default_rng(21)
    U = rng.uniform(-.05, .05, (900, D)).astype(np.float32) * 30; V = rng.uniform(-.05, .05, (1100, D)).astype(np.float32) * 30
    b = rng.uniform(-.05, .05, (1100, 1)).astype(np.float32)
    rng = np.random.default_rng(5)
    uid = rng.integers(0, 900, (K, B)).astype(np.int32); pid = rng.integers(0, 1100, (K, B)).astype(np.int32)
    nid = rng.integers(0, 1100, (K, B)).astype(np.int32)
    tU = rt.Table(900, D).write(U); tV = rt.Table(1100, D).write(V); tb = rt.Table(1100, 1).write(b)
    loss, l2 = rt.pairwise_step("ucml", rt.Optimizer.sgd(0.01), tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=True)
    oo = orc.SGD(lr=0.01)
    for s in range(K):
        lr, _ = orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=True)
        print(D, s, loss[s], lr, abs(loss[s] - lr) / abs(lr))
    gU = tU.read(); gV = tV.read()
    eU = np.abs(gU - U).max(1); eV = np.abs(gV - V).max(1)
    print(D, "bad U rows", (eU > 1e-5).sum(), "bad V rows", (eV > 1e-5).sum(), eU.max(), eV.max())
    cnt0 = np.bincount(uid[0], minlength=900); cnt1 = np.bincount(uid[1], minlength=900)
    for r in np.nonzero(eU > 1e-5)[0][:12]:
        print("  U row", r, "count step0", cnt0[r], "step1", cnt1[r], "err", eU[r], "norm", np.linalg.norm(gU[r]))
    c0 = np.bincount(np.concatenate([pid[0], nid[0]]), minlength=1100); c1 = np.bincount(np.concatenate([pid[1], nid[1]]), minlength=1100)
    for r in np.nonzero(eV > 1e-5)[0][:12]:
        print("  V row", r, "count step0", c0[r], "step1", c1[r], "err", eV[r], "norm", np.linalg.norm(gV[r]))
