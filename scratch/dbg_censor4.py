import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
D = 64; K = 2; B = 1500
for dbg in ["14", "12", "6", "10"]:
    os.environ["ORX_DBG"] = dbg
    rng = np.random.default_rng(21)
    U = rng.uniform(-.05, .05, (900, D)).astype(np.float32) * 30; V = rng.uniform(-.05, .05, (1100, D)).astype(np.float32) * 30
    b = rng.uniform(-.05, .05, (1100, 1)).astype(np.float32)
    rng = np.random.default_rng(5)
    uid = rng.integers(0, 900, (K, B)).astype(np.int32); pid = rng.integers(0, 1100, (K, B)).astype(np.int32)
    nid = rng.integers(0, 1100, (K, B)).astype(np.int32)
    tU = rt.Table(900, D).write(U); tV = rt.Table(1100, D).write(V); tb = rt.Table(1100, 1).write(b)
    loss, l2 = rt.pairwise_step("ucml", rt.Optimizer.sgd(0.01), tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=True)
    got = tU.read(); gotV = tV.read()
    if dbg == "14":      # no censor anywhere: must equal the plain steps
        oo = orc.SGD(lr=0.01)
        for s in range(K):
            orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=False)
        print(dbg, "vs no-censor oracle:", np.abs(got - U).max(), np.abs(gotV - V).max())
    else:
        nU = np.linalg.norm(got, axis=1); nV = np.linalg.norm(gotV, axis=1)
        print(dbg, "norm histogram U", np.histogram(nU, [0, .5, .99, 1.01, 2, 100])[0], "V", np.histogram(nV, [0, .5, .99, 1.01, 2, 100])[0])
