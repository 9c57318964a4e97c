import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
D = 64; K = 2; B = 1500
rng = np.random.default_rng(21)
U = rng.uniform(-.05, .05, (900, D)).astype(np.float32) * 30; V = rng.uniform(-.05, .05, (1100, D)).astype(np.float32) * 30
b = rng.uniform(-.05, .05, (1100, 1)).astype(np.float32)
rng = np.random.default_rng(5)
uid = rng.integers(0, 900, (K, B)).astype(np.int32); pid = rng.integers(0, 1100, (K, B)).astype(np.int32)
nid = rng.integers(0, 1100, (K, B)).astype(np.int32)
tU = rt.Table(900, D).write(U); tV = rt.Table(1100, D).write(V); tb = rt.Table(1100, 1).write(b)
U0 = U.copy()
loss, l2 = rt.pairwise_step("ucml", rt.Optimizer.sgd(0.01), tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=True)
gr = orc.ucml_grads(U, V, b, uid[0], pid[0], nid[0], 0.5)
print(type(gr), [type(x) for x in gr] if isinstance(gr, (tuple, list)) else gr.keys())
G = np.zeros_like(U); 
gu = gr[0] if isinstance(gr, (tuple, list)) else gr["gu"]
np.add.at(G, uid[0], gu)
got = tU.read()
cnt0 = np.bincount(uid[0], minlength=900); cnt1 = np.bincount(uid[1], minlength=900)
for r in np.nonzero((cnt0 >= 2) & (cnt1 == 0))[0][:6]:
    wn = U0[r] - 0.01 * G[r]
    cands = {"censor(w-lrG)": wn / max(np.linalg.norm(wn), 0.1), "w-lrG": wn, "(w-lrG)/|w|": wn / np.linalg.norm(U0[r]),
             "censor(w)-lrG": U0[r] / np.linalg.norm(U0[r]) - 0.01 * G[r], "w": U0[r], "censor(w)": U0[r] / np.linalg.norm(U0[r])}
    print("row", r, "cnt", cnt0[r], "|got|", np.linalg.norm(got[r]), {k: float(np.abs(got[r] - v).max()) for k, v in cands.items()})
print("---- fits got = alpha*w + beta*G")
for r in np.nonzero((cnt0 >= 2) & (cnt1 == 0))[0][:6]:
    A = np.stack([U0[r], G[r]], 1)
    coef, res, *_ = np.linalg.lstsq(A, got[r], rcond=None)
    print("row", r, "cnt", cnt0[r], "alpha", coef[0], "beta", coef[1], "resid", np.abs(A @ coef - got[r]).max(), "1/|w-lrG|", 1 / np.linalg.norm(U0[r] - 0.01 * G[r]))
