import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from openrec_amd import runtime as rt
from oracle import c_oracle
NU = NI = 40000
D, B, K = 32, 64, 700
rng = np.random.default_rng(21)
U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
nid = rng.integers(0, NI, (K, B)).astype(np.int32)
got = {}
for form in ("longgap", "plain", "dense"):
    os.environ.pop("ORX_ADAM_NO_LONGGAP", None); os.environ.pop("ORX_ADAM_DENSE", None)
    if form == "plain": os.environ["ORX_ADAM_NO_LONGGAP"] = "1"
    if form == "dense": os.environ["ORX_ADAM_DENSE"] = "1"
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    opt = rt.Optimizer.adam(0.002)
    loss, _ = rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid, K=K, B=B)
    got[form] = (loss.copy(), tU.read(), tV.read(), tb.read())
Uc, Vc, bc = U.copy(), V.copy(), b.copy()
cpu = c_oracle.PairwiseCPU("bpr", "adam", Uc, Vc, bc, lr=0.002)
ref = np.array([cpu.step(uid[s], pid[s], nid[s])[0] for s in range(K)])
for form, (loss, gU, gV, gb) in got.items():
    eU = np.abs(gU - Uc).max(axis=1) / np.abs(Uc).max(); eV = np.abs(gV - Vc).max(axis=1) / np.abs(Vc).max()
    print(form, "loss", np.abs(loss - ref).max() / np.abs(ref).max(), "U", eU.max(), int((eU > 5e-5).sum()), "V", eV.max(), int((eV > 5e-5).sum()),
          "b", np.abs(gb - bc.reshape(-1, 1)).max() / np.abs(bc).max())
for f in ("plain", "dense"):
    print("longgap vs", f, [float(np.abs(x - y).max() / np.abs(y).max()) for x, y in zip(got["longgap"][1:], got[f][1:])])
