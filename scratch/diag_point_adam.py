import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
def rel(a, b): return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
for model, NU, NI, B, K, D in [("gmf", 40000, 30000, 4096, 16, 64), ("wrmf", 3000, 2500, 8192, 10, 32), ("wrmf", 90000, 70000, 2048, 24, 128)]:
    rng = np.random.default_rng(K + D)
    U32 = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V32 = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b32 = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32); w32 = rng.uniform(-.3, .3, (D, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (2 * K, B)).astype(np.int32); iid = rng.integers(0, NI, (2 * K, B)).astype(np.int32)
    lab = (rng.uniform(size=(2 * K, B)) < 0.4).astype(np.float32)
    U, V, b, w = (x.astype(np.float64) for x in (U32, V32, b32, w32))
    oo = orc.AdamTFSparse(0.002, 0.9, 0.999, 1e-7)
    for s in range(K):
        if model == "gmf": orc.gmf_step(U, V, b, w, uid[s], iid[s], lab[s], oo)
        else: orc.wrmf_step(U, V, b, uid[s], iid[s], lab[s], oo, a=1.5, b_w=0.7)
    for form in ("lazy", "dense"):
        os.environ.pop("ORX_ADAM_DENSE", None)
        if form == "dense": os.environ["ORX_ADAM_DENSE"] = "1"
        tU = rt.Table(NU, D).write(U32); tV = rt.Table(NI, D).write(V32); tb = rt.Table(NI, 1).write(b32)
        tw = rt.Table(D, 1).write(w32) if model == "gmf" else None
        opt = rt.Optimizer.adam(0.002, 0.9, 0.999, 1e-7)
        rt.pointwise_step(model, opt, tU, tV, tb, tw, uid[:K], iid[:K], lab[:K], K=K, B=B, a=1.5, b_w=0.7)
        eU = np.abs(tU.read() - U); 
        print(model, D, form, "U", rel(tU.read(), U), "rows>5e-5:", int((eU.max(1) > 5e-5 * np.abs(U).max()).sum()), "V", rel(tV.read(), V), "b", rel(tb.read(), b),
              "mV", rel(opt.slot(tV, 0), oo.m["V"]), "w", rel(tw.read(), w) if tw is not None else None)
