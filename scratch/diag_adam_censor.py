import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
NU, NI, B, K, D = 3000, 2500, 4096, 12, 128
rng = np.random.default_rng(D + K)
U32 = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V32 = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
b32 = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
uid = rng.integers(0, NU, (3 * K, B)).astype(np.int32); pid = rng.integers(0, NI, (3 * K, B)).astype(np.int32)
nid = rng.integers(0, NI, (3 * K, B)).astype(np.int32)
res = {}
for form in ("lazy", "dense", "lazy_nofuse"):
    os.environ.pop("ORX_ADAM_DENSE", None); os.environ.pop("ORX_FORCE_FALLBACK", None)
    if form == "dense": os.environ["ORX_ADAM_DENSE"] = "1"
    if form == "lazy_nofuse": os.environ["ORX_FORCE_FALLBACK"] = "4"
    tU = rt.Table(NU, D).write(U32); tV = rt.Table(NI, D).write(V32); tb = rt.Table(NI, 1).write(b32)
    opt = rt.Optimizer.adam(0.002, 0.9, 0.999, 1e-7)
    snaps = []
    for rep in range(3):
        sl = slice(rep * K, (rep + 1) * K)
        rt.pairwise_step("ucml", opt, tU, tV, tb, uid[sl], pid[sl], nid[sl], K=K, B=B, censor=True)
        snaps.append((tU.read().copy(), tV.read().copy()))
    res[form] = snaps
U, V, b = U32.astype(np.float64), V32.astype(np.float64), b32.astype(np.float64)
oo = orc.AdamTFSparse(0.002, 0.9, 0.999, 1e-7)
ref = []
for s in range(3 * K):
    orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=True)
    if (s + 1) % K == 0: ref.append((U.copy(), V.copy()))
for form, snaps in res.items():
    for r in range(3):
        eu = np.abs(snaps[r][0] - ref[r][0]); ev = np.abs(snaps[r][1] - ref[r][1])
        print(form, r, "U err", eu.max() / np.abs(ref[r][0]).max(), "rows>1e-5:", int((eu.max(1) > 1e-5 * np.abs(ref[r][0]).max()).sum()),
              "V err", ev.max() / np.abs(ref[r][1]).max(), "rows:", int((ev.max(1) > 1e-5 * np.abs(ref[r][1]).max()).sum()))
