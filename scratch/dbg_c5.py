import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import test_gpu_c5_shapes as T
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
m, o, emb, rows_of, dense, sparse, csparse, label, spare = T._setup(False, 3)
opt, oo = rt.Optimizer.sgd(0.05), orc.SGD(0.05)
e0 = [x.copy() for x in o.emb]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
loss = m.step(opt, dense[:K], sparse[:K], label[:K], K=K)
for s in range(K):
    lw = o.step(dense[s], csparse[s], label[s], oo)
    print("loss", loss[s], lw)
for f in range(26):
    got = emb.gather(rows_of[f].astype(np.int32))
    dg, dw = got - e0[f], o.emb[f] - e0[f]
    err = np.abs(dg - dw)
    bad_rows = np.where(err.max(1) > 3e-8)[0]
    cnt = np.array([(csparse[:K, :, f] == r).sum() for r in bad_rows[:6]])
    print(f"table {f:2d} rows {len(rows_of[f]):7d} max|dw| {np.abs(dw).max():.2e} max err {err.max():.2e} bad rows {len(bad_rows)} first {bad_rows[:6]} refcounts {cnt} "
          f"ratio {np.abs(dg).max() / max(np.abs(dw).max(), 1e-30):.3f}")
f = 0
got = emb.gather(rows_of[f].astype(np.int32)); W0 = e0[f]; want = o.emb[f]
dg = got.astype(np.float64) - W0; dw = want.astype(np.float64) - W0
err = np.abs(dg - dw)
idx = np.argsort(err.reshape(-1))[::-1][:12]
for i in idx:
    r, c = divmod(i, 128)
    nref = [(csparse[s, :, f] == r).sum() for s in range(K)]
    print(f"row {r} col {c} w0 {W0[r,c]:+.3e} d_want {dw[r,c]:+.3e} d_got {dg[r,c]:+.3e} err {err[r,c]:.2e} ulp(w) {np.spacing(np.abs(W0[r,c])):.1e} refs {nref} relerr {err[r,c]/max(abs(dw[r,c]),1e-30):.2e}")
# arbiter: the same steps in float64
from oracle.dlrm_oracle import DLRMOracle
m2, o32, emb2, rows_of2, dense2, sparse2, csparse2, label2, spare2 = T._setup(False, 3)
o64 = DLRMOracle(ln_emb=[len(x) for x in o32.emb], dtype=np.float64, seed=3, reference_compat=False, **T.CFG)
o64.emb = [x.astype(np.float64) for x in o32.emb]
for l in range(len(o32.bot)): o64.bot[l] = [o32.bot[l][0].astype(np.float64), o32.bot[l][1].astype(np.float64)]
for l in range(len(o32.top)): o64.top[l] = [o32.top[l][0].astype(np.float64), o32.top[l][1].astype(np.float64)]
oo64 = orc.SGD(0.05)
for s in range(K): o64.step(dense[s].astype(np.float64), csparse[s], label[s].astype(np.float64), oo64)
d64 = o64.emb[f] - W0
print("err of device vs f64: max %.2e ; err of f32 oracle vs f64: max %.2e" % (np.abs(dg - d64).max(), np.abs(dw - d64).max()))
for i in idx[:6]:
    r, c = divmod(i, 128)
    print(f"row {r} col {c}: d64 {d64[r,c]:+.4e} d_got {dg[r,c]:+.4e} d_want32 {dw[r,c]:+.4e}")
tot = sum((np.bincount(csparse[s, :, f], minlength=len(rows_of[f])) for s in range(K)))
rowerr = np.abs(dg - d64).max(1); roww = np.abs(d64).max(1)
for n in range(0, 9):
    sel = tot == n
    if sel.any(): print(f"refs {n}: rows {sel.sum():5d} max err vs f64 {rowerr[sel].max():.2e} median {np.median(rowerr[sel]):.2e}  max|d| {roww[sel].max():.2e}")
# per-step: which step introduces it? run only step 0 on a fresh model
r = 1291
e = (dg - d64)[r]
print("row", r, "err vector (1e-9):", np.round(e[:32] * 1e9, 1))
print("  w0[:16]", np.round(W0[r, :16], 4)); print("  d64[:16] (1e-9)", np.round(d64[r, :16] * 1e9, 1))
for s in range(K): print("  step", s, "samples", np.where(csparse[s, :, f] == r)[0])
# correlation of the error with candidate directions
Zd = None
print("  err mean %.2e std %.2e ; corr with d64 %.3f ; corr with w0 %.3f" % (e.mean(), e.std(), np.corrcoef(e, d64[r])[0, 1], np.corrcoef(e, W0[r])[0, 1]))
