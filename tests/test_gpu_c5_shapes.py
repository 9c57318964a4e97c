"""BASELINE.json's DLRM configuration at its real sizes on one GPU: the 26 Criteo-Kaggle tables
(33.8 M rows x 128 = 17.3 GB, combined row offsets up to 3.4e7), bottom 13-512-256-128, top 479-1024-1024-512-256-1
(recommenders/dlrm.py:8-100 with tf2_examples/dlrm_criteo.py's train step).  The small-shape tests cannot see what these
sizes exercise: every tile shape of the MLP products at full width, the 7 tables of 3..27 rows that take the one-hot MFMA
gradient sums next to tables of 1e7 rows that take the scatter-add, the interaction reading rows 1.7e10 bytes into the table.
The oracle runs on the COMPACT problem (a step depends only on the rows it references: they are gathered before the steps, the
ids renumbered densely per table, oracle/dlrm_oracle.py stepped on the small tables); a sample of unreferenced rows must keep
its exact bits.  Exact fp32 mode: the parity bar (1e-5 on loss and tables, per-element update check -- see update_check).  fp16-MLP mode: tracks to fp16 accuracy."""
import numpy as np
import pytest

from conftest import TOL

pytestmark = [pytest.mark.gpu, pytest.mark.fp32_tie]
COUNTS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10,
          5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
CFG = dict(m_spa=128, ln_bot=[512, 256, 128], ln_top=[1024, 1024, 512, 256, 1], dense_dim=13)
B, K = 2048, 2


def update_check(W0, got, want, steps, what, frac=2e-2, worst=0.1):
    """conftest.delta_check for a model with relu layers at this size.  Every element's CHANGE is held to
    1e-5 |d_want| + (steps + 1) ulp32 as there -- but a handful of the 7.6 M relu pre-activations of a step lie within fp32
    summation noise of zero (|z| < 1e-7 on values of ~0.3: about two per step), where two correct fp32 implementations
    pick different sides; such a unit moves its sample's 26 embedding rows by a few per cent of that sample's share and
    every dense gradient below it by 1 / B.  (Seen: one row of 1380 in table 0 off by 3 % of its update while the fp32 and
    fp64 oracles agree -- scratch/dbg_c5.py.)  So, for an embedding table: at most 2 % of the elements outside the strict
    bound, none by more than 10 % of the table's largest update, and the projection <d_got, d_want> / <d_want, d_want> within
    2e-3 of 1.  For a dense parameter (frac = 1: every element carries the 1 / B share of a flipped sample, ~7e-4 of its own
    update at the lowest layer): no element off by more than 0.5 % of the largest update, same projection bound."""
    d_got = got.astype(np.float64) - W0
    d_want = want.astype(np.float64) - W0
    dmax = float(np.abs(d_want).max())
    ulp = np.spacing(np.maximum(np.maximum(np.abs(want), np.abs(W0)), np.float32(dmax))).astype(np.float64)
    err = np.abs(d_got - d_want)
    bad = err > 1e-5 * np.abs(d_want) + (steps + 1) * ulp
    assert bad.mean() <= frac, f"{what}: {bad.mean():.2%} of the elements beyond the update bound"
    assert err.max() <= worst * dmax + (steps + 1) * float(ulp.max()), f"{what}: worst element off by {err.max():.3g}, largest update {dmax:.3g}"
    den = float((d_want ** 2).sum())
    if den > 0:
        coef = float((d_got * d_want).sum() / den)
        assert abs(coef - 1.0) <= 2e-3, f"{what}: update scaled by {coef:.5f}"


def _setup(fp16, seed):
    from openrec_amd import runtime as rt
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(seed)
    dense = np.log1p(rng.integers(0, 100, (K, B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, c, (K, B)) for c in COUNTS], 2).astype(np.int32)
    sparse[:, 0, :] = [c - 1 for c in COUNTS]                 # the last row of every table ...
    sparse[:, 1, :] = [c - 1 for c in COUNTS]                 # ... twice in a step, and in every step
    label = (rng.uniform(size=(K, B)) < 0.25).astype(np.float32)
    m = rt.DLRMModel(ln_emb=COUNTS, reference_compat=False, fp16_mlp=fp16, **CFG)
    uniq = [np.unique(sparse[:, :, f]) for f in range(len(COUNTS))]
    o = DLRMOracle(ln_emb=[len(u) for u in uniq], dtype=np.float32, seed=seed, reference_compat=False, **CFG)
    offs = np.concatenate([[0], np.cumsum(COUNTS)[:-1]]).astype(np.int64)
    emb = m.param("emb")
    rows_of = [(offs[f] + uniq[f]).astype(np.int64) for f in range(len(COUNTS))]
    assert rows_of[-1].max() < 2 ** 31
    for f in range(len(COUNTS)):
        o.emb[f] = emb.gather(rows_of[f].astype(np.int32))
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            b[:] = rng.normal(size=b.shape).astype(np.float32) * 0.05
            m.param(nm + "_w", l).write(W); m.param(nm + "_b", l).write(b.reshape(1, -1))
    csparse = np.stack([np.searchsorted(uniq[f], sparse[:, :, f]) for f in range(len(COUNTS))], 2).astype(np.int32)
    spare = np.setdiff1d(rng.integers(0, int(np.sum(COUNTS)), 8192), np.concatenate(rows_of)).astype(np.int32)
    return m, o, emb, rows_of, dense, sparse, csparse, label, spare


@pytest.mark.parametrize("optname", ["sgd", "adagrad"])
def test_c5_shapes_exact_mode(optname):
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    m, o, emb, rows_of, dense, sparse, csparse, label, spare = _setup(False, 3)
    opt, oo = ((rt.Optimizer.sgd(0.05), orc.SGD(0.05)) if optname == "sgd" else (rt.Optimizer.adagrad(0.05, 0.1, 1e-7), orc.Adagrad(0.05, 0.1, 1e-7)))
    e0 = [x.copy() for x in o.emb]
    d0 = {(nm, l): (W.copy(), b.copy()) for nm, layers in (("bot", o.bot), ("top", o.top)) for l, (W, b) in enumerate(layers)}
    spare0 = emb.gather(spare)
    p = m.inference(dense[0], sparse[0]); pr = o.inference(dense[0], csparse[0])
    assert np.abs(p - pr).max() <= 2e-6
    loss = m.step(opt, dense, sparse, label, K=K)
    for s in range(K):
        lw = o.step(dense[s], csparse[s], label[s], oo)
        assert abs(loss[s] - lw) <= TOL * abs(lw), (s, loss[s], lw)
    for f in range(len(COUNTS)):
        got = emb.gather(rows_of[f].astype(np.int32))
        assert np.abs(got - o.emb[f]).max() <= TOL * np.abs(o.emb[f]).max(), f
        # scatter_add puts every occurrence into the fp32 row one after the other (the oracle, like TF on the CPU, in index
        # order; the device sums a tiny table's occurrences first): a row with n references takes n roundings of ulp(|w|) / 2,
        # a random walk of ~sqrt(n) of them -- 1374 references on the 3-row table: 7e-8 seen, against updates of 1.8e-5.
        # The rounding allowance of the update check scales with 4 sqrt(n) for the table's busiest row.
        refmax = max(int(np.bincount(csparse[s, :, f]).max()) for s in range(K))
        update_check(e0[f], got, o.emb[f], steps=K * int(np.ceil(4 * np.sqrt(refmax))), what=f"C5 {optname} table {f}")
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            gW, gb = m.param(nm + "_w", l).read(), m.param(nm + "_b", l).read().reshape(-1)
            assert np.abs(gW - W).max() <= TOL * np.abs(W).max() and np.abs(gb - b).max() <= TOL * max(np.abs(b).max(), 1e-3), (nm, l)
            update_check(d0[(nm, l)][0], gW, W, steps=K, what=f"C5 {optname} {nm}_w{l}", frac=1.0, worst=5e-3)
    assert np.array_equal(emb.gather(spare), spare0)


def test_c5_shapes_fp16_mode_tracks_the_oracle():
    """the performance mode at full size: loss to 5e-3, every update of the step to fp16 accuracy (see test_gpu_dlrm.py)"""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    m, o, emb, rows_of, dense, sparse, csparse, label, spare = _setup(True, 4)
    opt, oo = rt.Optimizer.sgd(0.02), orc.SGD(0.02)
    e0 = [x.copy() for x in o.emb]
    d0 = {(nm, l): (W.copy(), b.copy()) for nm, layers in (("bot", o.bot), ("top", o.top)) for l, (W, b) in enumerate(layers)}
    spare0 = emb.gather(spare)
    assert np.abs(m.inference(dense[0], sparse[0]) - o.inference(dense[0], csparse[0])).max() < 5e-3
    loss = m.step(opt, dense[0], sparse[0], label[0])[0]
    lw = o.step(dense[0], csparse[0], label[0], oo)
    assert abs(loss - lw) < 5e-3 * abs(lw)
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            W0, b0 = d0[(nm, l)]
            dW, rW = m.param(nm + "_w", l).read() - W0, W - W0
            db, rb = m.param(nm + "_b", l).read().reshape(-1) - b0, b - b0
            assert np.abs(dW - rW).max() < 0.06 * np.abs(rW).max() + 1e-8, (nm, l, "W")
            assert np.abs(db - rb).max() < 0.06 * np.abs(rb).max() + 1e-8, (nm, l, "b")
    for f in (0, 2, 8, 13, 25):                              # a mid-size, the largest, the 3-row, the 27-row and the last table
        du, dr = emb.gather(rows_of[f].astype(np.int32)) - e0[f], o.emb[f] - e0[f]
        assert np.abs(du - dr).max() < 0.06 * np.abs(dr).max() + 3e-8, f      # (+ a few ulp of |w| = 0.05: the updates are ~1e-7)
    assert np.array_equal(emb.gather(spare), spare0)
