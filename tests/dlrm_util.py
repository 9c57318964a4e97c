"""Shared pieces of the DLRM parity tests.

Strict parity of a relu network needs batches that stay away from its discontinuities: a pre-activation within fp32 summation
noise of zero goes to one side in the oracle and possibly to the other on the device (another order of the same additions), and
that sample's share of every gradient below then differs by whole per cents -- for ANY two correct fp32 implementations.
`draw_batch` therefore draws candidates and keeps the samples whose every relu pre-activation is at least `delta` (relative to
the sum of the magnitudes it was added up from) away from zero under the oracle's CURRENT parameters; the dropped ones are
counted and reported.  Everything that remains is held to 1e-5 (conftest.TOL), on the parameter UPDATES.
"""
import numpy as np

DELTA = 1e-6          # fp32 dot products of up to ~1000 terms agree to a few 1e-7 of sum |terms| between summation orders


def gen_dense(rng, n, dim=13):
    return np.log1p(rng.integers(0, 100, (n, dim))).astype(np.float32)      # tf2_examples/dataloader.py:72


def draw_batch(o, rng, B, ln_emb, delta=DELTA, label_p=0.25, dense=gen_dense, dense_dim=13, fix=None, stats=None):
    """B samples (dense, sparse, label) none of which lies within `delta` of a discontinuity of oracle `o` as it stands.
    `fix(sparse)` may overwrite the ids of the first samples (hot rows, repeated rows) and returns how many leading samples it
    pinned: a draw whose pinned samples fail the margin is repeated as a whole."""
    D, S, L = [], [], []
    have, drawn = 0, 0
    while have < B:
        n = max(64, (B - have) * 5 // 4 + 8)
        de = dense(rng, n, dense_dim) if dense is gen_dense else dense(rng, n)
        sp = np.stack([rng.integers(0, r, n) for r in ln_emb], 1).astype(np.int32)
        pinned = (fix(sp) or 0) if (fix is not None and have == 0) else 0
        la = (rng.uniform(size=n) < label_p).astype(np.float32)
        ok = o.tie_margin(de, sp, delta=delta) >= delta
        drawn += n
        assert drawn < 50 * B + 10000, "more than 98 % of the candidates sit on a relu tie: delta too large for this network"
        if pinned and not ok[:pinned].all():
            continue
        D.append(de[ok]); S.append(sp[ok]); L.append(la[ok]); have += int(ok.sum())
    de, sp, la = np.concatenate(D)[:B], np.concatenate(S)[:B], np.concatenate(L)[:B]
    if stats is not None:
        stats["drawn"] = stats.get("drawn", 0) + drawn
        stats["kept"] = stats.get("kept", 0) + B
    return de, sp, la


def round_to_fp32(o):
    """the oracle's parameters become fp32-representable (the device holds fp32: an oracle that starts 3e-9 away from the
    device's rounded copy shows that residue as an error of every small update)"""
    for f in range(len(o.emb)):
        o.emb[f] = o.emb[f].astype(np.float32).astype(o.dt)
    for layers in (o.bot, o.top):
        for l in range(len(layers)):
            layers[l][0] = layers[l][0].astype(np.float32).astype(o.dt); layers[l][1] = layers[l][1].astype(np.float32).astype(o.dt)
    return o


def load_model(m, o, dtype=np.float32):
    """the oracle's parameters into the device model"""
    m.param("emb").write(np.concatenate(o.emb).astype(dtype))
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            m.param(nm + "_w", l).write(W.astype(dtype)); m.param(nm + "_b", l).write(b.astype(dtype).reshape(1, -1))


def params_of(o):
    """name -> array view of every oracle parameter"""
    out = {"emb": np.concatenate(o.emb)}
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            out[f"{nm}_w{l}"] = W; out[f"{nm}_b{l}"] = b.reshape(1, -1)
    return out


def snapshot(m, o, opt=None):
    """name -> host copy of every device parameter (and the optimizer slots of the embedding table)"""
    out = {"emb": m.param("emb").read()}
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l in range(len(layers)):
            out[f"{nm}_w{l}"] = m.param(nm + "_w", l).read(); out[f"{nm}_b{l}"] = m.param(nm + "_b", l).read()
    if opt is not None and opt.kind != "sgd":
        out["emb_slot0"] = opt.slot(m.param("emb"), 0)
        if opt.kind == "adam":
            out["emb_slot1"] = opt.slot(m.param("emb"), 1)
    return out


def assert_same_bits(a, b):
    """two runs of the same step sequence from the same start: every parameter and slot bit for bit"""
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), f"{k}: two runs of the same steps differ ({int((a[k] != b[k]).sum())} elements): the step is not deterministic"


def update_err(w0, got, want, steps=1):
    """(error, bound ratio) of a parameter UPDATE: max |d_got - d_want| over the largest update of the tensor, and the same
    after subtracting what fp32 storage allows -- (steps + 1) ulp32 of the tensor's largest magnitude: a correct fp32
    implementation rounds w - lr g once per step, and an embedding row moves by lr g ~ 1e-7 on a value of 0.05, whose ulp is
    3.7e-9 (conftest.delta_check has the element-wise form of this bound for the pairwise tables)"""
    w0 = np.asarray(w0, np.float64)
    d_got, d_want = np.asarray(got, np.float64) - w0, np.asarray(want, np.float64) - w0
    err = float(np.abs(d_got - d_want).max())
    dmax = max(float(np.abs(d_want).max()), 1e-30)
    slack = (steps + 1) * float(np.spacing(np.float32(max(np.abs(w0).max(), np.abs(np.asarray(want)).max()))))
    return err / dmax, max(err - slack, 0.0) / dmax


def assert_updates(start, got, want, tol, what="", skip=(), steps=1, tol_of=None):
    """every parameter's UPDATE since `start` within tol of the tensor's largest update (beyond the fp32 storage slack);
    tol_of: name -> bound for the tensors that take another one"""
    tol0 = tol
    for k in want:
        if k in skip or k not in start:
            continue
        tol = tol_of.get(k, tol0) if tol_of else tol0
        e, ex = update_err(start[k], got[k], want[k], steps)
        if np.abs(np.asarray(want[k], np.float64) - start[k]).max() == 0:      # a parameter that must not move (reference_compat embeddings)
            assert np.array_equal(got[k], start[k]), f"{what} {k}: moved although its gradient is zero"
            continue
        assert ex < tol, f"{what} {k}: update off by {e:.3g} of its largest element ({ex:.3g} beyond the fp32 storage slack; bound {tol:g})"


def projection(w0, got, want):
    """<d_got, d_want> / <d_want, d_want>: 1 for a faithful update; a wrong scale anywhere in the chain (an epilogue factor, a
    loss scale not divided out, a gradient applied twice) moves it at first order, while isolated per-sample deviations -- a relu
    unit on the other side of zero -- average out over the tensor"""
    w0 = np.asarray(w0, np.float64)
    dg, dw = np.asarray(got, np.float64) - w0, np.asarray(want, np.float64) - w0
    den = float((dw * dw).sum())
    return float((dg * dw).sum() / den) if den > 0 else 1.0


def assert_fp16_updates(start, got, want, B, n_emb, tol, tol_emb, what="", steps=1, flips=4, stats=None):
    """fp16-MLP mode against the fp16-operand oracle.  What two correct fp16 implementations may differ by: (a) fp32
    summation noise and fp16 roundings of sums over the batch -- well below `tol` on every dense update; (b) a relu unit that
    comes out on the other side of zero because an input activation was rounded to the neighbouring fp16 value (the two sides
    hold it 1e-7 apart, and 2e-4 of all values sit that close to a rounding boundary): ONE sample's share of the gradients, the
    unit's column of that layer and everything below it -- 1 / B of an update; a handful per step at these sizes however the
    batch is drawn (oracle: tie_margin removes the likely ones).  So: every dense update within tol + flips / B of its largest
    element AND its projection on the oracle's update within 5 tol of 1 (a systematic error -- a wrong epilogue factor, a loss
    scale not divided out -- shows there at first order, whatever the flips do; a flipped unit high in the top MLP changes its
    sample's whole bottom-MLP gradient by per cents, i.e. a bottom tensor's projection by ~0.05 / B each: seen 1.4e-4 at B = 1155);
    an embedding row is one sample's gradient: all but `flips` samples' rows within tol_emb, projection as above."""
    for k in want:
        if k not in start:
            continue
        dw = np.asarray(want[k], np.float64) - start[k]
        if np.abs(dw).max() == 0:
            assert np.array_equal(got[k], start[k]), f"{what} {k}: moved although its gradient is zero"
            continue
        c = projection(start[k], got[k], want[k])
        e, ex = update_err(start[k], got[k], want[k], steps)
        if stats is not None:
            stats[k] = max(stats.get(k, 0.0), ex); stats[k + "_proj"] = max(stats.get(k + "_proj", 0.0), abs(c - 1))
        assert abs(c - 1) < 5 * tol, f"{what} {k}: update scaled by {c:.6f} against the oracle's"
        if k == "emb":
            slack = (steps + 1) * float(np.spacing(np.float32(np.abs(start[k]).max())))
            rows = (np.abs((np.asarray(got[k], np.float64) - start[k]) - dw).max(axis=1) - slack) / np.abs(dw).max()
            n_bad = int((rows > tol_emb).sum())
            if stats is not None:
                stats["emb_bad_rows"] = max(stats.get("emb_bad_rows", 0), n_bad)
            assert n_bad <= flips * n_emb, f"{what} emb: {n_bad} rows beyond {tol_emb:g} of the largest update (more than {flips} samples' worth)"
        else:
            assert ex < tol + flips / B, f"{what} {k}: update off by {e:.3g} of its largest element ({ex:.3g} beyond the fp32 storage slack; bound {tol + flips / B:.3g})"


def record(name, **values):
    """measured errors of a test, appended to $ORX_TEST_RECORD (a jsonl file) when set: the bounds in the tests are
    chosen from these records"""
    import json
    import os
    path = os.environ.get("ORX_TEST_RECORD")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(test=name, **{k: (float(v) if np.isscalar(v) else v) for k, v in values.items()})) + "\n")
