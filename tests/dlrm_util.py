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
        ok = o.tie_margin(de, sp) >= delta
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


def update_err(w0, got, want):
    """max |update error| over the largest update of the tensor"""
    w0 = np.asarray(w0, np.float64)
    d_got, d_want = np.asarray(got, np.float64) - w0, np.asarray(want, np.float64) - w0
    return float(np.abs(d_got - d_want).max() / max(np.abs(d_want).max(), 1e-30))


def assert_updates(start, got, want, tol, what="", skip=()):
    """every parameter's UPDATE since `start` within tol of the tensor's largest update"""
    for k in want:
        if k in skip or k not in start:
            continue
        e = update_err(start[k], got[k], want[k])
        assert e < tol, f"{what} {k}: update off by {e:.3g} of its largest element (bound {tol:g})"


def record(name, **values):
    """measured errors of a test, appended to $ORX_TEST_RECORD (a jsonl file) when set: the bounds in the tests are
    chosen from these records"""
    import json
    import os
    path = os.environ.get("ORX_TEST_RECORD")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(test=name, **{k: (float(v) if np.isscalar(v) else v) for k, v in values.items()})) + "\n")
