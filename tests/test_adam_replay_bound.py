"""The bounded replay of the lazily-applied TF-2.0 Adam (openrec_amd/csrc/orx_device.h: adam_replay1) restated in
NumPy fp32, CPU only.  A row that waits n steps between references owes n gradient-free steps
    m <- b1*m;  v <- b2*v;  w <- w - lr_k * m / (sqrt(v) + eps)
The device loop stops at the first step whose update term u satisfies (w - 4u) == w in fp32 and finishes m, v in closed
form.  Claim checked here: stopping there changes NOTHING -- w is bit-identical to the full replay (in the same
arithmetic), m and v agree to rounding -- because u shrinks by >= 9 % per step from then on."""
import numpy as np
import pytest

f32 = np.float32


def lr_table(lr, b1, b2, n):
    k = np.arange(n + 1, dtype=np.float64)
    t = np.zeros(n + 1)
    t[1:] = lr * np.sqrt(1.0 - b2 ** k[1:]) / (1.0 - b1 ** k[1:])
    return t.astype(f32)


def replay(w, m, v, frm, to, lrt, b1, b2, eps, bounded):
    """element-wise model of adam_replay1<false> (the per-element form used by the table flush)"""
    b1, b2, eps = f32(b1), f32(b2), f32(eps)
    sb2 = np.sqrt(b2, dtype=f32)
    ce = f32(eps * (f32(1) - sb2))
    w, m, v = w.astype(f32).copy(), m.astype(f32).copy(), v.astype(f32).copy()
    d = np.sqrt(v, dtype=f32) + eps
    q = (f32(1) / d).astype(f32)
    live = np.ones(w.shape, bool)
    steps = np.zeros(w.shape, np.int64)                  # steps taken in the loop, per element
    for k in range(frm + 1, to + 1):
        if not live.any():
            break
        mm = (m * b1).astype(f32); vv = (v * b2).astype(f32); dd = (d * sb2 + ce).astype(f32)
        qq = (q * (f32(2) - dd * q)).astype(f32)         # Newton step on the carried reciprocal
        u = ((lrt[k] * mm) * qq).astype(f32)
        ww = (w - u).astype(f32)
        m = np.where(live, mm, m); v = np.where(live, vv, v); d = np.where(live, dd, d); q = np.where(live, qq, q)
        w = np.where(live, ww, w)
        steps += live
        if bounded:
            live &= ~((ww - f32(4) * u).astype(f32) == ww)
    rem = (to - frm) - steps
    m = np.where(rem > 0, m * np.exp2(rem * np.log2(b1, dtype=f32), dtype=f32), m).astype(f32)
    v = np.where(rem > 0, v * np.exp2(rem * np.log2(b2, dtype=f32), dtype=f32), v).astype(f32)
    return w, m, v, steps


@pytest.mark.parametrize("b2", [0.999, 0.99])
@pytest.mark.parametrize("frm,to", [(1, 2000), (37, 900), (500, 520), (3, 4)])
def test_bounded_replay_is_the_full_replay(frm, to, b2):
    rng = np.random.default_rng(frm + to)
    n = 4096
    b1, eps, lr = 0.9, 1e-7, 0.01
    lrt = lr_table(lr, b1, b2, to + 1)
    w = rng.uniform(-0.05, 0.05, n)
    g = rng.normal(0, 1, n) * 10.0 ** rng.uniform(-9, -1, n)           # gradients over eight decades (sqrt(v) << eps ... >> eps)
    m = (1 - b1) * g * rng.uniform(0.1, 3, n)
    v = (1 - b2) * g * g * rng.uniform(0.1, 3, n)
    m[:64] = 0; v[:64] = 0                                              # never-referenced elements
    wf, mf, vf, sf = replay(w, m, v, frm, to, lrt, b1, b2, eps, bounded=False)
    wb, mb, vb, sb = replay(w, m, v, frm, to, lrt, b1, b2, eps, bounded=True)
    assert np.array_equal(wf, wb)                                       # bit-identical weights
    assert (sf == to - frm).all() and sb.max() <= min(to - frm, 260)    # the loop is bounded (~170 steps at the defaults)
    ok = np.abs(mf) > 1e-35                                             # (below that fp32 goes denormal in the sequential form)
    if ok.any():
        assert np.abs(mb[ok] / mf[ok] - 1).max() < 2e-4
    assert np.abs(mb[~ok]).max(initial=0) < 1e-30
    assert np.abs(vb[v > 0] / vf[v > 0] - 1).max() < 2e-4
    # and both are the fp64 dense rule to rounding
    W, M, V = w.copy(), m.copy(), v.copy()
    for k in range(frm + 1, to + 1):
        M *= b1; V *= b2
        W -= float(lrt[k]) * M / (np.sqrt(V) + eps)
    assert np.abs(wb - W).max() <= 1e-4 * np.abs(W).max()       # (up to ~200 fp32 additions per element)
