"""The full-size parity check must be able to FAIL (round-1 verdict, weak #2): at BASELINE's batch size the loss gradient
is below a 1e-5 bound on the tables.  These CPU tests hold `conftest.delta_check` itself to that: it accepts the C
oracle against the NumPy oracle's update and rejects an update without the loss gradient (what a kernel with g = 0 in
pairwise_log_loss.py:32's derivative would produce) and one whose loss gradient is 5 % off, at configs[1] shapes."""
import numpy as np
import pytest

from conftest import delta_check


def _case(NU, NI, D, B, seed=0):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, B).astype(np.int32); pid = rng.integers(0, NI, B).astype(np.int32); nid = rng.integers(0, NI, B).astype(np.int32)
    return U, V, b, uid, pid, nid


def _sgd_step(U, V, b, uid, pid, nid, lr, gscale=1.0, l2w=1.0):
    """fp32 restatement of one BPR SGD step (bpr.py:21-37, pairwise_log_loss.py:15-34) with the loss coefficient scaled
    by `gscale` (0: the loss gradient dropped) -- per-occurrence scatter like TF's ResourceScatterAdd"""
    B = len(uid)
    u, p, n = U[uid], V[pid], V[nid]
    x = (u * p).sum(1) + b[pid, 0] - (u * n).sum(1) - b[nid, 0]
    g = (np.where(x >= -30, -1.0 / (1.0 + np.exp(x.astype(np.float64))), 0.0) / B * gscale).astype(np.float32)[:, None]
    lr = np.float32(lr); l2w = np.float32(l2w)
    Un, Vn, bn = U.copy(), V.copy(), b.copy()
    np.subtract.at(Un, uid, lr * (g * (p - n) + l2w * u))
    np.subtract.at(Vn, pid, lr * (g * u + l2w * p))
    np.subtract.at(Vn, nid, lr * (-g * u + l2w * n))
    np.subtract.at(bn, pid, lr * g)
    np.subtract.at(bn, nid, lr * -g)
    return Un, Vn, bn


@pytest.mark.parametrize("l2w", [1.0, 0.0])
def test_delta_check_accepts_a_faithful_update_and_rejects_a_missing_loss_gradient(l2w):
    from oracle import c_oracle
    NU = NI = 1_000_000
    D, B = 64, 65536
    U, V, b, uid, pid, nid = _case(NU, NI, D, B)
    U0, V0, b0 = U.copy(), V.copy(), b.copy()
    c_oracle.PairwiseCPU("bpr", "sgd", U, V, b, lr=0.05, l2w=l2w).step(uid, pid, nid)       # U, V, b now hold the oracle's step
    ok = _sgd_step(U0, V0, b0, uid, pid, nid, 0.05, 1.0, l2w)
    for w0, got, want in zip((U0, V0, b0), ok, (U, V, b)):
        assert abs(delta_check(w0, got, want, steps=1) - 1.0) <= 1e-3
    # the old bound on the tables is blind to the loss gradient when the l2 term is on ...
    no_g = _sgd_step(U0, V0, b0, uid, pid, nid, 0.05, 0.0, l2w)
    if l2w == 1.0:      # (rows; an item referenced four times with one sign moves its bias by 1.5e-6, just visible)
        for got, want in zip(no_g[:2], (U, V)):
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    # ... delta_check is not: on the bias table (the whole update is the loss gradient) and on the rows
    for w0, got, want in zip((U0, V0, b0), no_g, (U, V, b)):
        with pytest.raises(AssertionError):
            delta_check(w0, got, want, steps=1)
    off5 = _sgd_step(U0, V0, b0, uid, pid, nid, 0.05, 1.05, l2w)
    with pytest.raises(AssertionError):
        delta_check(b0, off5[2], b, steps=1)
    if l2w == 0.0:      # without l2 the scale of the update is measured directly
        coef = None
        try:
            coef = delta_check(U0, off5[0], U, steps=1)
        except AssertionError:
            coef = 1.05
        assert abs(coef - 1.0) > 1e-2
