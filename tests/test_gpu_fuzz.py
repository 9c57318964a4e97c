"""Randomized differential test of the exact pairwise step (recommenders/bpr.py:21-37, ucml.py:21-48 through
tf2_examples/bpr_citeulike.py:33-39) against the fp64 NumPy oracle: a seeded 34-case subset of scratch/fuzz_pairwise.py -- models,
optimizers, float4 and generic dims, table sizes 30 .. 500 k, batch 1 .. 65 536, uniform / Zipf / one-hot / five-row id
distributions, censor -- plus the four forced cases that reach the corners of the duplicate machinery: more than 6656
thrice-referenced rows per row range (global rank counters), 65 536 references on ONE row (three reduction-tree levels), more
steps than one plan chunk, and a 500 k-item Zipf table with censor."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORCED = [dict(NI=30000, NU=30000, B=65536, K=2, D=64, skew="uniform"),
          dict(NI=5000, NU=5000, B=65536, K=2, D=64, skew="one_hot_item", model="bpr", opt="sgd"),
          dict(NI=2000, NU=2000, B=1000, K=300, D=16, skew="zipf", opt="sgd"),
          dict(NI=500000, NU=300000, B=20000, K=3, D=128, skew="zipf", model="ucml", censor=True)]


def make_case(rng, forced=None):
    c = dict(model=str(rng.choice(["bpr", "ucml"])), opt=str(rng.choice(["sgd", "adagrad", "adam"])), D=int(rng.choice([16, 32, 64, 128, 256, 50, 20])),
             NU=int(rng.choice([50, 700, 5000, 40000, 300000])), NI=int(rng.choice([30, 900, 6000, 30000, 500000])),
             B=int(rng.choice([1, 7, 256, 1000, 4096, 8191, 20000])), K=int(rng.choice([1, 2, 3, 5, 9])),
             skew=str(rng.choice(["uniform", "zipf", "one_hot_item", "one_hot_user", "few"])), censor=bool(rng.random() < 0.2))
    if forced:
        c.update(forced)
    if c["model"] == "bpr":
        c["censor"] = False
    return c


def run_case(c, rng):
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, B, K, D = c["NU"], c["NI"], c["B"], c["K"], c["D"]
    scale = 25 if c["censor"] else 1
    U = (rng.uniform(-.05, .05, (NU, D)) * scale).astype(np.float32); V = (rng.uniform(-.05, .05, (NI, D)) * scale).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32)
    if c["skew"] == "zipf":
        w = 1.0 / np.arange(1, NI + 1) ** 1.1; cdf = np.cumsum(w / w.sum()); perm = rng.permutation(NI)
        draw = lambda: perm[np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1)].astype(np.int32)
        pid, nid = draw(), draw()
    elif c["skew"] == "one_hot_item":
        pid = np.full((K, B), int(rng.integers(0, NI)), np.int32); nid = rng.integers(0, NI, (K, B)).astype(np.int32)
        nid[:, ::2] = pid[:, ::2]
    elif c["skew"] == "few":
        pid = rng.integers(0, min(NI, 5), (K, B)).astype(np.int32); nid = rng.integers(0, min(NI, 5), (K, B)).astype(np.int32)
        uid = rng.integers(0, min(NU, 3), (K, B)).astype(np.int32)
    else:
        pid = rng.integers(0, NI, (K, B)).astype(np.int32); nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    if c["skew"] == "one_hot_user":
        uid[:, : max(1, B // 2)] = int(rng.integers(0, NU))
    lr = 0.001 if c["skew"] != "uniform" else 0.02
    if c["opt"] == "adam":
        lr = 0.0005                                  # Adam moves every weight by ~lr per step whatever the gradient
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    opt = {"sgd": lambda: rt.Optimizer.sgd(lr), "adagrad": lambda: rt.Optimizer.adagrad(lr), "adam": lambda: rt.Optimizer.adam(lr)}[c["opt"]]()
    oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr), "adam": lambda: orc.AdamTFSparse(lr)}[c["opt"]]()
    loss, l2 = rt.pairwise_step(c["model"], opt, tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=c["censor"])
    U, V, b = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
    worst = 0.0
    for s in range(K):
        if c["model"] == "bpr":
            lw, l2w = orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)
        else:
            lw, l2w = orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=c["censor"])
        worst = max(worst, abs(loss[s] - lw) / max(abs(lw), 1e-30), abs(l2[s] - l2w) / max(abs(l2w), 1e-30))
    for got, want in ((tU.read(), U), (tV.read(), V), (tb.read(), b)):
        worst = max(worst, np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
    diverged = max(np.abs(U).max(), np.abs(V).max()) > 50 * scale       # lr x duplicates > 1: rounding is amplified without bound
    return worst, diverged


@pytest.mark.parametrize("i", range(34))
def test_fuzz_case(i):
    rng = np.random.default_rng(4200 + i)
    c = make_case(rng, FORCED[i] if i < len(FORCED) else None)
    worst, diverged = run_case(c, rng)
    if diverged:
        pytest.skip(f"the run diverges in the oracle too (lr x duplicates > 1): {c}")
    # float4 dims: no atomics anywhere, 5e-5 (conftest.TOL_ADAM: the Adam cases set the bound); generic dims (20, 50) put the
    # references of a hot row into fp32 atomics, whose arrival order is a sequential sum of thousands of terms
    tol = 5e-5 if c["D"] % 4 == 0 and c["D"] >= 16 and c["D"] not in (20,) else 3e-4
    assert worst < tol, (worst, c)
