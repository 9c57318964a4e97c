"""The C restatement (oracle/orx_oracle.c, the cpu_baseline port) must agree
with the NumPy oracle.  CPU only."""
import numpy as np
import pytest

from conftest import TOL, TOL_ADAM, rel_err
from oracle import numpy_oracle as orc
from oracle import c_oracle


def _inputs(seed, NU, NI, B, D):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    ids = [rng.integers(0, n, B).astype(np.int32) for n in (NU, NI, NI)]
    ids[0][:9] = 3
    ids[2][9:14] = ids[1][9:14]
    return U, V, b, ids


@pytest.mark.parametrize("model", ["bpr", "ucml"])
@pytest.mark.parametrize("opt", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("D", [50, 64, 128])
def test_c_oracle_matches_numpy(model, opt, D):
    U, V, b, ids = _inputs(5, 60, 80, 300, D)
    U2, V2, b2 = U.copy(), V.copy(), b.copy()
    lr = 0.005 if opt == "adam" else 0.05             # (Adam moves every weight by ~lr per step)
    cpu = c_oracle.PairwiseCPU(model, opt, U2, V2, b2, lr=lr)
    o = {"sgd": lambda: orc.SGD(lr=0.05), "adagrad": lambda: orc.Adagrad(lr=0.05, initial_accumulator_value=0.1, epsilon=1e-7),
         "adam": lambda: orc.AdamTFSparse(lr)}[opt]()
    for s in range(3 if opt == "adam" else 2):
        u, p, n = np.roll(ids[0], s), np.roll(ids[1], 2 * s), np.roll(ids[2], 3 * s)
        if model == "bpr":
            l_ref = orc.bpr_step(U, V, b, u, p, n, o)
        else:
            l_ref = orc.ucml_step(U, V, b, u, p, n, o, margin=0.5, do_censor=False)
        l_c = cpu.step(u, p, n)
        assert rel_err(l_c, l_ref) < (TOL_ADAM if opt == "adam" else TOL)
    tol = TOL_ADAM if opt == "adam" else TOL          # (conftest.TOL_ADAM: the derivation)
    assert rel_err(U2, U) < tol and rel_err(V2, V) < tol and rel_err(b2, b) < tol
    if opt == "adam":
        assert rel_err(cpu.m[1], o.m["V"]) < tol and rel_err(cpu.v[0], o.v["U"]) < tol and rel_err(cpu.m[2], o.m["b"][:, 0]) < tol
    if opt == "adagrad":
        assert rel_err(cpu.accU, o.acc["U"]) < 1e-5 and rel_err(cpu.accb, o.acc["b"][:, 0]) < 1e-5


def test_c_censor():
    rng = np.random.default_rng(0)
    W = rng.uniform(-1, 1, (30, 16)).astype(np.float32)
    W[4] *= 0.01
    ids = rng.integers(0, 30, 40).astype(np.int32)
    W2 = W.copy()
    orc.censor(W, ids)
    c_oracle.censor(W2, ids)
    assert rel_err(W2, W) < 1e-6
