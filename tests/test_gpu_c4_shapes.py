"""BASELINE.json configs[3] shapes on ONE GPU: BPR D=64 over 10 M users x 50 M items (15.4 GB of tables; 8-way sharded in
the config, but they fit one MI355X), B = 65536.  What these sizes exercise that configs[1] does not: row indices up to
5e7 (row offsets beyond 2^32 bytes: 12.8 GB into the item table), 611 + 3052 row ranges in the duplicate plan, duplicate
rates of 0.3 % instead of 6-13 %.  The oracle runs on the COMPACT problem: a step only depends on the rows it references,
so the referenced rows are gathered before the steps, the ids renumbered densely, the NumPy oracle stepped on those small
tables (bpr.py:21-37 through numpy_oracle.bpr_step) and the same rows gathered again afterwards; a sample of unreferenced
rows must keep its exact bits."""
import numpy as np
import pytest

from conftest import TOL, delta_check

pytestmark = pytest.mark.gpu
NU, NI, D, B, K = 10_000_000, 50_000_000, 64, 65536, 3


def _ids(seed):
    rng = np.random.default_rng(seed)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    # the corners of the index space, duplicated within and across steps
    uid[:, 0] = NU - 1; uid[:, 1] = NU - 1; uid[:, 2] = 0
    pid[:, 0] = NI - 1; nid[:, 1] = NI - 1; pid[:, 2] = 0; nid[:, 3] = (1 << 25) + 7; pid[:, 4] = (1 << 25) + 7
    return uid, pid, nid


def _compact(uid, pid, nid):
    users = np.unique(uid); items = np.unique(np.concatenate([pid.reshape(-1), nid.reshape(-1)]))
    return users, items, np.searchsorted(users, uid).astype(np.int32), np.searchsorted(items, pid).astype(np.int32), \
        np.searchsorted(items, nid).astype(np.int32)


def _check(read_rows, users, items, cu, cp, cn, run_steps, optname):
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(99)
    spare_u = np.setdiff1d(rng.integers(0, NU, 4096), users).astype(np.int32)
    spare_i = np.setdiff1d(np.concatenate([rng.integers(0, NI, 4096), [NI - 2, 1, (1 << 25) + 8]]), items).astype(np.int32)
    U0, V0, b0 = read_rows("U", users), read_rows("V", items), read_rows("b", items)
    su0, si0, sb0 = read_rows("U", spare_u), read_rows("V", spare_i), read_rows("b", spare_i)
    loss = run_steps()
    U, V, b = U0.copy(), V0.copy(), b0.copy()
    oo = orc.SGD(0.05) if optname == "sgd" else orc.Adagrad(0.05, 0.1, 1e-7)
    for s in range(K):
        lw, _ = orc.bpr_step(U, V, b, cu[s], cp[s], cn[s], oo)
        if loss is not None:
            assert abs(loss[s] - lw) <= TOL * abs(lw), (s, loss[s], lw)
    for name, ids, w0, want in (("U", users, U0, U), ("V", items, V0, V), ("b", items, b0, b)):
        got = read_rows(name, ids)
        assert np.abs(got - want).max() <= TOL * np.abs(want).max(), name
        delta_check(w0, got, want, steps=K, what=f"C4 {optname} {name}")
    assert np.array_equal(read_rows("U", spare_u), su0) and np.array_equal(read_rows("V", spare_i), si0)
    assert np.array_equal(read_rows("b", spare_i), sb0)


@pytest.mark.parametrize("optname", ["sgd", "adagrad"])
def test_c4_shapes_single_gpu(optname):
    from openrec_amd import runtime as rt
    uid, pid, nid = _ids(1)
    users, items, cu, cp, cn = _compact(uid, pid, nid)
    tU = rt.Table(NU, D).init_uniform(seed=11); tV = rt.Table(NI, D).init_uniform(seed=12); tb = rt.Table(NI, 1).init_uniform(seed=13)
    tabs = dict(U=tU, V=tV, b=tb)
    opt = rt.Optimizer.sgd(0.05) if optname == "sgd" else rt.Optimizer.adagrad(0.05, 0.1, 1e-7)
    _check(lambda n, ids: tabs[n].gather(ids), users, items, cu, cp, cn,
           lambda: rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid, K=K, B=B)[0], optname)


def test_c4_shapes_sharded_engine_world1():
    """the row-sharded engine with one rank holds the same tables: its planned K-step path at these sizes"""
    import torch
    from openrec_amd import sharded
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    uid, pid, nid = _ids(2)
    users, items, cu, cp, cn = _compact(uid, pid, nid)
    eng = sharded.ShardedPairwise("bpr", "sgd", NU, NI, D, lr=0.05, rank=0, world=1, device=dev, seed=3)
    tabs = dict(U=eng.U, V=eng.V, b=eng.b)

    def run():
        eng.steps(*(torch.from_numpy(x).to(dev) for x in (uid, pid, nid)), plan_chunk=2)
        eng.check()
        return None

    _check(lambda n, ids: tabs[n].gather(ids), users, items, cu, cp, cn, run, "sgd")
