"""Edge cases of the train step through the C ABI (empty and ragged inputs, the corners of the index space,
collisions, maximum sizes): what the reference's TF ops do at those points is restated by the NumPy oracle
(bpr.py:21-37, pairwise_log_loss.py:15-34) or stated next to the check."""
import numpy as np
import pytest

from conftest import TOL, delta_check, rel_err

pytestmark = pytest.mark.gpu


def _case(seed, NU, NI, B, D):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    return U, V, b, rng.integers(0, NU, B).astype(np.int32), rng.integers(0, NI, B).astype(np.int32), rng.integers(0, NI, B).astype(np.int32)


def test_empty_batch_and_zero_steps():
    """tf.reduce_mean of an empty tensor is NaN (BPR), a sum over nothing is 0 (UCML); the tables keep their bits"""
    from openrec_amd import runtime as rt
    U, V, b, *_ = _case(0, 50, 60, 1, 64)
    tU = rt.Table(50, 64).write(U); tV = rt.Table(60, 64).write(V); tb = rt.Table(60, 1).write(b)
    e = np.zeros(0, np.int32)
    loss, l2 = rt.pairwise_step("bpr", rt.Optimizer.sgd(0.1), tU, tV, tb, e, e, e, K=1, B=0)
    assert np.isnan(loss[0]) and l2[0] == 0
    loss, l2 = rt.pairwise_step("ucml", rt.Optimizer.sgd(0.1), tU, tV, tb, e, e, e, K=1, B=0)
    assert loss[0] == 0 and l2[0] == 0
    assert np.array_equal(tU.read(), U) and np.array_equal(tV.read(), V) and np.array_equal(tb.read(), b)


@pytest.mark.parametrize("B", [1, 3, 17, 63, 65, 4097])
@pytest.mark.parametrize("D", [16, 50, 64, 256])
def test_ragged_batch_sizes(B, D):
    """batches that fill neither a lane group, a wavefront nor a workgroup; D = 50 takes the generic-dim kernels"""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    U, V, b, u, p, n = _case(B + D, 40, 70, B, D)
    tU = rt.Table(40, D).write(U); tV = rt.Table(70, D).write(V); tb = rt.Table(70, 1).write(b)
    for model, step in (("bpr", orc.bpr_step), ("ucml", lambda *a: orc.ucml_step(*a, margin=0.5, do_censor=False))):
        loss, l2 = rt.pairwise_step(model, rt.Optimizer.adagrad(0.05), tU, tV, tb, u, p, n, margin=0.5)
        U1, V1, b1 = tU.read(), tV.read(), tb.read()
        Ur, Vr, br = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
        lw, l2w = step(Ur, Vr, br, u, p, n, orc.Adagrad(0.05, 0.1, 1e-7))
        assert abs(loss[0] - lw) <= TOL * abs(lw) + 1e-12 and abs(l2[0] - l2w) <= TOL * abs(l2w)
        assert rel_err(U1, Ur) < TOL and rel_err(V1, Vr) < TOL and rel_err(b1, br) < TOL
        tU.write(U); tV.write(V); tb.write(b)


def test_collisions_inside_one_triplet_and_corner_ids():
    """p == n in a triplet (score 0, both lookups of one row: the gradients g*u and -g*u cancel, the l2 term counts twice),
    the same row as positive of one triplet and negative of another, first and last row of both tables"""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, D = 33, 47, 64
    U, V, b, u, p, n = _case(3, NU, NI, 64, D)
    p[:8] = n[:8]                                   # p == n
    p[8:16] = 5; n[16:24] = 5                       # row 5: positive here, negative there
    u[24:28] = 0; u[28:32] = NU - 1; p[32:36] = 0; n[36:40] = NI - 1; p[40:44] = NI - 1; n[44:48] = 0
    for optk, mk, mo in (("sgd", lambda: rt.Optimizer.sgd(0.05), lambda: orc.SGD(0.05)),
                         ("adagrad", lambda: rt.Optimizer.adagrad(0.05), lambda: orc.Adagrad(0.05, 0.1, 1e-7))):
        tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
        K = 3
        ids = [np.stack([np.roll(x, s) for s in range(K)]) for x in (u, p, n)]
        loss, _ = rt.pairwise_step("bpr", mk(), tU, tV, tb, *ids, K=K, B=64)
        Ur, Vr, br = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
        oo = mo()
        for s in range(K):
            lw, _ = orc.bpr_step(Ur, Vr, br, ids[0][s], ids[1][s], ids[2][s], oo)
            assert abs(loss[s] - lw) <= TOL * abs(lw), (optk, s)
        assert rel_err(tU.read(), Ur) < TOL and rel_err(tV.read(), Vr) < TOL and rel_err(tb.read(), br) < TOL, optk


def test_negative_and_too_large_ids_raise_and_leave_the_tables_alone():
    """TF's CPU gather raises on an out-of-range index; a triplet with a bad id is skipped, the call raises"""
    from openrec_amd import runtime as rt
    U, V, b, u, p, n = _case(4, 20, 30, 16, 64)
    for bad_list, bad in ((0, -1), (0, 20), (1, 30), (2, -7), (2, 2 ** 31 - 1)):
        tU = rt.Table(20, 64).write(U); tV = rt.Table(30, 64).write(V); tb = rt.Table(30, 1).write(b)
        ids = [u.copy(), p.copy(), n.copy()]
        ids[bad_list][7] = bad
        with pytest.raises(IndexError):
            rt.pairwise_step("bpr", rt.Optimizer.sgd(0.05), tU, tV, tb, *ids)
        # every row that only the bad triplet referenced keeps its bits
        others = np.delete(np.arange(16), 7)
        for t, ref, lst in ((tU, U, [u]), (tV, V, [p, n])):
            touched = np.unique(np.concatenate([x[others] for x in lst]))
            keep = np.setdiff1d(np.arange(ref.shape[0]), touched)
            assert np.array_equal(t.read()[keep], ref[keep]), (bad_list, bad)


def test_id_stride_and_device_ids_match_host_ids():
    """K steps read from [K, stride] id matrices with stride > B; ids already on the device give the same tables bit for bit"""
    import torch
    from openrec_amd import runtime as rt
    NU, NI, D, B, K, S = 500, 700, 64, 300, 4, 333
    U, V, b, *_ = _case(5, NU, NI, 1, D)
    rng = np.random.default_rng(6)
    ids = [rng.integers(0, hi, (K, S)).astype(np.int32) for hi in (NU, NI, NI)]
    out = []
    for dev in (False, True):
        tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
        arg = [torch.from_numpy(x).to("cuda:0") for x in ids] if dev else ids
        loss, _ = rt.pairwise_step("bpr", rt.Optimizer.sgd(0.05), tU, tV, tb, *arg, K=K, B=B, id_stride=S)
        out.append((loss.copy(), tU.read(), tV.read(), tb.read()))
    tight = [np.ascontiguousarray(x[:, :B]) for x in ids]
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    loss, _ = rt.pairwise_step("bpr", rt.Optimizer.sgd(0.05), tU, tV, tb, *tight, K=K, B=B)
    for got in out:
        assert np.allclose(got[0], loss, rtol=1e-6)
        for x, y in zip(got[1:], (tU.read(), tV.read(), tb.read())):
            assert np.abs(x - y).max() <= 2e-7 * np.abs(y).max()       # (only the fp32 order of >= 3-reference sums may differ)


@pytest.mark.parametrize("rows", [(1 << 28) - 3, (1 << 28) + 5])
def test_maximum_table_sizes(rows):
    """The index space's upper end: 2^28 - 3 user rows is the largest table whose ids still carry role / urgent bits (bits
    31:28 of the rewritten id), 2^28 + 5 rows takes the path without them (byte flags, atomics for every duplicate,
    dedup_kernel).  D = 16: 17 GB per table.  The oracle steps the COMPACT problem (a step only depends on the rows it
    references); unreferenced neighbours keep their bits."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, D, B, K = rows, 100_000, 16, 8192, 3
    rng = np.random.default_rng(9)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    uid[:, 0] = NU - 1; uid[:, 1] = NU - 1; uid[:, 2] = 0; uid[:, 3] = (1 << 27) + 1; uid[1:, 4] = (1 << 27) + 1
    users = np.unique(uid); cu = np.searchsorted(users, uid).astype(np.int32)
    tU = rt.Table(NU, D).init_uniform(seed=21); tV = rt.Table(NI, D).init_uniform(seed=22); tb = rt.Table(NI, 1).init_uniform(seed=23)
    spare = np.setdiff1d(np.array([1, NU - 2, (1 << 27), (1 << 27) + 2, NU // 3], np.int32), users)
    U0, s0 = tU.gather(users), tU.gather(spare)
    V, b = tV.read(), tb.read()
    opt = rt.Optimizer.sgd(0.05)
    loss, _ = rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid, K=K, B=B)
    U = U0.copy(); V0, b0 = V.copy(), b.copy()
    oo = orc.SGD(0.05)
    for s in range(K):
        lw, _ = orc.bpr_step(U, V, b, cu[s], pid[s], nid[s], oo)
        assert abs(loss[s] - lw) <= TOL * abs(lw), s
    for name, w0, got, want in (("U", U0, tU.gather(users), U), ("V", V0, tV.read(), V), ("b", b0, tb.read(), b)):
        assert np.abs(got - want).max() <= TOL * np.abs(want).max(), name
        delta_check(w0, got, want, steps=K, what=f"rows={rows} {name}")
    assert np.array_equal(tU.gather(spare), s0)


def test_ids_produced_on_a_busy_torch_stream_are_ordered_before_the_step():
    """orx_ctx_wait_stream: device ids produced on ANOTHER stream must be complete before the library's stream reads them.  An idle
    producer stream costs no cross-stream wait (round 5: hipStreamQuery) -- a BUSY one must still be waited for: here the ids are
    copied behind ~10 ms of matrix products on a side stream, and the step is issued right away."""
    import torch
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31)
    NU, NI, B, D = 5000, 6000, 4096, 64
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    ids = [rng.integers(0, hi, B).astype(np.int32) for hi in (NU, NI, NI)]
    ctx = rt.Context(0)
    tU, tV, tb = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b))
    opt = rt.Optimizer.sgd(0.05, ctx=ctx)
    pinned = [torch.from_numpy(x).pin_memory() for x in ids]
    bufs = [torch.full((B,), 2 ** 30, dtype=torch.int32, device=dev) for _ in ids]      # (stale contents: out-of-range ids)
    big = torch.randn((4096, 4096), device=dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        acc = big
        for _ in range(20):
            acc = (acc @ big) * 1e-3                        # keeps the side stream busy for milliseconds
        for dst, src in zip(bufs, pinned):
            dst.copy_(src, non_blocking=True)
        loss, l2 = rt.pairwise_step("bpr", opt, tU, tV, tb, bufs[0], bufs[1], bufs[2])       # (after_torch: torch's CURRENT stream = side)
    lo, l2o = orc.bpr_step(U, V, b, ids[0], ids[1], ids[2], orc.SGD(lr=0.05))
    assert abs(loss[0] - lo) <= 1e-5 * abs(lo)
    for got, want in ((tU.read(), U), (tV.read(), V), (tb.read(), b)):
        assert rel_err(got, want) < 1e-5
