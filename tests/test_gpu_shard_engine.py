"""The sharded step inside the library (openrec_amd/csrc/sharded_engine.hip): the pieces that the per-phase tests do not see.
  * orx_shard_regroup: the regrouping around the plan's exchanges, against a NumPy transpose;
  * orx_shard_request_dedup_steps: the per-destination dedup plan, field by field against its restatement in
    tests/sharded_ref_backend.py (the contract the gloo world-2 tests run on), including a bucket overflow;
  * orx_sharded_pairwise_steps on a one-rank communicator without RCCL: every flag combination against the oracle on a batch
    with heavy item duplication (shared slots summed) -- the RCCL side of it runs in tests/test_gpu_rccl_rank1.py."""
import ctypes

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_regroup_is_the_transpose_of_steps_and_peers():
    import torch
    from openrec_amd import runtime as rt, _ffi
    ctx = rt.default_context(); lib = ctx._lib
    dev = torch.device("cuda", 0)
    for K, N, w in ((5, 4, 7), (1, 8, 3), (64, 2, 300), (3, 1, 11)):
        x = torch.arange(K * N * w, dtype=torch.int32, device=dev).reshape(K, N, w)
        y = torch.empty((N, K, w), dtype=torch.int32, device=dev); z = torch.empty_like(x)
        torch.cuda.synchronize()
        _ffi.check(lib.orx_shard_regroup(ctx._h, x.data_ptr(), y.data_ptr(), K, N, w, 0))
        _ffi.check(lib.orx_shard_regroup(ctx._h, y.data_ptr(), z.data_ptr(), K, N, w, 1))
        ctx.synchronize()
        assert torch.equal(y, x.transpose(0, 1).contiguous()) and torch.equal(z, x)


@pytest.mark.parametrize("world,T,NI,cap", [(1, 700, 90, 1500), (4, 1000, 333, 600), (8, 513, 5000, 200), (2, 800, 40, 10)])
def test_dedup_request_plan_matches_its_restatement(world, T, NI, cap):
    import torch
    from openrec_amd import runtime as rt, _ffi
    from sharded_ref_backend import FastOracleBackend
    rng = np.random.default_rng(world * 1000 + T)
    K = 3
    trip = np.stack([rng.integers(0, 5000, (K, T)), rng.integers(0, NI, (K, T)), rng.integers(0, NI, (K, T))], -1).astype(np.int32)
    trip[:, rng.random(T) < 0.1] = -1                                   # empty slots of the receive buffer
    trip[0, :40, 1] = 7; trip[0, 40:60, 2] = 7                          # one item asked for 60 times (p and n references)
    ctx = rt.default_context(); lib = ctx._lib
    dev = torch.device("cuda", 0)
    d = dict(send=torch.empty((K, world * cap), dtype=torch.int32, device=dev), slot=torch.empty((K, 2 * T), dtype=torch.int32, device=dev),
             u_loc=torch.empty((K, T), dtype=torch.int32, device=dev), dup=torch.empty((K, 2 * T), dtype=torch.uint8, device=dev),
             ovf=torch.zeros(1, dtype=torch.int32, device=dev), so=torch.empty((K, 2 * T, 2), dtype=torch.int32, device=dev),
             sl=torch.empty((K, T, 2), dtype=torch.int32, device=dev), sc=torch.empty(K, dtype=torch.int32, device=dev))
    tt = torch.from_numpy(trip).to(dev)
    torch.cuda.synchronize()
    _ffi.check(lib.orx_shard_request_dedup_steps(ctx._h, tt.data_ptr(), K, T, world, cap, NI, d["send"].data_ptr(), d["slot"].data_ptr(),
                                                 d["u_loc"].data_ptr(), d["dup"].data_ptr(), d["so"].data_ptr(), d["sl"].data_ptr(), d["sc"].data_ptr(), d["ovf"].data_ptr()))
    ctx.synchronize()
    r = dict(send=torch.empty((K, world * cap), dtype=torch.int32), slot=torch.empty((K, 2 * T), dtype=torch.int32),
             u_loc=torch.empty((K, T), dtype=torch.int32), dup=torch.empty((K, 2 * T), dtype=torch.uint8), ovf=torch.zeros(1, dtype=torch.int32))
    FastOracleBackend("sgd", 0.1).shard_request_dedup_steps(torch.from_numpy(trip), world, cap, NI, r["send"], r["slot"], r["u_loc"], r["dup"], r["ovf"])
    for k in ("send", "slot", "u_loc", "dup", "ovf"):
        assert torch.equal(d[k].cpu(), r[k]), k
    assert int(r["ovf"]) == (1 if cap == 10 else 0)
    # the list of shared slots: every slot with more than one reference, once
    for k in range(K):
        sl_k, dup_k = d["slot"].cpu().numpy()[k], d["dup"].cpu().numpy()[k] != 0
        want = set(int(x) for x in sl_k[dup_k & (sl_k >= 0)])
        got = [int(x) for x in d["sl"].cpu().numpy()[k, :int(d["sc"][k]), 0]]
        assert len(got) == len(set(got)) and set(got) == want
    s, live = d["slot"].cpu().numpy()[0], trip[0, :, 0] >= 0
    if cap != 10:
        shared = set(s[:40][live[:40]]) | set(s[T + 40:T + 60][live[40:60]])
        assert len(shared) == 1 and -1 not in shared and d["dup"].cpu().numpy()[0][:40][live[:40]].all()     # the 60 references share one slot


@pytest.mark.parametrize("optk", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("dedup", [True, False])
def test_library_engine_with_and_without_dedup(optk, dedup):
    import torch
    from openrec_amd import sharded
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(3)
    NU, NI, D, B, K = 500, 60, 32, 2048, 5                              # 4096 item references over 60 items: every slot is shared
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32); nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    dev = torch.device("cuda", 0)
    lr = 0.002 if optk == "adam" else 0.05
    eng = sharded.ShardedPairwise("bpr", optk, NU, NI, D, lr=lr, rank=0, world=1, device=dev, slack=1.0, dedup=dedup)
    eng.U.write(U); eng.V.write(V); eng.b.write(b)
    tu, tp, tn = (torch.from_numpy(x).to(dev) for x in (uid, pid, nid))
    torch.cuda.synchronize()
    eng.steps(tu, tp, tn, plan_chunk=2)
    assert eng._comm is not None                                        # the K-step call ran inside the library
    eng.check()
    oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[optk]()
    tl = sum(float(orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)[0]) for s in range(K))
    loss, _ = eng.loss_sums()
    assert abs(loss - tl) <= 1e-5 * abs(tl)
    tol = 2e-4 if optk == "adam" else 2e-5                               # (sums of ~70 gradient rows per item row in fp32)
    for got, want, nm in ((eng.U.read(), U, "U"), (eng.V.read(), V, "V"), (eng.b.read(), b, "b")):
        assert rel_err(got, want) < tol, (nm, rel_err(got, want))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("model,optk", [("bpr", "sgd"), ("bpr", "adagrad"), ("ucml", "sgd"), ("bpr", "adam")])
@pytest.mark.parametrize("overlap,dedup", [(False, False), (True, True), (False, True), (True, False)])
def test_library_engine_with_virtual_ranks(world, model, optk, overlap, dedup):
    """orx_sharded_pairwise_steps with world > 1 on one GPU: `world` ranks in threads of this process, each with its own context,
    exchanging through an orx_vgroup (host barrier + copy kernels in place of the ncclSend / ncclRecv groups).  The whole schedule
    of the engine -- plan regrouped by peer, gathers, gradients, applies, the halves of the overlapped path -- against the
    single-process oracle on the global batch."""
    import threading
    import torch
    from openrec_amd import sharded, _ffi
    from oracle import numpy_oracle as orc
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    NU, NI, D, Bg, K = 1001, 1503, 64, 4096, 5
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, Bg)).astype(np.int32); pid = rng.integers(0, NI, (K, Bg)).astype(np.int32); nid = rng.integers(0, NI, (K, Bg)).astype(np.int32)
    uid[:, :17] = 5; pid[:, :40] = 9                                    # hot rows: duplicates within and across the ranks' slices
    lib = _ffi.load()
    vg = ctypes.c_void_p()
    _ffi.check(lib.orx_vgroup_create(world, ctypes.byref(vg)))
    lr = 0.002 if optk == "adam" else 0.05
    engs, errs = [None] * world, []

    def run(rank):
        try:
            e = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=lr, rank=rank, world=world, device=dev, slack=1.5, vgroup=vg, dedup=dedup)
            e.U.write(U[rank::world]); e.V.write(V[rank::world]); e.b.write(b[rank::world])
            engs[rank] = e
            per = Bg // world
            sl = slice(rank * per, (rank + 1) * per)
            tu, tp, tn = (torch.from_numpy(np.ascontiguousarray(x[:, sl])).to(dev) for x in (uid, pid, nid))
            torch.cuda.synchronize()
            e.steps(tu, tp, tn, plan_chunk=2, overlap=overlap)
            assert e._comm is not None
            e.be.stream.synchronize()
        except Exception as ex:                                         # pragma: no cover
            errs.append(ex)
            lib.orx_vgroup_abort(vg)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[optk]()
    step = orc.bpr_step if model == "bpr" else (lambda *a: orc.ucml_step(*a, do_censor=False))
    tl = sum(float(step(U, V, b, uid[s], pid[s], nid[s], oo)[0]) for s in range(K))
    got = 0.0
    tol = 5e-5 if optk == "adam" else 1e-5
    for r, e in enumerate(engs):
        assert int(e._ovf[0]) == 0
        for have, want, nm in ((e.U.read(), U, "U"), (e.V.read(), V, "V"), (e.b.read(), b, "b")):
            assert rel_err(have[:len(want[r::world])], want[r::world]) < tol, (r, nm)
        got += float(e.accum[0])
    assert abs(got - tl) <= 1e-5 * abs(tl)
    engs.clear()
    import gc; gc.collect()
    lib.orx_vgroup_destroy(vg)
