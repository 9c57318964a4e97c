"""orx_apply_rows on the deterministic path (kernels_rowsort.hip): stable radix sort of (row, position) + segmented sums in
position order + the optimizer rule once per distinct row.  Against the NumPy oracle's IndexedSlices rules (SGD per occurrence,
Adagrad / Adam on the summed duplicates -- SURVEY.md A.3-A.5), over duplicate structures from "every row once" to "one row
takes the whole list", and bit-identical from run to run."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ids(kind, rng, rows, n):
    if kind == "uniform":
        ids = rng.integers(0, rows, n)
    elif kind == "onehot":                                    # one row takes everything: runs across hundreds of 64-entry blocks
        ids = np.full(n, rows // 3)
    elif kind == "five":
        ids = rng.choice(rng.integers(0, rows, 5), n)
    elif kind == "zipf":
        ids = np.minimum(rng.zipf(1.05, n) - 1, rows - 1)
    else:                                                     # runs that end exactly on block boundaries
        ids = np.repeat(rng.permutation(rows)[:n // 64 + 1], 64)[:n]
    ids = ids.astype(np.int32)
    ids[rng.random(n) < 0.04] = -1                            # padding slots (the dense slot of a DLRM id matrix)
    return ids


def _apply(rt, lib, ffi, torch, opt, t, ids, grads):
    ffi.check(lib.orx_apply_rows(t.ctx._h, opt._h, t._h, None, ids.data_ptr(), ids.numel(), grads.data_ptr(), grads.shape[1]))


@pytest.mark.parametrize("kind", ["uniform", "onehot", "five", "zipf", "blocks"])
@pytest.mark.parametrize("optname", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("rows,D,n", [(7, 4, 300), (5000, 24, 9000), (200000, 128, 40000), (1_200_000, 64, 70001)])
def test_apply_rows_sorted(kind, optname, rows, D, n):
    import torch
    from openrec_amd import runtime as rt, _ffi
    from oracle import numpy_oracle as orc
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{kind} {optname} {rows}".encode()))
    ids = _ids(kind, rng, rows, n)
    W0 = rng.uniform(-0.05, 0.05, (rows, D)).astype(np.float32)
    steps = 3
    G = (rng.normal(size=(steps, n, D)) * 0.01).astype(np.float32)
    dev = torch.device("cuda", 0)
    ctx = rt.default_context()
    lib = ctx._lib
    d_ids = torch.from_numpy(ids).to(dev)

    def run():
        t = rt.Table(rows, D).write(W0)
        opt = {"sgd": lambda: rt.Optimizer.sgd(0.05), "adagrad": lambda: rt.Optimizer.adagrad(0.05, 0.1, 1e-7),
               "adam": lambda: rt.Optimizer.adam(0.002, 0.9, 0.999, 1e-7)}[optname]()
        for s in range(steps):
            if optname == "adam":
                opt.advance([t])
            g = torch.from_numpy(G[s]).to(dev)
            torch.cuda.synchronize()
            _apply(rt, lib, _ffi, torch, opt, t, d_ids, g)
            ctx.synchronize()
        out = [t.read()]
        if optname != "sgd":
            out.append(opt.slot(t, 0))
        if optname == "adam":
            out.append(opt.slot(t, 1))
        return out

    a, b = run(), run()
    for x, y in zip(a, b):
        assert np.array_equal(x, y), "two runs differ: the path is not deterministic"
    o = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(0.002, 0.9, 0.999, 1e-7)}[optname]()
    W = W0.astype(np.float64)
    live = ids >= 0
    for s in range(steps):
        if hasattr(o, "begin_step"):
            o.begin_step()
        o.apply(W, ids[live], G[s][live].astype(np.float64), key="t")
    maxrefs = int(np.bincount(ids[live]).max())
    if optname == "adam":
        # the slots are linear in the summed gradient: held to 1e-5 (4e-5 for sums of tens of thousands of fp32 terms).  The
        # weights take Adam's normalised update, whose slope at g ~ 0 is lr_1 (1 - b1) / eps: the fp32 rounding of an n-term
        # sum enters amplified (conftest.TOL_ADAM is that bound for n = 1..2; it grows with sqrt(n))
        assert rel_err(a[1], o.m["t"]) < 1e-5 * (4 if maxrefs > 1000 else 1), (kind, optname)
        assert rel_err(a[2], o.v["t"]) < 5e-5, (kind, optname)          # (1 - fp32(0.999) = 1e-3 (1 + 1.29e-5): the cast TF's apply op makes too)
        assert rel_err(a[0], W) < min(2e-3, 5e-5 * max(1.0, np.sqrt(maxrefs) / 2)), (kind, optname)
    else:
        # hot rows: a sum of tens of thousands of fp32 terms carries sqrt(n) ulps
        assert rel_err(a[0], W) < 1e-5 * (4 if maxrefs > 1000 else 1), (kind, optname)
    untouched = np.ones(rows, bool); untouched[ids[live]] = False
    if optname != "adam":                                     # (TF's Adam moves every row)
        assert np.array_equal(a[0][untouched], W0[untouched])


def test_apply_rows_out_of_range_id_is_flagged():
    import torch
    from openrec_amd import runtime as rt, _ffi
    dev = torch.device("cuda", 0)
    t = rt.Table(100, 16).fill(0.0)
    opt = rt.Optimizer.sgd(0.1)
    ids = torch.tensor([1, 2, 100, 3], dtype=torch.int32, device=dev)
    g = torch.ones(4, 16, device=dev)
    torch.cuda.synchronize()
    lib = t.ctx._lib
    _ffi.check(lib.orx_apply_rows(t.ctx._h, opt._h, t._h, None, ids.data_ptr(), 4, g.data_ptr(), 16))
    with pytest.raises(IndexError):
        _ffi.check(lib.orx_check_index_error(t.ctx._h))
    w = t.read()
    assert np.allclose(w[[1, 2, 3]], -0.1) and np.count_nonzero(w) == 48      # the valid rows were applied, nothing else touched
