"""GPU parity of the DLRM train step against the torch-autograd golden fixtures
and the NumPy oracle."""
import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err, OPT_KW

pytestmark = [pytest.mark.gpu, pytest.mark.fp32_tie]

CFG = dict(m_spa=4, ln_emb=[7, 5, 11], ln_bot=[8, 4], ln_top=[16, 8, 1], dense_dim=13)
KW = {
    "compat": dict(reference_compat=True),
    "compatself": dict(reference_compat=True, arch_interaction_itself=True),
    "intended": dict(reference_compat=False),
    "intendedbce": dict(reference_compat=False, loss_func="bce", loss_threshold=0.05, sigmoid_bot=True),
}


def _opt(rt, kind):
    kw = OPT_KW[kind]
    if kind == "sgd":
        return rt.Optimizer.sgd(kw["lr"])
    if kind == "adagrad":
        return rt.Optimizer.adagrad(kw["lr"], kw["initial_accumulator_value"], kw["epsilon"])
    return rt.Optimizer.adam(kw["lr"], kw["beta_1"], kw["beta_2"], kw["epsilon"])


def _load(m, g, prefix="in_"):
    emb = np.concatenate([g[f"{prefix}emb{f}"] for f in range(len(CFG["ln_emb"]))])
    m.param("emb").write(emb)
    for nm, n in (("bot", len(CFG["ln_bot"])), ("top", len(CFG["ln_top"]))):
        for l in range(n):
            m.param(nm + "_w", l).write(g[f"{prefix}{nm}{l}W"])
            m.param(nm + "_b", l).write(g[f"{prefix}{nm}{l}b"].reshape(1, -1))


@pytest.mark.parametrize("fname", golden_files("dlrm"))
def test_dlrm_golden(fname):
    from openrec_amd import runtime as rt
    _, name, optkind = fname[:-4].split("_")
    g = load_golden(fname)
    m = rt.DLRMModel(**CFG, **KW[name])
    _load(m, g)
    opt = _opt(rt, optkind)
    losses = [m.step(opt, g["dense"], g["sparse"], g["label"])[0] for _ in range(2)]
    tol = 1e-5
    assert rel_err(losses, g["losses"]) < tol
    emb = m.param("emb").read()
    ref = np.concatenate([g[f"out_emb{f}"] for f in range(3)])
    assert rel_err(emb, ref) < tol
    for nm, n in (("bot", 2), ("top", 3)):
        for l in range(n):
            assert rel_err(m.param(nm + "_w", l).read(), g[f"out_{nm}{l}W"]) < 5 * tol, (nm, l)
            assert rel_err(m.param(nm + "_b", l).read().reshape(-1), g[f"out_{nm}{l}b"]) < 5 * tol, (nm, l)
    if name == "compat":      # the reference's bug: embeddings never move
        assert np.array_equal(emb, np.concatenate([g[f"in_emb{f}"] for f in range(3)]))


@pytest.mark.parametrize("compat", [True, False])
def test_dlrm_example_shapes_vs_oracle(compat):
    """tf2_examples/dlrm_criteo.py shapes: dim 4, bottom [8,4], top [128,64,1], 26 tables, batch 1024."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(0)
    ln_emb = [int(x) for x in rng.integers(3, 2000, 26)]
    cfg = dict(m_spa=4, ln_emb=ln_emb, ln_bot=[8, 4], ln_top=[128, 64, 1], dense_dim=13)
    o = DLRMOracle(dtype=np.float32, seed=2, reference_compat=compat, **cfg)
    m = rt.DLRMModel(reference_compat=compat, **cfg)
    m.param("emb").write(np.concatenate(o.emb))
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            m.param(nm + "_w", l).write(W); m.param(nm + "_b", l).write(b.reshape(1, -1))
    B = 1024
    dense = np.log1p(rng.integers(0, 100, (B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, B) for n in ln_emb], 1).astype(np.int32)
    label = (rng.uniform(size=B) < 0.25).astype(np.float32)
    assert rel_err(m.inference(dense, sparse), o.inference(dense, sparse)) < 1e-5
    opt, oo = rt.Optimizer.adagrad(0.05, 0.1, 1e-7), orc.Adagrad(0.05, 0.1, 1e-7)
    for s in range(2):
        l = m.step(opt, dense, sparse, label)[0]
        lr = o.step(dense, sparse, label, oo)
        assert abs(l - lr) <= 1e-5 * abs(lr)
    assert rel_err(m.param("emb").read(), np.concatenate(o.emb)) < 1e-4
    assert rel_err(m.param("top_w", 0).read(), o.top[0][0]) < 1e-4
    assert rel_err(m.param("bot_w", 0).read(), o.bot[0][0]) < 1e-4


def test_dlrm_api_surface():
    from openrec_amd.tf2.recommenders import DLRM
    from openrec_amd.tf2.compat import tf, optimizers
    rng = np.random.default_rng(1)
    counts = [10, 20, 30]
    dlrm_model = DLRM(m_spa=4, ln_emb=counts, ln_bot=[8, 4], ln_top=[128, 64, 1])
    optimizer = optimizers.Adam()
    dense = rng.uniform(0, 3, (64, 13)).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, 64) for n in counts], 1).astype(np.int32)
    label = (rng.uniform(size=64) < 0.5).astype(np.float32)
    first = None
    for it in range(30):
        with tf.GradientTape() as tape:
            loss_value = dlrm_model(dense, sparse, label)
        gradients = tape.gradient(loss_value, dlrm_model.trainable_variables)
        optimizer.apply_gradients(zip(gradients, dlrm_model.trainable_variables))
        first = float(loss_value) if first is None else first
    assert float(loss_value) < first                       # it trains
    pred = dlrm_model.inference(dense, sparse)
    assert pred.shape == (64,) and (pred > 0).all() and (pred < 1).all()
    with pytest.raises(AttributeError):
        DLRM(m_spa=4, ln_emb=counts, ln_bot=[8, 4], ln_top=[8, 1], arch_interaction_op='cat')


def test_dlrm_fp16_mlp_mode_tracks_the_fp32_oracle():
    """ORX_DLRM_FP16_MLP: MLP products on fp16 MFMA (performance mode).  Not a parity mode -- the check
    is that loss and every parameter UPDATE agree with the fp32 oracle to fp16 accuracy, which also
    pins the MFMA fragment layouts of all three products (X*W, dY*W^T, X^T*dY) on asymmetric data."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(4)
    ln_emb = [50, 300, 7, 1000]
    cfg = dict(m_spa=32, ln_emb=ln_emb, ln_bot=[96, 32], ln_top=[200, 72, 1], dense_dim=13)
    o = DLRMOracle(dtype=np.float32, seed=5, reference_compat=False, **cfg)
    m = rt.DLRMModel(reference_compat=False, fp16_mlp=True, **cfg)
    m.param("emb").write(np.concatenate(o.emb))
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            b[:] = rng.normal(size=b.shape).astype(np.float32) * 0.1
            m.param(nm + "_w", l).write(W); m.param(nm + "_b", l).write(b.reshape(1, -1))
    B = 333
    dense = np.log1p(rng.integers(0, 100, (B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, B) for n in ln_emb], 1).astype(np.int32)
    label = (rng.uniform(size=B) < 0.3).astype(np.float32)
    before = {("top", 0): o.top[0][0].copy(), ("bot", 1): o.bot[1][0].copy(), "emb": np.concatenate(o.emb).copy()}
    p16, p32 = m.inference(dense, sparse), o.inference(dense, sparse)
    assert np.abs(p16 - p32).max() < 5e-3
    l16 = m.step(rt.Optimizer.sgd(0.1), dense, sparse, label)[0]
    l32 = o.step(dense, sparse, label, orc.SGD(0.1))
    assert abs(l16 - l32) < 5e-3 * abs(l32)
    for key, dev, ref in ((("top", 0), m.param("top_w", 0).read(), o.top[0][0]), (("bot", 1), m.param("bot_w", 1).read(), o.bot[1][0]),
                          ("emb", m.param("emb").read(), np.concatenate(o.emb))):
        du, dr = dev - before[key], ref - before[key]
        assert np.abs(du - dr).max() < 0.03 * np.abs(dr).max() + 1e-7, key


@pytest.mark.parametrize("compat", [False, True])
@pytest.mark.parametrize("cfg_name", ["narrow", "wide", "ragged"])
def test_dlrm_fp16_mlp_mode_over_several_steps_and_shapes(cfg_name, compat):
    """ORX_DLRM_FP16_MLP across several steps (the fp16 weight copies must follow every update: a copy refreshed one step late
    would show as a first-order error in the second step's update), layer widths that exercise every tile shape of
    kernels_gemm16.hip (256x128 / 128x128 / 128x64 blocks, partial tiles, widths not a multiple of 8 that fall back to the
    fp32-operand kernels), the fp16-only ("lean") activations, and both interaction modes (reference_compat: R16 comes from the
    cast kernel instead of the MFMA interaction)."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(11)
    ln_emb = [50, 300, 7, 1000, 33]
    cfg = {"narrow": dict(m_spa=16, ln_bot=[64, 16], ln_top=[128, 64, 1], B=700),
           "wide": dict(m_spa=32, ln_bot=[512, 256, 32], ln_top=[1024, 512, 256, 1], B=2304),
           "ragged": dict(m_spa=24, ln_bot=[100, 36, 24], ln_top=[136, 100, 40, 1], B=517)}[cfg_name]
    B = cfg.pop("B")
    cfg.update(ln_emb=ln_emb, dense_dim=13)
    o = DLRMOracle(dtype=np.float32, seed=5, reference_compat=compat, **cfg)
    m = rt.DLRMModel(reference_compat=compat, fp16_mlp=True, **cfg)
    m.param("emb").write(np.concatenate(o.emb))
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            b[:] = rng.normal(size=b.shape).astype(np.float32) * 0.1
            m.param(nm + "_w", l).write(W); m.param(nm + "_b", l).write(b.reshape(1, -1))
    opt, oo = rt.Optimizer.sgd(0.02), orc.SGD(0.02)     # (at lr 0.2 units go borderline-dead and fp16 noise flips relu masks: 22 % seen)
    B_full = B
    for step in range(4):
        # the batch size changes from call to call (the last batch of an epoch: tf2_examples/dlrm_criteo.py batches without
        # drop_remainder): the split-K workspaces and their reduce descriptors are sized once, for the largest batch
        B = (B_full, B_full // 2 + 3, B_full // 5 + 1, B_full)[step]
        tol = 0.06 if B >= 500 else 0.15                   # (fewer samples average less of the fp16 rounding out)
        dense = np.log1p(rng.integers(0, 100, (B, 13))).astype(np.float32)
        sparse = np.stack([rng.integers(0, n, B) for n in ln_emb], 1).astype(np.int32)
        label = (rng.uniform(size=B) < 0.3).astype(np.float32)
        before = [(nm, l, o.__dict__[nm][l][0].copy(), o.__dict__[nm][l][1].copy()) for nm in ("bot", "top") for l in range(len(o.__dict__[nm]))]
        l16 = m.step(opt, dense, sparse, label)[0]
        l32 = o.step(dense, sparse, label, oo)
        assert abs(l16 - l32) < 5e-3 * abs(l32), (step, l16, l32)
        for nm, l, W0, b0 in before:                       # every dense parameter's UPDATE of this step, to fp16 accuracy
            W1, b1 = o.__dict__[nm][l]
            dW, db = m.param(nm + "_w", l).read() - W0, m.param(nm + "_b", l).read().reshape(-1) - b0.reshape(-1)
            rW, rb = W1 - W0, (b1 - b0).reshape(-1)
            assert np.abs(dW - rW).max() < tol * np.abs(rW).max() + 1e-7, (step, nm, l, "W")
            assert np.abs(db - rb).max() < tol * np.abs(rb).max() + 1e-7, (step, nm, l, "b")   # (fp16 operands through up to 7 chained products)
            # the device keeps training from ITS parameters: re-sync so the per-step comparison stays first order
            m.param(nm + "_w", l).write(W1); m.param(nm + "_b", l).write(b1.reshape(1, -1))
        m.param("emb").write(np.concatenate(o.emb))
    assert np.abs(m.inference(dense, sparse) - o.inference(dense, sparse)).max() < 5e-3


@pytest.mark.parametrize("m_spa,n_emb,itself,optname", [(32, 26, False, "sgd"), (64, 5, True, "sgd"), (128, 26, False, "adagrad"),
                                                        (32, 31, True, "sgd"), (64, 16, False, "sgd")])
def test_dlrm_wide_embeddings_mfma_interaction(m_spa, n_emb, itself, optname):
    """dim 32 / 64 / 128 with up to 32 feature slots: the feature interaction runs on the MFMA kernels
    (interact_*_mfma_kernel; 16 feature slots = exactly one row tile, 32 = the maximum), the MLP products on the
    128x128 fp32 MFMA kernel; tiny tables (<= 64 rows) take the LDS gradient path with SGD."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(4)
    ln_emb = [int(x) for x in rng.integers(3, 3000, n_emb)]
    ln_emb[0], ln_emb[-1] = 3, 40
    P = (n_emb + 1) * (n_emb + 2) // 2 if itself else (n_emb + 1) * n_emb // 2
    cfg = dict(m_spa=m_spa, ln_emb=ln_emb, ln_bot=[48, m_spa], ln_top=[96, 32, 1], dense_dim=13, arch_interaction_itself=itself)
    o = DLRMOracle(dtype=np.float32, seed=3, reference_compat=False, loss_func="bce", **cfg)
    m = rt.DLRMModel(reference_compat=False, loss_func="bce", **cfg)
    assert o.top[0][0].shape[0] == m_spa + P
    m.param("emb").write(np.concatenate(o.emb))
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            m.param(nm + "_w", l).write(W); m.param(nm + "_b", l).write(b.reshape(1, -1))
    B = 777
    dense = np.log1p(rng.integers(0, 100, (B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, B) for n in ln_emb], 1).astype(np.int32)
    label = (rng.uniform(size=B) < 0.25).astype(np.float32)
    assert rel_err(m.inference(dense, sparse), o.inference(dense, sparse)) < 1e-5
    opt, oo = ((rt.Optimizer.sgd(0.05), orc.SGD(0.05)) if optname == "sgd" else (rt.Optimizer.adagrad(0.05, 0.1, 1e-7), orc.Adagrad(0.05, 0.1, 1e-7)))
    for s in range(3):
        l = m.step(opt, dense, sparse, label)[0]
        lr = o.step(dense, sparse, label, oo)
        assert abs(l - lr) <= 1e-5 * abs(lr)
    assert rel_err(m.param("emb").read(), np.concatenate(o.emb)) < 1e-5
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            assert rel_err(m.param(nm + "_w", l).read(), W) < 1e-5, (nm, l)
            assert rel_err(m.param(nm + "_b", l).read().reshape(-1), b) < 1e-5, (nm, l)


@pytest.mark.parametrize("m_spa,beta2", [(4, 0.999), (128, 0.999), (32, 0.95)])
def test_dlrm_lazy_adam_is_the_dense_decay_adam(m_spa, beta2, monkeypatch):
    """dlrm_criteo.py:31 trains with Keras Adam, whose TF-2.0 sparse apply moves every embedding row every step.
    The embedding table takes that rule lazily (rows replay their gradient-free steps when next gathered or given a
    gradient); 40 steps on tables where most rows wait many steps between references, with an inference (touch of
    its rows only) and a parameter read (full flush) in the middle, against the oracle's dense rule — and the
    whole-table-sweep form of the library (ORX_ADAM_DENSE=1) against the same numbers."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(5)
    ln_emb = [3, 40, 30000, 700, 9000, 20]
    # (reference_compat would reproduce the reference's triangle bug: zero embedding gradients)
    cfg = dict(m_spa=m_spa, ln_emb=ln_emb, ln_bot=[16, m_spa], ln_top=[64, 32, 1], dense_dim=13, reference_compat=False)
    B, K = 96, 40
    dense = np.log1p(rng.integers(0, 100, (K, B, 13))).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, (K, B)) for n in ln_emb], 2).astype(np.int32)
    label = (rng.uniform(size=(K, B)) < 0.25).astype(np.float32)
    o = DLRMOracle(dtype=np.float64, seed=2, **cfg)
    oo = orc.AdamTFSparse(0.002, 0.9, beta2, 1e-7)
    ref_loss, ref_mid_pred, ref_mid_emb = [], None, None
    start = [np.concatenate(o.emb).astype(np.float32)] + [(W.astype(np.float32), b.astype(np.float32)) for W, b in o.bot + o.top]
    for s in range(K):
        ref_loss.append(o.step(dense[s], sparse[s], label[s], oo))
        if s == 14:
            ref_mid_pred = o.inference(dense[0], sparse[0])
        if s == 24:
            ref_mid_emb = np.concatenate(o.emb).copy()
    for form in ("lazy", "dense"):
        if form == "dense":
            monkeypatch.setenv("ORX_ADAM_DENSE", "1")
        m = rt.DLRMModel(**cfg)
        m.param("emb").write(start[0])
        for nm, n0, cnt in (("bot", 1, len(o.bot)), ("top", 1 + len(o.bot), len(o.top))):
            for l in range(cnt):
                m.param(nm + "_w", l).write(start[n0 + l][0]); m.param(nm + "_b", l).write(start[n0 + l][1].reshape(1, -1))
        opt = rt.Optimizer.adam(0.002, 0.9, beta2, 1e-7)
        loss = []
        for lo, hi in ((0, 15), (15, 25), (25, K)):
            loss += list(m.step(opt, dense[lo:hi].reshape(-1, 13), sparse[lo:hi].reshape(-1, len(ln_emb)), label[lo:hi].reshape(-1), K=hi - lo))
            if hi == 15:
                assert rel_err(m.inference(dense[0], sparse[0]), ref_mid_pred) < 5e-5, form
            if hi == 25:
                assert rel_err(m.param("emb").read(), ref_mid_emb) < 5e-5, form
        assert np.abs(np.array(loss) - np.array(ref_loss)).max() <= 5e-5 * np.abs(ref_loss).max(), form
        assert rel_err(m.param("emb").read(), np.concatenate(o.emb)) < 5e-5, form
        assert rel_err(opt.slot(m.param("emb"), 0), np.concatenate([oo.m[("emb", f)] for f in range(len(ln_emb))])) < 1e-4, form
        assert rel_err(opt.slot(m.param("emb"), 1), np.concatenate([oo.v[("emb", f)] for f in range(len(ln_emb))])) < 5e-4, form
        assert rel_err(m.param("top_w", 0).read(), o.top[0][0]) < 1e-4, form


def test_dlrm_lazy_adam_long_gaps_take_the_bounded_replay():
    """Rows of a large table wait hundreds of steps between references: the replay then stops once the update
    term can no longer change the weight in fp32 (~170 steps at the default betas) and finishes m, v in closed
    form.  500 steps of a tiny batch on a 6000-row table (a row is referenced every ~750 steps on average, a few
    rows many times), checked against the oracle's dense rule on every row -- referenced, waiting and never
    touched -- including after the flush that a full read triggers (same bound, per element)."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(9)
    ln_emb = [6000, 5]
    cfg = dict(m_spa=16, ln_emb=ln_emb, ln_bot=[8, 16], ln_top=[16, 1], dense_dim=4, reference_compat=False)
    B, K = 8, 500
    dense = rng.uniform(0, 2, (K, B, 4)).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, (K, B)) for n in ln_emb], 2).astype(np.int32)
    sparse[::7, 0, 0] = 11                               # one row comes back every 7 steps
    sparse[3, 1, 0] = 4242; sparse[K - 2, 1, 0] = 4242   # one row exactly twice, 495 steps apart
    label = (rng.uniform(size=(K, B)) < 0.3).astype(np.float32)
    o = DLRMOracle(dtype=np.float64, seed=4, **cfg)
    oo = orc.AdamTFSparse(0.01, 0.9, 0.999, 1e-7)
    m = rt.DLRMModel(**cfg)
    m.param("emb").write(np.concatenate(o.emb).astype(np.float32))
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            m.param(nm + "_w", l).write(W.astype(np.float32)); m.param(nm + "_b", l).write(b.astype(np.float32).reshape(1, -1))
    opt = rt.Optimizer.adam(0.01, 0.9, 0.999, 1e-7)
    loss = m.step(opt, dense.reshape(-1, 4), sparse.reshape(-1, 2), label.reshape(-1), K=K)
    ref = [o.step(dense[s], sparse[s], label[s], oo) for s in range(K)]
    assert np.abs(np.array(loss) - np.array(ref)).max() <= 1e-4 * np.abs(ref).max()
    emb, want = m.param("emb").read(), np.concatenate(o.emb)
    e = np.abs(emb - want).max(axis=1) / np.abs(want).max()
    assert (e > 1e-4).mean() <= 2e-3 and e.max() < 5e-3, (float(e.max()), float((e > 1e-4).mean()))
    mslot, vslot = opt.slot(m.param("emb"), 0), opt.slot(m.param("emb"), 1)
    wm, wv = np.concatenate([oo.m[("emb", f)] for f in range(2)]), np.concatenate([oo.v[("emb", f)] for f in range(2)])
    assert np.abs(mslot - wm).max() <= 1e-3 * np.abs(wm).max() and np.abs(vslot - wv).max() <= 1e-3 * np.abs(wv).max()
    # Per referenced row, the MEDIAN element ratio of the slots against the oracle: m has decayed by up to 0.9^495 and v
    # by 0.999^495 there, so only a replay that takes exactly the right number of steps (loop + closed-form tail)
    # matches: one step off is 10 % in m, 0.1 % in v.  (Single elements may differ more: a gradient element that
    # nearly cancels in fp32 differs from the fp64 oracle's by several percent, and v carries its square.)
    live_rows = np.where((np.abs(wv) > 1e-30).all(axis=1) & (np.abs(wm) > 1e-30).all(axis=1))[0]
    assert live_rows.size > 1000
    rv = np.median(vslot[live_rows] / wv[live_rows], axis=1)
    rm = np.median(mslot[live_rows] / wm[live_rows], axis=1)
    assert np.abs(rv - 1).max() < 2e-3, float(np.abs(rv - 1).max())     # (a row's gradient scale itself carries ~3e-4 of fp32 noise)
    assert np.abs(rm - 1).max() < 1e-2, float(np.abs(rm - 1).max())
