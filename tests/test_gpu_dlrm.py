"""GPU parity of the DLRM train step (recommenders/dlrm.py:63-100, tf2_examples/dlrm_criteo.py:42-48) against the
torch-autograd golden fixtures and the NumPy oracle.

Every test runs its step sequence TWICE from the same start and requires the two results to be bit-identical (the step has no
fp32 atomics and no arrival-order decisions: split-K partial products, bias-gradient column sums and the embedding rows'
gradient sums are all added in a fixed order).  Batches are drawn away from the network's relu ties (tests/dlrm_util.py), and
what remains is held to 1e-5 on every parameter UPDATE in exact mode; the fp16-MLP mode is held to 1e-4 against the oracle
with fp16-rounded operands (oracle/dlrm_oracle.py: operand_dtype)."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err, OPT_KW, TOL, TOL_ADAM
from dlrm_util import DELTA, assert_fp16_updates, assert_same_bits, assert_updates, draw_batch, load_model, params_of, record, round_to_fp32, snapshot, update_err

pytestmark = pytest.mark.gpu

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


def _pair(rt, orc, name, **kw):
    if name == "sgd":
        lr = kw.get("lr", 0.05); return rt.Optimizer.sgd(lr), orc.SGD(lr)
    if name == "adagrad":
        return rt.Optimizer.adagrad(0.05, 0.1, 1e-7), orc.Adagrad(0.05, 0.1, 1e-7)
    a = (kw.get("lr", 0.002), 0.9, kw.get("beta_2", 0.999), 1e-7)
    return rt.Optimizer.adam(*a), orc.AdamTFSparse(*a)


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

    def run():
        m = rt.DLRMModel(**CFG, **KW[name])
        _load(m, g)
        opt = _opt(rt, optkind)
        losses = [m.step(opt, g["dense"], g["sparse"], g["label"])[0] for _ in range(2)]
        out = {"emb": m.param("emb").read(), "losses": np.array(losses)}
        for nm, n in (("bot", 2), ("top", 3)):
            for l in range(n):
                out[f"{nm}{l}W"] = m.param(nm + "_w", l).read(); out[f"{nm}{l}b"] = m.param(nm + "_b", l).read().reshape(-1)
        return out

    r, r2 = run(), run()
    assert_same_bits(r, r2)
    tol = 1e-5
    assert rel_err(r["losses"], g["losses"]) < tol
    emb = r["emb"]
    ref = np.concatenate([g[f"out_emb{f}"] for f in range(3)])
    assert rel_err(emb, ref) < tol
    for nm, n in (("bot", 2), ("top", 3)):
        for l in range(n):
            assert rel_err(r[f"{nm}{l}W"], g[f"out_{nm}{l}W"]) < 5 * tol, (nm, l)
            assert rel_err(r[f"{nm}{l}b"], g[f"out_{nm}{l}b"]) < 5 * tol, (nm, l)
    if name == "compat":      # the reference's bug: embeddings never move
        assert np.array_equal(emb, np.concatenate([g[f"in_emb{f}"] for f in range(3)]))


def _run_steps(rt, cfg, o_start, batches, make_opt, model_kw, K_call=1, mid=None):
    """a fresh device model loaded with `o_start`'s parameters, stepped through `batches`; returns (losses, snapshot, model, opt)"""
    m = rt.DLRMModel(**cfg, **model_kw)
    load_model(m, o_start)
    opt = make_opt()
    losses = []
    if K_call == 1:
        for i, (de, sp, la) in enumerate(batches):
            losses.append(m.step(opt, de, sp, la)[0])
            if mid is not None:
                mid(i, m)
    else:
        de = np.concatenate([b[0] for b in batches]); sp = np.concatenate([b[1] for b in batches]); la = np.concatenate([b[2] for b in batches])
        losses = list(m.step(opt, de, sp, la, K=len(batches)))
    return np.array(losses), snapshot(m, o_start, opt), m, opt


def _exact_case(name, cfg, model_kw, oracle_kw, optname, B, steps, seed, tol=TOL, K_call=1, delta=DELTA, bias_noise=0.0):
    """exact (fp32) mode against the fp64 oracle: loss and every parameter update of the whole sequence within `tol`"""
    import copy
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(seed)
    o = DLRMOracle(dtype=np.float64, seed=seed + 1, **cfg, **oracle_kw)
    if bias_noise:
        for W, b in o.bot + o.top:
            b[:] = rng.normal(size=b.shape) * bias_noise
    round_to_fp32(o)
    o0 = copy.deepcopy(o)
    _, oo = _pair(rt, orc, optname)
    batches, ref, stats = [], [], {}
    for s in range(steps):
        bt = draw_batch(o, rng, B, cfg["ln_emb"], delta=delta, dense_dim=cfg["dense_dim"], stats=stats)
        batches.append(bt)
        ref.append(o.step(*bt, oo))
    start = {k: v.astype(np.float32) for k, v in params_of(o0).items()}
    runs = [_run_steps(rt, cfg, o0, batches, lambda: _pair(rt, orc, optname)[0], model_kw, K_call) for _ in range(2)]
    assert_same_bits(runs[0][1], runs[1][1])
    assert np.array_equal(runs[0][0], runs[1][0])
    loss, got = runs[0][0], runs[0][1]
    want = params_of(o)
    errs = {k: update_err(start[k], got[k], want[k], steps)[1] for k in want}
    record(name, loss=float(np.abs(loss - np.array(ref)).max() / np.abs(ref).max()), dropped=stats["drawn"] - stats["kept"], kept=stats["kept"], **errs)
    assert np.abs(loss - np.array(ref)).max() <= tol * np.abs(ref).max()
    assert_updates(start, got, want, tol, what=name, steps=steps)
    return runs[0][2], o, batches


@pytest.mark.parametrize("compat", [True, False])
def test_dlrm_example_shapes_vs_oracle(compat):
    """tf2_examples/dlrm_criteo.py shapes: dim 4, bottom [8,4], top [128,64,1], 26 tables, batch 1024."""
    rng = np.random.default_rng(0)
    ln_emb = [int(x) for x in rng.integers(3, 2000, 26)]
    cfg = dict(m_spa=4, ln_emb=ln_emb, ln_bot=[8, 4], ln_top=[128, 64, 1], dense_dim=13)
    m, o, batches = _exact_case(f"example_compat{int(compat)}", cfg, dict(reference_compat=compat), dict(reference_compat=compat), "adagrad", 1024, 2, seed=2)
    de, sp, _ = batches[0]
    assert rel_err(m.inference(de, sp), o.inference(de, sp)) < 1e-5


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


TOL_FP16 = 1e-4      # fp16-MLP mode against the oracle with fp16-rounded operands (fp32 accumulation on the device, fp64 in the oracle)
# ... for the dense parameters, whose gradients are sums over the batch.  An EMBEDDING row's update is one sample's gradient
# (or a few), and that passes through fp16 roundings of per-sample scalars -- the head's dZ is ONE number per sample: two
# correct implementations that hold it 1e-6 apart round it to neighbouring fp16 values for 0.4 % of the samples, and those
# samples' embedding gradients then differ by an fp16 ulp, 2^-11 .. 2^-10 (narrow hidden layers add their own roundings).
TOL_FP16_EMB = 1.5e-3


@pytest.mark.parametrize("compat", [False, True])
@pytest.mark.parametrize("cfg_name", ["small", "narrow", "wide", "ragged", "thinbot", "fewdense"])
def test_dlrm_fp16_mlp_mode_against_the_fp16_operand_oracle(cfg_name, compat):
    """ORX_DLRM_FP16_MLP (north_star: the dense MLPs on fp16 MFMA): every MLP product rounds its operands to fp16 once and
    accumulates in fp32, the backward pass under a power-of-two loss scale -- exactly what DLRMOracle(operand_dtype=float16)
    restates.  Layer widths exercise every tile shape of kernels_gemm16.hip (256x128 / 128x128 / 128x64 blocks, partial
    tiles), widths off a multiple of 8 and hidden layers below 32 units or fewer than 8 inputs that fall back to the
    fp32-operand kernels (which round in the kernel: the same arithmetic), the fp16-only ("lean") activations, both interaction
    modes, and a batch size that changes from call to call.

    (1) step by step, every step from the oracle's (fp32) parameters: loss to 1e-4; every dense update within 1e-4 + 4 / B of
    its largest element and its projection on the oracle's update within 5e-4 of 1; the embedding rows of all but 4 samples within
    TOL_FP16_EMB (dlrm_util.assert_fp16_updates has the accounting: the 4 / B is the share of the few samples per step whose relu
    units two correct fp16 implementations resolve differently).  (2) FREE-RUNNING, the same four steps without
    re-synchronising, twice: bit-identical, and first-order faithful (a weight copy refreshed one step late or a gradient
    applied twice is an error of ~1)."""
    import copy
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    rng = np.random.default_rng(11)
    ln_emb = [50, 300, 7, 1000, 33]
    cfg = {"small": dict(m_spa=32, ln_bot=[96, 32], ln_top=[200, 72, 1], B=333, dense_dim=13),
           "narrow": dict(m_spa=16, ln_bot=[64, 16], ln_top=[128, 64, 1], B=700, dense_dim=13),
           "wide": dict(m_spa=32, ln_bot=[512, 256, 32], ln_top=[1024, 512, 256, 1], B=2304, dense_dim=13),
           "ragged": dict(m_spa=24, ln_bot=[100, 36, 24], ln_top=[136, 100, 40, 1], B=517, dense_dim=13),
           "thinbot": dict(m_spa=64, ln_bot=[64, 16, 64], ln_top=[128, 24, 64, 1], B=600, dense_dim=13),     # hidden layers < 32 wide
           "fewdense": dict(m_spa=32, ln_bot=[64, 32], ln_top=[96, 64, 1], B=450, dense_dim=5)}[cfg_name]    # fewer than 8 dense features
    B_full = cfg.pop("B")
    cfg.update(ln_emb=ln_emb)
    o = DLRMOracle(dtype=np.float64, operand_dtype=np.float16, seed=5, reference_compat=compat, **cfg)
    for W, b in o.bot + o.top:
        b[:] = rng.normal(size=b.shape) * 0.1
    round_to_fp32(o)
    oo = orc.SGD(0.02)
    batches, ref, states, stats = [], [], [copy.deepcopy(o)], {}
    for step in range(4):
        # (the last batch of an epoch is smaller: dlrm_criteo.py batches without drop_remainder; the split-K workspaces and
        # partial-row buffers are sized once, for the largest batch)
        B = (B_full, B_full // 2 + 3, B_full // 5 + 1, B_full)[step]
        bt = draw_batch(o, rng, B, ln_emb, label_p=0.3, dense_dim=cfg["dense_dim"], stats=stats)
        batches.append(bt); ref.append(o.step(*bt, oo))
        round_to_fp32(o)                                    # the model is STORED in fp32, on the device as in TF
        states.append(copy.deepcopy(o))
    kw = dict(reference_compat=compat, fp16_mlp=True)
    # ---- (2) free-running, twice: bit-identical, and first-order faithful
    start = {k: v.astype(np.float32) for k, v in params_of(states[0]).items()}
    runs = [_run_steps(rt, cfg, states[0], batches, lambda: rt.Optimizer.sgd(0.02), kw) for _ in range(2)]
    assert_same_bits(runs[0][1], runs[1][1])
    want = params_of(o)
    free = {k: update_err(start[k], runs[0][1][k], want[k], 4)[1] for k in want}
    assert_updates(start, runs[0][1], want, 1e-2 + 16.0 / B_full, what=cfg_name + " free-running", steps=4, tol_of={"emb": 0.2})
    # ---- (1) strict, every step from the oracle's parameters
    m = rt.DLRMModel(**cfg, **kw)
    opt = rt.Optimizer.sgd(0.02)
    worst = {}
    n_emb = len(ln_emb)
    for step, bt in enumerate(batches):
        load_model(m, states[step])
        before = {k: v.astype(np.float32) for k, v in params_of(states[step]).items()}
        loss = m.step(opt, *bt)[0]
        assert abs(loss - ref[step]) <= TOL_FP16 * abs(ref[step]), (step, loss, ref[step])
        got, want_s = snapshot(m, o), params_of(states[step + 1])
        assert_fp16_updates(before, got, want_s, len(bt[2]), n_emb, TOL_FP16, TOL_FP16_EMB, what=f"{cfg_name} step {step}", stats=worst)
    record(f"fp16_{cfg_name}_compat{int(compat)}", dropped=stats["drawn"] - stats["kept"], kept=stats["kept"],
           dense=max(v for k, v in worst.items() if k != "emb" and not k.endswith("_proj") and k != "emb_bad_rows"),
           proj=max(v for k, v in worst.items() if k.endswith("_proj")), emb=worst.get("emb", 0.0), emb_bad_rows=worst.get("emb_bad_rows", 0), free=max(free.values()))
    de, sp, _ = batches[-1]
    load_model(m, o)
    assert np.abs(m.inference(de, sp) - o.inference(de, sp)).max() < TOL_FP16


@pytest.mark.parametrize("m_spa,n_emb,itself,optname", [(32, 26, False, "sgd"), (64, 5, True, "sgd"), (128, 26, False, "adagrad"),
                                                        (32, 31, True, "sgd"), (64, 16, False, "sgd")])
def test_dlrm_wide_embeddings_mfma_interaction(m_spa, n_emb, itself, optname):
    """dim 32 / 64 / 128 with up to 32 feature slots: the feature interaction runs on the MFMA kernels
    (interact_*_mfma_kernel; 16 feature slots = exactly one row tile, 32 = the maximum), the MLP products on the
    128x128 fp32 MFMA kernel; tables of 3 and 40 rows (hundreds of gradient rows per table row) next to tables of thousands."""
    rng = np.random.default_rng(4)
    ln_emb = [int(x) for x in rng.integers(3, 3000, n_emb)]
    ln_emb[0], ln_emb[-1] = 3, 40
    cfg = dict(m_spa=m_spa, ln_emb=ln_emb, ln_bot=[48, m_spa], ln_top=[96, 32, 1], dense_dim=13)
    kw = dict(reference_compat=False, loss_func="bce", arch_interaction_itself=itself)
    m, o, batches = _exact_case(f"wide_{m_spa}_{n_emb}_{optname}", cfg, kw, kw, optname, 777, 3, seed=3)
    de, sp, _ = batches[0]
    assert rel_err(m.inference(de, sp), o.inference(de, sp)) < 1e-5


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
    o = round_to_fp32(DLRMOracle(dtype=np.float64, seed=2, **cfg))
    oo = orc.AdamTFSparse(0.002, 0.9, beta2, 1e-7)
    ref_loss, ref_mid_pred, ref_mid_emb = [], None, None
    start = [np.concatenate(o.emb).astype(np.float32)] + [(W.astype(np.float32), b.astype(np.float32)) for W, b in o.bot + o.top]
    dense, sparse, label = [], [], []
    for s in range(K):
        # Adam pins the weights to TOL_ADAM = 5e-5 only (conftest.py): the batches keep that much distance from the relu ties
        de, sp, la = draw_batch(o, rng, B, ln_emb, delta=2 * TOL_ADAM)
        dense.append(de); sparse.append(sp); label.append(la)
        ref_loss.append(o.step(de, sp, la, oo))
        if s == 14:
            ref_mid_pred = o.inference(dense[0], sparse[0])
        if s == 24:
            ref_mid_emb = np.concatenate(o.emb).copy()
    dense, sparse, label = np.stack(dense), np.stack(sparse), np.stack(label)
    for form in ("lazy", "dense"):
        if form == "dense":
            monkeypatch.setenv("ORX_ADAM_DENSE", "1")
        snaps = []
        for rep in range(2):
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
                    assert rel_err(m.inference(dense[0], sparse[0]), ref_mid_pred) < TOL_ADAM, form
                if hi == 25:
                    assert rel_err(m.param("emb").read(), ref_mid_emb) < TOL_ADAM, form
            snaps.append(snapshot(m, o, opt))
        assert_same_bits(snaps[0], snaps[1])
        assert np.abs(np.array(loss) - np.array(ref_loss)).max() <= TOL_ADAM * np.abs(ref_loss).max(), form
        assert rel_err(m.param("emb").read(), np.concatenate(o.emb)) < TOL_ADAM, form
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
    o = round_to_fp32(DLRMOracle(dtype=np.float64, seed=4, **cfg))
    oo = orc.AdamTFSparse(0.002, 0.9, 0.999, 1e-7)
    w0 = np.concatenate(o.emb).astype(np.float32)
    d0 = [(W.astype(np.float32), b.astype(np.float32)) for W, b in o.bot + o.top]
    dense, sparse, label, ref = [], [], [], []
    for s in range(K):
        def fix(sp, s=s):
            if s % 7 == 0:
                sp[0, 0] = 11                                # one row comes back every 7 steps
            if s in (3, K - 2):
                sp[1, 0] = 4242                              # one row exactly twice, 495 steps apart
            return 2
        # 500 Adam steps: the weights are pinned to ~1e-4 by then (conftest.TOL_ADAM per step), and so is every pre-activation
        de, sp, la = draw_batch(o, rng, B, ln_emb, delta=3e-3, label_p=0.3, dense=lambda r, n: r.uniform(0, 2, (n, 4)).astype(np.float32), fix=fix)
        dense.append(de); sparse.append(sp); label.append(la)
        ref.append(o.step(de, sp, la, oo))
    dense, sparse, label = np.stack(dense), np.stack(sparse), np.stack(label)
    assert (sparse[::7, 0, 0] == 11).all() and sparse[3, 1, 0] == 4242 and sparse[K - 2, 1, 0] == 4242
    snaps = []
    for rep in range(2):
        m = rt.DLRMModel(**cfg)
        m.param("emb").write(w0)
        for nm, n0, cnt in (("bot", 0, len(o.bot)), ("top", len(o.bot), len(o.top))):
            for l in range(cnt):
                m.param(nm + "_w", l).write(d0[n0 + l][0]); m.param(nm + "_b", l).write(d0[n0 + l][1].reshape(1, -1))
        opt = rt.Optimizer.adam(0.002, 0.9, 0.999, 1e-7)
        loss = m.step(opt, dense.reshape(-1, 4), sparse.reshape(-1, 2), label.reshape(-1), K=K)
        snaps.append(dict(snapshot(m, o, opt), loss=np.array(loss)))
    assert_same_bits(snaps[0], snaps[1])
    emb, want = snaps[0]["emb"], np.concatenate(o.emb)
    e = np.abs(emb - want).max(axis=1) / np.abs(want).max()
    mslot, vslot = snaps[0]["emb_slot0"], snaps[0]["emb_slot1"]
    wm, wv = np.concatenate([oo.m[("emb", f)] for f in range(2)]), np.concatenate([oo.v[("emb", f)] for f in range(2)])
    record("adam_long_gaps", loss=float(np.abs(np.array(loss) - np.array(ref)).max() / np.abs(ref).max()), emb_max=float(e.max()),
           emb_frac_1e4=float((e > 1e-4).mean()), m=float(np.abs(mslot - wm).max() / np.abs(wm).max()), v=float(np.abs(vslot - wv).max() / np.abs(wv).max()))
    assert np.abs(np.array(loss) - np.array(ref)).max() <= 1e-4 * np.abs(ref).max()
    assert e.max() < 2e-4, (float(e.max()), float((e > 1e-4).mean()))          # (no row beyond Adam's own noise: see TOL_ADAM)
    assert np.abs(mslot - wm).max() <= 1e-3 * np.abs(wm).max() and np.abs(vslot - wv).max() <= 1e-3 * np.abs(wv).max()
    # Per referenced row, the MEDIAN element ratio of the slots against the oracle: m has decayed by up to 0.9^495 and v
    # by 0.999^495 there, so only a replay that takes exactly the right number of steps (loop + closed-form tail)
    # matches: one step off is 10 % in m, 0.1 % in v.  (Single elements may differ more: a gradient element that
    # nearly cancels in fp32 differs from the fp64 oracle's by several percent, and v carries its square.)
    live_rows = np.where((np.abs(wv) > 1e-30).all(axis=1) & (np.abs(wm) > 1e-30).all(axis=1))[0]
    assert live_rows.size > 1000
    rv = np.abs(np.median(vslot[live_rows] / wv[live_rows], axis=1) - 1)
    rm = np.abs(np.median(mslot[live_rows] / wm[live_rows], axis=1) - 1)
    record("adam_long_gaps_slots", rv_max=float(rv.max()), rv_frac=float((rv > 2e-3).mean()), rm_max=float(rm.max()), rm_frac=float((rm > 1e-2).mean()),
           rv_med=float(np.median(rv)), rm_med=float(np.median(rm)))
    # A wrong replay count shifts EVERY waiting row.  A single row may be further off: its slots are one sample's gradient (B = 8),
    # and after hundreds of Adam steps the two sides' dense weights are ~1e-3 apart (TOL_ADAM per step), enough for a relu unit
    # of one sample to fall on the other side; (a row's gradient scale itself carries ~3e-4 of fp32 noise)
    assert np.median(rv) < 2e-4 and (rv > 2e-3).mean() <= 0.01, (float(np.median(rv)), float((rv > 2e-3).mean()))
    assert np.median(rm) < 2e-4 and (rm > 1e-2).mean() <= 0.01, (float(np.median(rm)), float((rm > 1e-2).mean()))


def test_dlrm_fp16_staging_forms_are_bit_identical():
    """The fp16 products exist in several forms behind environment switches: tiles staged through registers or by the
    LDS-DMA with two or three stages (forward / input gradient: ORX_GEMM16_DMA, weight gradient: ORX_GEMM16_TN_DMA), ordinary,
    nontemporal or write-through output stores (ORX_GEMM16_NTS).  None of them changes which products are summed in which order: two
    steps at shapes that take every tile configuration end in the same bits under each of them."""
    import subprocess
    import sys as _sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dlrm_variant_worker.py")
    # two families: within each, no switch changes which products are summed in which order.  A: the default weight-gradient kernel (round 6: eight
    # wavefronts in two K groups, a slice = first half + second half); B: the one-group forms of rounds 3-5 (ORX_GEMM16_TN_DMA=3 | 2 | 0), which the
    # grouped dW + dX launch (gemm16_group_kernel) also carries -- so the grouped and the ungrouped step agree bit for bit there.
    fam_a = [{}, {"ORX_GEMM16_NTS": "0"}, {"ORX_GEMM16_NTS": "1"},
             {"ORX_DLRM_COLPARTS_LAUNCH": "1"},            # the partial-row sums added by the optimizer launch vs a reduce launch
             {"ORX_DLRM_FINISH_LAUNCH": "1"},              # the sorted apply's finish pass as a launch of its own vs carried by the optimizer launch
             {"ORX_DLRM_HEAD_FWD_LAUNCH": "1"},            # the head's forward as a launch of its own vs inside the head's backward launch
             {"ORX_INTERACT_NO_XCD": "1"},                 # the interaction kernels' samples dealt round-robin to the XCDs vs in contiguous eighths
             {"ORX_GEMM16_NO_MASK": "1"},                  # relu' from the fp16 output itself instead of the forward launch's mask words
             {"ORX_DLRM_NO_PAD_DX": "1"},                  # the first top layer's input gradient on its 479 columns instead of 480
             {"ORX_DLRM_DEFER_DW": "1"},                   # the top MLP's weight gradients beside the interaction backward (side stream; measured slower: off)
             {"ORX_GEMM16_WAVE_TILE": "128"}]              # the 256 x 128 tile on four wavefronts of 128 x 64 (measured slower: off)
    # C: no grouped launches (every weight gradient on the two-group kernel): the staging forms of the forward / input-gradient products
    # (ORX_GEMM16_DMA = 0 | 2 switch the grouped launch off as well)
    fam_c = [{"ORX_GEMM16_NO_GROUP": "1"}, {"ORX_GEMM16_DMA": "0"}, {"ORX_GEMM16_DMA": "2"}]
    fam_b = [{"ORX_GEMM16_TN_DMA": "3"}, {"ORX_GEMM16_TN_DMA": "0"}, {"ORX_GEMM16_TN_DMA": "2"},
             {"ORX_GEMM16_TN_DMA": "3", "ORX_GEMM16_NO_GROUP": "1"}]      # a layer's dW and dX in one launch vs two
    for variants in (fam_a, fam_b, fam_c):
        digests = []
        for v in variants:
            env = dict(os.environ); env.update(v)
            r = subprocess.run([_sys.executable, worker], env=env, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, f"{v}: {r.stderr[-2000:]}"
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST ")]
            assert line, f"{v}: no digest in {r.stdout[-500:]}"
            digests.append(line[-1])
        assert all(d == digests[0] for d in digests), [str(v) for v, d in zip(variants, digests) if d != digests[0]]
    # round 6: the loss folded into the head's backward (per-workgroup fp64 partials summed once per call) against the dlrm_loss_kernel
    # launch per step: another order of the loss SUM, the same gradient -- every parameter bit for bit
    params = []
    for v in ({}, {"ORX_DLRM_NO_FOLDED_LOSS": "1"}):
        env = dict(os.environ); env.update(v)
        r = subprocess.run([_sys.executable, worker], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, f"{v}: {r.stderr[-2000:]}"
        params.append([ln for ln in r.stdout.splitlines() if ln.startswith("PARAMS ")][-1])
    assert params[0] == params[1], params


@pytest.mark.parametrize("m_spa,optname,K_call", [(128, "sgd", 1), (128, "adagrad", 3), (64, "sgd", 3), (256, "adagrad", 1)])
def test_dlrm_rows_referenced_once_are_applied_by_the_interaction_backward(m_spa, optname, K_call, monkeypatch):
    """round 6 (north_star's "one pass"; dlrm.py:83-85 + second_order_feature_interaction.py:20-32 + tf2_examples/dlrm_criteo.py:45-46):
    an embedding row that a single lookup of the step references takes its SGD / Adagrad update inside interact_bwd_mfma_kernel
    (its gradient never reaches HBM), rows with several references keep the sorted segmented sums.  Tables of a million rows
    (every lookup a singleton), of thousands (a mix) and of 3 / 7 rows (hundreds of references per row), against the fp64 oracle;
    and the same steps with ORX_DLRM_NO_FUSED_SPARSE=1 (everything through the sorted apply): the two paths compute the same
    expression on the same gradient, every parameter agrees to one fp32 rounding of its update."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    ln_emb = [1000000, 3, 5000, 7, 200000, 1200, 50]
    cfg = dict(m_spa=m_spa, ln_emb=ln_emb, ln_bot=[32, m_spa], ln_top=[64, 32, 1], dense_dim=13)
    kw = dict(reference_compat=False)
    monkeypatch.delenv("ORX_DLRM_NO_FUSED_SPARSE", raising=False)
    m, o, batches = _exact_case(f"fusedrows_{m_spa}_{optname}_{K_call}", cfg, kw, kw, optname, 600, 3, seed=21, K_call=K_call)
    # A/B on the same start, same batches
    import copy
    from oracle.dlrm_oracle import DLRMOracle
    o0 = DLRMOracle(dtype=np.float64, seed=5, **cfg, **kw)
    round_to_fp32(o0)
    res = {}
    for name, env in (("fused", None), ("sorted", "1")):
        if env is None:
            monkeypatch.delenv("ORX_DLRM_NO_FUSED_SPARSE", raising=False)
        else:
            monkeypatch.setenv("ORX_DLRM_NO_FUSED_SPARSE", env)
        res[name] = _run_steps(rt, cfg, copy.deepcopy(o0), batches, lambda: _pair(rt, orc, optname)[0], kw, K_call)
    monkeypatch.delenv("ORX_DLRM_NO_FUSED_SPARSE", raising=False)
    assert np.array_equal(res["fused"][0], res["sorted"][0]) or np.allclose(res["fused"][0], res["sorted"][0], rtol=1e-6)
    a, b = res["fused"][1], res["sorted"][1]
    for k in a:
        ulp = np.spacing(np.maximum(np.abs(a[k]), np.abs(b[k])).astype(np.float32))
        assert (np.abs(a[k].astype(np.float64) - b[k]) <= 3 * ulp).all(), (k, float(np.abs(a[k] - b[k]).max()))
    # the singleton rows DID move (a path that skipped them in both kernels would leave them at their start)
    start = np.concatenate(o0.emb).astype(np.float32)
    sp0 = batches[0][1]
    assert np.abs(a["emb"][sp0[:, 0]] - start[sp0[:, 0]]).max() > 0
