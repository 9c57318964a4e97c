"""The reference's recommenders are COMPOSITIONS of its modules (openrec/tf2/recommenders/bpr.py:5-43, wrmf.py:5-40:
`LatentFactor` lookups -> `PairwiseLogLoss` / `PointwiseMSELoss` -> `tf.nn.l2_loss`).  Here the reference's own class text
is executed against this package's modules (`openrec.tf2.modules`, `tensorflow.keras.Model`, `tf.nn`, `tf.linalg` resolved by
compat.install()) and trained with the example's train step (tf2_examples/bpr_citeulike.py:33-39): the composition must run
as the fused device step -- the K-step queue of the native pairwise / pointwise entry points -- and give the oracle's
tables, not a host-side forward without gradients.  The class text comes from /root/reference here and from the git-ignored
blob __graft_entry__.build() writes on the GPU box."""
import json
import os
import warnings
import zlib

import numpy as np
import pytest

from conftest import TOL, TOL_ADAM, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref_class(fname, cls):
    ref = os.path.join("/root/reference/openrec/tf2/recommenders", fname)
    if os.path.exists(ref):
        text = open(ref).read()
    else:
        blob = os.path.join(ROOT, "tests", "golden", "_ref_scripts.bin")
        texts = json.loads(zlib.decompress(open(blob, "rb").read()).decode()) if os.path.exists(blob) else {}
        if "recommenders/" + fname not in texts:
            pytest.skip("reference recommender sources unavailable: neither /root/reference nor the blob of __graft_entry__.build()")
        text = texts["recommenders/" + fname]
    from openrec_amd.tf2 import compat
    compat.install()
    ns = {"__name__": "ref_" + fname[:-3]}
    exec(compile(text, fname, "exec"), ns)
    return ns[cls]


def _train_step(model, optimizer):
    import tensorflow as tf          # the shim (or a real TensorFlow, which these tests then do not exercise)

    @tf.function
    def train_step(*batch):
        with tf.GradientTape() as tape:
            loss_value = model(*batch)
        gradients = tape.gradient(loss_value, model.trainable_variables)
        optimizer.apply_gradients(zip(gradients, model.trainable_variables))
        return loss_value
    return train_step


def _native_calls(ctx):
    return dict(ctx.profile()) if hasattr(ctx, "profile") else None


@pytest.mark.parametrize("optk", ["sgd", "adagrad", "adam"])
def test_reference_bpr_class_text_trains_on_the_fused_path(optk):
    from openrec_amd.tf2.compat import optimizers
    from openrec_amd.tf2.modules._compose import _models
    from oracle import numpy_oracle as orc
    BPR = _ref_class("bpr.py", "BPR")
    NU, NI, D, B = 700, 900, 32, 1024
    model = BPR(dim_user_embed=D, dim_item_embed=D, total_users=NU, total_items=NI)
    names = [v.name for v in model.trainable_variables]
    assert names == ["user_latent_factor/embeddings", "item_latent_factor/embeddings", "item_bias/embeddings"]
    opt, oo = {"sgd": (optimizers.SGD(0.05), orc.SGD(0.05)), "adagrad": (optimizers.Adagrad(0.05), orc.Adagrad(0.05)),
               "adam": (optimizers.Adam(), orc.AdamTFSparse())}[optk]
    step = _train_step(model, opt)
    U, V, b = (v.numpy() for v in model.trainable_variables)
    U0 = U.copy()
    rng = np.random.default_rng(5)
    n_models = len(_models)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                   # a host fallback would warn
        out, want = [], []
        for it in range(5):
            u, p, n = (rng.integers(0, hi, B).astype(np.int32) for hi in (NU, NI, NI))
            out.append(step(u, p, n))
            want.append(orc.bpr_step(U, V, b, u, p, n, oo))
    assert len(_models) == n_models + 1                                  # one composed model for the five steps
    q = next(iter(model.user_latent_factor._composed.values()))._queue
    assert len(q.steps) == 5                                             # ... which are still QUEUED: one K=5 device call follows
    for (loss, l2), (lr, l2r) in zip(out, want):
        assert abs(float(loss) - lr) <= TOL * abs(lr) and abs(float(l2) - l2r) <= TOL * abs(l2r)
    assert len(q.steps) == 0
    tol = TOL_ADAM if optk == "adam" else TOL
    Ud, Vd, bd = (v.numpy() for v in model.trainable_variables)
    assert rel_err(Ud, U) < tol and rel_err(Vd, V) < tol and rel_err(bd, b) < tol
    assert np.abs(Ud - U0).max() > 1e-4                                  # it trained
    # inference (bpr.py:39-43) through tf.linalg.matmul(...) + tf.reshape(...): the device scorer
    pred = model.inference(np.arange(16, dtype=np.int32))
    assert pred.shape == (16, NI) and rel_err(np.asarray(pred), orc.bpr_inference(U, V, b, np.arange(16))) < 1e-4
    # eager call outside a tape: forward only
    u, p, n = (rng.integers(0, hi, B).astype(np.int32) for hi in (NU, NI, NI))
    loss, l2 = model(u, p, n)
    lr, l2r, _ = orc.bpr_forward(U, V, b, u, p, n)
    assert abs(float(loss) - lr) <= TOL * abs(lr) and abs(float(l2) - l2r) <= TOL * abs(l2r)
    assert rel_err(model.trainable_variables[0].numpy(), U) < tol
    # the composition (and its 3 tables in HBM) dies with the model that made it
    import gc
    del model, step, q, out, loss, l2
    gc.collect()
    assert len(_models) == n_models


def test_reference_wrmf_class_text_trains_on_the_fused_path():
    from openrec_amd.tf2.compat import optimizers
    from oracle import numpy_oracle as orc
    WRMF = _ref_class("wrmf.py", "WRMF")
    NU, NI, D, B = 500, 800, 64, 2048
    model = WRMF(dim_user_embed=D, dim_item_embed=D, total_users=NU, total_items=NI, a=2.0, b=0.5)
    opt, oo = optimizers.Adagrad(0.02), orc.Adagrad(0.02)
    step = _train_step(model, opt)
    U, V, b = (v.numpy() for v in model.trainable_variables)
    rng = np.random.default_rng(6)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for it in range(4):
            u, i = rng.integers(0, NU, B).astype(np.int32), rng.integers(0, NI, B).astype(np.int32)
            lab = (rng.random(B) < 0.3).astype(np.float32)
            loss, l2 = step(u, i, lab)
            lr, l2r = orc.wrmf_step(U, V, b, u, i, lab, oo, a=2.0, b_w=0.5)
            assert abs(float(loss) - lr) <= TOL * abs(lr) and abs(float(l2) - l2r) <= TOL * abs(l2r)
    Ud, Vd, bd = (v.numpy() for v in model.trainable_variables)
    assert rel_err(Ud, U) < TOL and rel_err(Vd, V) < TOL and rel_err(bd, b) < TOL
    pred = model.inference(np.arange(8, dtype=np.int32))
    assert rel_err(np.asarray(pred), orc.bpr_inference(U, V, b, np.arange(8))) < 1e-4


@pytest.mark.parametrize("optk", ["sgd", "adagrad"])
def test_reference_ucml_class_text_trains_on_the_fused_path(optk):
    """ucml.py:21-42 spells its score out in raw ops (tf.math.square(user_vec - item_vec) -> reduce_sum -> hinge): the lazy
    expression tree of modules/_expr.py must match it onto the fused UCML step, censor_vec (ucml.py:44-48) included."""
    from openrec_amd.tf2.compat import optimizers
    from oracle import numpy_oracle as orc
    UCML = _ref_class("ucml.py", "UCML")
    NU, NI, D, B = 600, 800, 32, 1024
    model = UCML(dim_user_embed=D, dim_item_embed=D, total_users=NU, total_items=NI, margin=0.7)
    opt, oo = {"sgd": (optimizers.SGD(0.05), orc.SGD(0.05)), "adagrad": (optimizers.Adagrad(0.05), orc.Adagrad(0.05))}[optk]
    step = _train_step(model, opt)
    U, V, b = (v.numpy() for v in model.trainable_variables)
    U0 = U.copy()
    rng = np.random.default_rng(8)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                   # a host fallback would warn
        for it in range(4):
            u, p, n = (rng.integers(0, hi, B).astype(np.int32) for hi in (NU, NI, NI))
            loss, l2 = step(u, p, n)
            model.censor_vec(u, p, n)
            lr, l2r = orc.ucml_step(U, V, b, u, p, n, oo, margin=0.7, do_censor=True)
            assert abs(float(loss) - lr) <= TOL * abs(lr) and abs(float(l2) - l2r) <= TOL * abs(l2r), it
        Ud, Vd, bd = (v.numpy() for v in model.trainable_variables)
        assert rel_err(Ud, U) < TOL and rel_err(Vd, V) < TOL and rel_err(bd, b) < TOL
        assert np.abs(Ud - U0).max() > 1e-4
        # inference (ucml.py:50-53): -reduce_sum(square(expand_dims(user_vec, 1) - V), -1) + reshape(b): the device L2 scorer
        pred = model.inference(np.arange(16, dtype=np.int32))
    assert pred.shape == (16, NI) and rel_err(np.asarray(pred), orc.ucml_inference(U, V, b, np.arange(16))) < 1e-4


def test_reference_gmf_class_text_trains_on_the_fused_path():
    """gmf.py:22-34: MLP([1], no bias) of (user_vec * item_vec) + item_bias -> Keras BCE with logits; l2 over the lookups and the
    Dense kernel.  The composition must run as the fused GMF step (the kernel is a device-resident dense parameter)."""
    from openrec_amd.tf2.compat import optimizers
    from oracle import numpy_oracle as orc
    GMF = _ref_class("gmf.py", "GMF")
    NU, NI, D, B = 500, 700, 32, 2048
    model = GMF(dim_user_embed=D, dim_item_embed=D, total_users=NU, total_items=NI)
    opt, oo = optimizers.SGD(0.05), orc.SGD(0.05)
    step = _train_step(model, opt)
    rng = np.random.default_rng(9)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                   # a host fallback would warn
        # an eager call builds the Dense kernel (Keras builds on first use) and is forward-only
        u, i = rng.integers(0, NU, B).astype(np.int32), rng.integers(0, NI, B).astype(np.int32)
        lab = (rng.random(B) < 0.4).astype(np.float32)
        loss, l2 = model(u, i, lab)
        names = [v.name for v in model.trainable_variables]
        assert len(names) == 4 and names[:3] == ["user_latent_factor/embeddings", "item_latent_factor/embeddings", "item_bias/embeddings"]
        U, V, b, w = (v.numpy() for v in model.trainable_variables)
        assert w.shape == (D, 1)
        lr, l2r = orc.gmf_forward(U, V, b, w, u, i, lab)[:2]
        assert abs(float(loss) - lr) <= TOL * abs(lr) and abs(float(l2) - l2r) <= TOL * abs(l2r)
        W0 = w.copy()
        for it in range(4):
            u, i = rng.integers(0, NU, B).astype(np.int32), rng.integers(0, NI, B).astype(np.int32)
            lab = (rng.random(B) < 0.4).astype(np.float32)
            loss, l2 = step(u, i, lab)
            lr, l2r = orc.gmf_step(U, V, b, w, u, i, lab, oo)
            assert abs(float(loss) - lr) <= TOL * abs(lr) and abs(float(l2) - l2r) <= TOL * abs(l2r), it
        Ud, Vd, bd, wd = (v.numpy() for v in model.trainable_variables)
        assert rel_err(Ud, U) < TOL and rel_err(Vd, V) < TOL and rel_err(bd, b) < TOL and rel_err(wd, w) < TOL
        assert np.abs(wd - W0).max() > 1e-5                              # the Dense kernel trained
        # inference (gmf.py:36-41): squeeze(mlp(expand_dims(user_vec, 1) * V), -1) + reshape(b): the device GMF scorer
        pred = model.inference(np.arange(8, dtype=np.int32))
    want = (U[:8, None, :] * V[None, :, :]) @ w.reshape(-1) + b.reshape(-1)
    assert pred.shape == (8, NI) and rel_err(np.asarray(pred), want) < 1e-4


def test_loss_only_objective_and_host_uses_of_a_lookup():
    """tape.gradient(loss) alone (no l2 term) is the fused step with no_l2; a lookup used as data is the table's rows"""
    from openrec_amd.tf2.compat import tf, optimizers
    from openrec_amd.tf2.modules import LatentFactor, PairwiseLogLoss
    from oracle import numpy_oracle as orc
    NU, NI, D, B = 300, 400, 16, 512
    Uf, Vf, bf = LatentFactor(NU, D, name="u"), LatentFactor(NI, D, name="v"), LatentFactor(NI, 1, name="b")
    loss_fn, opt = PairwiseLogLoss(), optimizers.SGD(0.1)
    U, V, b = Uf.variables[0].numpy(), Vf.variables[0].numpy(), bf.variables[0].numpy()
    rng = np.random.default_rng(7)
    u, p, n = (rng.integers(0, hi, B).astype(np.int32) for hi in (NU, NI, NI))
    rows = Uf(u)
    assert rows.shape == (B, D) and np.array_equal(np.asarray(rows), U[u]) and np.array_equal(rows[3], U[u[3]])
    assert np.array_equal(rows * 2.0, U[u] * 2.0) and np.array_equal(np.asarray(bf(p.reshape(2, -1))), b[p.reshape(2, -1)])
    assert float(tf.nn.l2_loss(rows)) == pytest.approx(0.5 * float((U[u].astype(np.float64) ** 2).sum()), rel=1e-6)
    with pytest.raises(IndexError):
        np.asarray(Uf(np.array([0, NU], np.int32)))
    vars_ = Uf.variables + Vf.variables + bf.variables
    with tf.GradientTape() as tape:
        loss = loss_fn(Uf(u), Vf(p), Vf(n), bf(p), bf(n))
    opt.apply_gradients(zip(tape.gradient(loss, vars_), vars_))
    gr = orc.bpr_grads(U, V, b, u, p, n)
    lr, _, _ = orc.bpr_forward(U, V, b, u, p, n)
    l2g = {"gu": U[u], "gp": V[p], "gn": V[n]}                             # bpr_grads differentiates loss + l2: take the l2 part out
    oo = orc.SGD(0.1)
    oo.apply(U, u, gr["gu"] - l2g["gu"]); oo.apply(V, np.concatenate([p, n]), np.concatenate([gr["gp"] - l2g["gp"], gr["gn"] - l2g["gn"]]))
    oo.apply(b, np.concatenate([p, n]), np.concatenate([gr["gbp"], gr["gbn"]])[:, None])
    assert abs(float(loss) - lr) <= TOL * abs(lr)
    assert rel_err(Uf.variables[0].numpy(), U) < TOL and rel_err(Vf.variables[0].numpy(), V) < TOL and rel_err(bf.variables[0].numpy(), b) < TOL


def test_unrecognised_composition_warns_outside_a_tape_and_raises_under_one():
    from openrec_amd.tf2.compat import tf
    from openrec_amd.tf2.modules import LatentFactor, PairwiseLogLoss
    Uf, Vf, bf = LatentFactor(50, 8), LatentFactor(60, 8), LatentFactor(60, 1)
    ids = np.arange(10, dtype=np.int32)
    import openrec_amd.tf2.modules._compose as C
    C._warned.clear()
    with pytest.warns(RuntimeWarning, match="WITHOUT gradients"):
        v = PairwiseLogLoss()(Uf(ids), Vf(ids), Vf(ids + 1))                   # no biases: not bpr.py's composition
    U, V = Uf.variables[0].numpy(), Vf.variables[0].numpy()
    x = ((U[ids] * V[ids]).sum(1) - (U[ids] * V[ids + 1]).sum(1)).astype(np.float64)
    assert float(v) == pytest.approx(float(np.mean(np.log1p(np.exp(-x)))), rel=1e-5)
    with tf.GradientTape():                                                    # a tape expects to train through it: refused
        with pytest.raises(NotImplementedError, match="WITHOUT gradients"):
            PairwiseLogLoss()(Uf(ids), Vf(ids), Vf(ids + 1))


@pytest.mark.parametrize("optk", ["sgd", "adagrad"])
def test_pointwise_mse_loss_with_sigmoid_trains_on_the_fused_path(optk):
    """PointwiseMSELoss(sigmoid=True) (modules/pointwise_mse_loss.py:24-25) over the three lookups of wrmf.py:23-25: the same fused
    kernel as WRMF with the prediction through a sigmoid (ORX_POINT_SIGMOID), against the oracle"""
    from openrec_amd.tf2.compat import tf, optimizers
    from openrec_amd.tf2.modules import LatentFactor, PointwiseMSELoss
    from oracle import numpy_oracle as orc
    NU, NI, D, B = 300, 400, 32, 512
    Uf, Vf, bf = LatentFactor(NU, D, name="u"), LatentFactor(NI, D, name="v"), LatentFactor(NI, 1, name="b")
    U, V, b = (f.variables[0].numpy() for f in (Uf, Vf, bf))
    opt, oo = {"sgd": (optimizers.SGD(0.05), orc.SGD(0.05)), "adagrad": (optimizers.Adagrad(0.05), orc.Adagrad(0.05))}[optk]
    vars_ = Uf.variables + Vf.variables + bf.variables
    rng = np.random.default_rng(2)
    loss_mod = PointwiseMSELoss(a=2.0, b=0.5, sigmoid=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for it in range(4):
            u, i = rng.integers(0, NU, B).astype(np.int32), rng.integers(0, NI, B).astype(np.int32)
            u[:7] = 3; i[5:11] = 9                                             # duplicates
            y = (rng.random(B) < 0.4).astype(np.float32)
            with tf.GradientTape() as tape:
                uv, iv = Uf(u), Vf(i)
                loss = loss_mod(uv, iv, bf(i), y)
                l2 = tf.nn.l2_loss(uv) + tf.nn.l2_loss(iv)
            opt.apply_gradients(zip(tape.gradient((loss, l2), vars_), vars_))
            want, _, _ = orc.wrmf_forward(U, V, b, u, i, y, 2.0, 0.5, sigmoid=True)
            gr = orc.wrmf_grads(U, V, b, u, i, y, 2.0, 0.5, sigmoid=True)
            if hasattr(oo, "begin_step"):
                oo.begin_step()
            oo.apply(U, u, gr["gu"], key="U"); oo.apply(V, i, gr["gi"], key="V"); oo.apply(b, i, gr["gb"][:, None], key="b")
            assert abs(float(loss) - float(want)) <= TOL * abs(float(want)), it
    assert rel_err(Uf.variables[0].numpy(), U) < TOL and rel_err(Vf.variables[0].numpy(), V) < TOL and rel_err(bf.variables[0].numpy(), b) < TOL


def test_a_lookup_made_before_a_step_holds_the_pre_step_rows_and_a_scaled_l2_term_is_refused():
    """TensorFlow gathers at call time: `v = lf(ids); train_step(...); np.asarray(v)` gives the rows as they were at the lookup
    (the lazy lookup is snapshotted when a step on its table is recorded).  `l2_reg * tf.nn.l2_loss(vec)` evaluates, but a tape over
    it is refused with a message that names the supported form (the fused step takes the l2 term with weight 1 or not at all)."""
    from openrec_amd.tf2.compat import tf, optimizers
    from openrec_amd.tf2.modules import LatentFactor, PairwiseLogLoss
    NU, NI, D, B = 200, 300, 16, 256
    Uf, Vf, bf = LatentFactor(NU, D, name="u"), LatentFactor(NI, D, name="v"), LatentFactor(NI, 1, name="b")
    U0 = Uf.variables[0].numpy()
    rng = np.random.default_rng(3)
    u, p, n = (rng.integers(0, hi, B).astype(np.int32) for hi in (NU, NI, NI))
    probe = np.arange(40, dtype=np.int32)
    v = Uf(probe)                                                          # lazy: nothing gathered yet
    vars_ = Uf.variables + Vf.variables + bf.variables
    opt = optimizers.SGD(0.5)
    with tf.GradientTape() as tape:
        uv, pv, nv = Uf(u), Vf(p), Vf(n)
        loss = PairwiseLogLoss()(uv, pv, nv, bf(p), bf(n))
        l2 = tf.nn.l2_loss(uv) + tf.nn.l2_loss(pv) + tf.nn.l2_loss(nv)
    opt.apply_gradients(zip(tape.gradient((loss, l2), vars_), vars_))
    after = Uf.variables[0].numpy()
    assert np.abs(after[probe] - U0[probe]).max() > 1e-4                   # the step moved some of the probed rows ...
    assert np.array_equal(np.asarray(v), U0[probe])                        # ... and the earlier lookup still holds the old ones
    with tf.GradientTape() as tape:
        uv, pv, nv = Uf(u), Vf(p), Vf(n)
        loss = PairwiseLogLoss()(uv, pv, nv, bf(p), bf(n))
        l2 = 0.01 * (tf.nn.l2_loss(uv) + tf.nn.l2_loss(pv) + tf.nn.l2_loss(nv))
    U1, V1 = Uf.variables[0].numpy(), Vf.variables[0].numpy()
    want = 0.01 * 0.5 * float((U1[u].astype(np.float64) ** 2).sum() + (V1[p].astype(np.float64) ** 2).sum() + (V1[n].astype(np.float64) ** 2).sum())
    assert float(l2) == pytest.approx(want, rel=1e-5)
    with pytest.raises(NotImplementedError, match="scaled by 0.01"):
        tape.gradient((loss, l2), vars_)
    # `k * l2(a) + k * l2(b) + k * l2(c)`: the sum keeps the common weight (value k * sum, and the tape still refuses it) ...
    with tf.GradientTape() as tape:
        uv, pv, nv = Uf(u), Vf(p), Vf(n)
        loss = PairwiseLogLoss()(uv, pv, nv, bf(p), bf(n))
        l2 = 0.01 * tf.nn.l2_loss(uv) + 0.01 * tf.nn.l2_loss(pv) + 0.01 * tf.nn.l2_loss(nv)
    assert float(l2) == pytest.approx(want, rel=1e-5)
    with pytest.raises(NotImplementedError, match="scaled by 0.01"):
        tape.gradient((loss, l2), vars_)
    # ... and terms with different weights evaluate to their weighted sum and never resolve to a fused step
    mixed = 0.5 * tf.nn.l2_loss(uv) + 0.25 * tf.nn.l2_loss(pv)
    want_mixed = 0.5 * 0.5 * float((U1[u].astype(np.float64) ** 2).sum()) + 0.25 * 0.5 * float((V1[p].astype(np.float64) ** 2).sum())
    assert float(mixed) == pytest.approx(want_mixed, rel=1e-5)
    assert mixed.resolve() is None


@pytest.mark.parametrize("optk,loss_func,compat", [("sgd", "mse", True), ("sgd", "bce", False), ("adagrad", "mse", False), ("adam", "bce", False)])
def test_reference_dlrm_class_text_trains_on_the_fused_path(optk, loss_func, compat):
    """recommenders/dlrm.py of the reference, executed as it stands: its `call` composes LatentFactor lookups, the bottom MLP,
    SecondOrderFeatureInteraction, tf.concat, the top MLP, [tf.clip_by_value,] tf.reshape and a keras loss object
    (dlrm.py:76-100, :63-74).  The loss object recognises the tree and the model trains through `orx_dlrm_step` -- the modules'
    tables become that step's parameters -- and gives the oracle's parameters, not a host forward without gradients."""
    from openrec_amd.tf2.compat import optimizers
    from openrec_amd.tf2.modules import SecondOrderFeatureInteraction
    from openrec_amd.tf2.modules._compose import _models, _RowRangeTable
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    from dlrm_util import draw_batch
    DLRM = _ref_class("dlrm.py", "DLRM")
    cfg = dict(m_spa=16, ln_emb=[50, 7, 300, 3], ln_bot=[32, 16], ln_top=[64, 32, 1])
    model = DLRM(loss_func=loss_func, loss_threshold=0.0 if loss_func == "mse" else 0.01, **cfg)
    if not compat:      # (the module's own switch: the evidently intended interaction instead of the reference's all-zero one)
        model._dot_interaction = SecondOrderFeatureInteraction(self_interaction=False, reference_compat=False)
    rng = np.random.default_rng(11)
    B = 256
    d0 = np.log1p(rng.integers(0, 100, (B, 13))).astype(np.float32)
    s0 = np.stack([rng.integers(0, n, B) for n in cfg["ln_emb"]], 1).astype(np.int32)
    y0 = (rng.random(B) < 0.3).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        first = float(model(d0, s0, y0))                                  # forward only: the composition is recognised and bound
        assert len(model._latent_factors[0]._composed) == 1
        assert all(isinstance(f.table, _RowRangeTable) for f in model._latent_factors)
        names = [v.name for v in model.trainable_variables]
        assert len(names) == 4 + 2 * (2 + 3)                              # four embedding tables, kernel + bias of five Dense layers
        # the oracle starts from the model's own parameters
        o = DLRMOracle(dtype=np.float64, seed=1, reference_compat=compat, loss_func=loss_func,
                       loss_threshold=0.0 if loss_func == "mse" else 0.01, dense_dim=13, **cfg)
        for f, lf in enumerate(model._latent_factors):
            o.emb[f] = lf.variables[0].numpy().astype(np.float64)
        for layers, mlp in ((o.bot, model._mlp_bot), (o.top, model._mlp_top)):
            for l, layer in enumerate(mlp.layers):
                layers[l][0] = layer.kernel.read().astype(np.float64); layers[l][1] = layer.bias.read().reshape(-1).astype(np.float64)
        assert first == pytest.approx(float(o.loss_and_grads(d0, s0, y0)[0]), rel=1e-5)
        opt, oo = {"sgd": (optimizers.SGD(0.05), orc.SGD(0.05)), "adagrad": (optimizers.Adagrad(0.05), orc.Adagrad(0.05)),
                   "adam": (optimizers.Adam(0.002), orc.AdamTFSparse(0.002))}[optk]
        step = _train_step(model, opt)
        got, want = [], []
        for it in range(4):
            de, sp, la = draw_batch(o, rng, B, cfg["ln_emb"])              # (away from the network's relu ties: tests/dlrm_util.py)
            got.append(step(de, sp, la))
            want.append(float(o.step(de, sp, la, oo)))
        pred = model.inference(d0, s0)                                    # dlrm.py:76-100 again, outside a tape: orx_dlrm_inference
        assert np.asarray(pred).shape == (B,)
    assert len(model._latent_factors[0]._composed) == 1                   # one fused model for all of it
    tol = TOL_ADAM if optk == "adam" else 2e-5
    assert np.allclose([float(g) for g in got], want, rtol=2e-5)
    for f, lf in enumerate(model._latent_factors):
        assert rel_err(lf.variables[0].numpy(), o.emb[f]) < tol, f
    for layers, mlp in ((o.bot, model._mlp_bot), (o.top, model._mlp_top)):
        for l, layer in enumerate(mlp.layers):
            assert rel_err(layer.kernel.read(), layers[l][0]) < tol and rel_err(layer.bias.read().reshape(-1), layers[l][1]) < tol, l
    assert np.allclose(np.asarray(pred), np.asarray(o.forward(d0, s0)["pred"]).reshape(-1), rtol=1e-4, atol=1e-6)


def test_an_unrecognised_composition_under_a_tape_raises_instead_of_training_nothing():
    """A tree of modules that is none of the reference's compositions has no device path; under a GradientTape its host
    evaluation raises (it has no gradients) -- outside a tape it is just the values."""
    from openrec_amd.tf2.compat import tf
    from openrec_amd.tf2.modules import LatentFactor, MLP
    lf = LatentFactor(40, 8, name="x")
    mlp = MLP([4, 1])
    ids = np.arange(10, dtype=np.int32)
    out = mlp(lf(ids) * lf(ids))                                          # (not gmf.py:28: two layers, bias)
    assert np.asarray(out).shape == (10, 1)                               # values, on the host
    with tf.GradientTape():
        with pytest.raises(NotImplementedError, match="not one of the compositions"):
            float(tf.reduce_sum(mlp(lf(ids) * lf(ids))))
