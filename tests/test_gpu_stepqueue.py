"""The drop-in training loop (one `train_step` per batch, tf2_examples/bpr_citeulike.py:33-39) queues its
steps and runs them as K-step device calls; nothing observable may change: losses, tables, metrics and
interleaved reads are compared with the oracle stepped one batch at a time."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _loop(model_name, n_steps, B, NU, NI, D, read_at=(), device_ids=False, censor=False, seed=0):
    from openrec_amd.tf2 import compat as tf
    from openrec_amd.tf2 import recommenders as R
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(seed)
    m = {"bpr": R.BPR, "ucml": R.UCML}[model_name](dim_user_embed=D, dim_item_embed=D, total_users=NU, total_items=NI)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    if censor:
        U *= 30; V *= 30
    m.user_latent_factor.variables[0].assign(U); m.item_latent_factor.variables[0].assign(V); m.item_bias.variables[0].assign(b)
    optimizer = tf.keras.optimizers.SGD(0.02)
    mean = tf.keras.metrics.Mean()
    oo = orc.SGD(0.02)
    got, want = [], []
    for s in range(n_steps):
        u = rng.integers(0, NU, B).astype(np.int32); p = rng.integers(0, NI, B).astype(np.int32); n = rng.integers(0, NI, B).astype(np.int32)
        ids = (u, p, n)
        if device_ids:
            import torch
            ids = tuple(torch.from_numpy(x).cuda() for x in ids)
        with tf.GradientTape() as tape:
            loss, l2 = m(*ids)
        grads = tape.gradient((loss, l2), m.trainable_variables)
        optimizer.apply_gradients(zip(grads, m.trainable_variables))
        if censor:
            m.censor_vec(*ids)
        mean.update_state(loss)
        got.append(loss)
        if model_name == "bpr":
            lw, _ = orc.bpr_step(U, V, b, u, p, n, oo)
        else:
            lw, _ = orc.ucml_step(U, V, b, u, p, n, oo, margin=0.5, do_censor=censor)
        want.append(float(lw))
        if s in read_at:                                          # observing the model mid-way flushes the queue
            assert rel_err(m.user_latent_factor.variables[0].numpy(), U) < 2e-5
            assert abs(float(loss) - want[-1]) <= 2e-5 * abs(want[-1])
    assert abs(float(mean.result()) - np.mean(want)) <= 2e-5 * abs(np.mean(want))
    for g, w in zip(got, want):
        assert abs(float(g) - w) <= 2e-5 * abs(w)
    assert rel_err(m.user_latent_factor.variables[0].numpy(), U) < 2e-5
    assert rel_err(m.item_latent_factor.variables[0].numpy(), V) < 2e-5
    assert rel_err(m.item_bias.variables[0].numpy(), b) < 2e-5
    return m


def test_queued_loop_equals_stepwise_oracle():
    m = _loop("bpr", 75, 512, 700, 900, 64)                       # 2 full queues + a partial one
    assert not m._queue.steps


def test_reads_in_the_middle_flush_the_queue():
    _loop("bpr", 40, 256, 300, 400, 32, read_at=(0, 7, 8, 33))


def test_device_ids_and_ucml_censor():
    _loop("bpr", 37, 512, 700, 900, 64, device_ids=True)
    _loop("ucml", 37, 512, 700, 900, 64, censor=True)             # censor_vec folds into the queued steps


def test_queue_is_really_used(monkeypatch):
    from openrec_amd import runtime as rt
    calls = []
    real = rt.pairwise_step
    monkeypatch.setattr(rt, "pairwise_step", lambda *a, **k: (calls.append(k.get("K")), real(*a, **k))[1])
    _loop("bpr", 70, 256, 300, 400, 32)
    assert calls.count(32) == 2 and sum(calls) == 70 and len(calls) <= 4


def test_pointwise_loop_is_queued():
    from openrec_amd.tf2 import compat as tf
    from openrec_amd.tf2 import recommenders as R
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(2)
    NU, NI, D, B = 500, 600, 32, 384
    m = R.WRMF(dim_user_embed=D, dim_item_embed=D, total_users=NU, total_items=NI, a=2.0, b=0.5)
    U = m.user_latent_factor.variables[0].numpy(); V = m.item_latent_factor.variables[0].numpy(); b = m.item_bias.variables[0].numpy()
    optimizer = tf.keras.optimizers.SGD(0.01)
    oo = orc.SGD(0.01)
    got, want = [], []
    for s in range(45):
        u = rng.integers(0, NU, B).astype(np.int32); i = rng.integers(0, NI, B).astype(np.int32); y = (rng.random(B) < 0.4).astype(np.float32)
        with tf.GradientTape() as tape:
            loss, l2 = m(u, i, y)
        optimizer.apply_gradients(zip(tape.gradient((loss, l2), m.trainable_variables), m.trainable_variables))
        got.append(loss)
        want.append(float(orc.wrmf_step(U, V, b, u, i, y, oo, a=2.0, b_w=0.5)[0]))
    for g, w in zip(got, want):
        assert abs(float(g) - w) <= 2e-5 * abs(w)
    assert rel_err(m.user_latent_factor.variables[0].numpy(), U) < 2e-5 and rel_err(m.item_latent_factor.variables[0].numpy(), V) < 2e-5


def test_dlrm_loop_is_queued(monkeypatch):
    from openrec_amd import runtime as rt
    from openrec_amd.tf2 import compat as tf
    from openrec_amd.tf2.recommenders import DLRM
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle
    calls = []
    real = rt.DLRMModel.step
    monkeypatch.setattr(rt.DLRMModel, "step", lambda self, *a, **k: (calls.append(k.get("K", 1)), real(self, *a, **k))[1])
    rng = np.random.default_rng(6)
    counts = [40, 7, 300, 3, 90]
    cfg = dict(m_spa=4, ln_emb=counts, ln_bot=[8, 4], ln_top=[16, 8, 1])
    o = DLRMOracle(dtype=np.float32, seed=9, reference_compat=False, dense_dim=13, **cfg)
    m = DLRM(reference_compat=False, **cfg)
    tv = m.trainable_variables
    tv[0].assign(np.concatenate(o.emb))
    k = 1
    for layers in (o.bot, o.top):
        for W, b in layers:
            tv[k].assign(W); tv[k + 1].assign(b.reshape(1, -1)); k += 2
    optimizer = tf.keras.optimizers.SGD(0.05)
    oo = orc.SGD(0.05)
    got, want = [], []
    for s in range(70):
        dense = rng.normal(size=(64, 13)).astype(np.float32)
        sparse = np.stack([rng.integers(0, n, 64) for n in counts], 1).astype(np.int32)
        label = (rng.random(64) < 0.4).astype(np.float32)
        with tf.GradientTape() as tape:
            loss = m(dense, sparse, label)
        optimizer.apply_gradients(zip(tape.gradient(loss, m.trainable_variables), m.trainable_variables))
        got.append(loss)
        want.append(float(o.step(dense, sparse, label, oo)))
    pred = m.inference(dense, sparse)                        # observing the model runs the rest of the queue
    assert calls.count(32) == 2 and sum(calls) == 70
    for g, w in zip(got, want):
        assert abs(float(g) - w) <= 2e-5 * abs(w)
    assert rel_err(pred, o.inference(dense, sparse)) < 2e-5
    assert rel_err(m.trainable_variables[0].numpy(), np.concatenate(o.emb)) < 2e-5


@pytest.mark.parametrize("optname", ["sgd", "adam"])
def test_learning_rate_assigned_mid_run_applies_from_the_next_step(optname):
    """Keras scripts schedule the rate by assigning `optimizer.learning_rate`.  Steps already applied (still queued or
    not) keep their rate; `optimizer.iterations` counts the applied steps, queued ones included."""
    from openrec_amd.tf2 import compat as tf
    from openrec_amd.tf2 import recommenders as R
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(3)
    NU, NI, D, B, n_steps = 600, 800, 32, 256, 50
    m = R.BPR(dim_user_embed=D, dim_item_embed=D, total_users=NU, total_items=NI)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    m.user_latent_factor.variables[0].assign(U); m.item_latent_factor.variables[0].assign(V); m.item_bias.variables[0].assign(b)
    lr0 = 0.02 if optname == "sgd" else 0.002
    optimizer = tf.keras.optimizers.SGD(lr0) if optname == "sgd" else tf.keras.optimizers.Adam(lr0)
    oo = orc.SGD(lr0) if optname == "sgd" else orc.AdamTFSparse(lr0)
    losses, want = [], []
    for s in range(n_steps):
        if s == 20:                                   # 20 steps sit in the queue at this point
            optimizer.learning_rate = lr0 / 4
            oo.lr = lr0 / 4
            assert optimizer.learning_rate == lr0 / 4 and optimizer.lr == lr0 / 4
        if s == 37:
            assert optimizer.iterations == 37
        u = rng.integers(0, NU, B).astype(np.int32); p = rng.integers(0, NI, B).astype(np.int32); n = rng.integers(0, NI, B).astype(np.int32)
        with tf.GradientTape() as tape:
            loss, l2 = m(u, p, n)
        grads = tape.gradient((loss, l2), m.trainable_variables)
        optimizer.apply_gradients(zip(grads, m.trainable_variables))
        losses.append(loss)
        want.append(float(orc.bpr_step(U, V, b, u, p, n, oo)[0]))
    tol = 5e-5 if optname == "adam" else 2e-5
    for g, w in zip(losses, want):
        assert abs(float(g) - w) <= tol * abs(w)
    assert rel_err(m.user_latent_factor.variables[0].numpy(), U) < tol and rel_err(m.item_latent_factor.variables[0].numpy(), V) < tol
    assert optimizer.iterations == n_steps
