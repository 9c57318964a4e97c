"""The host-only pieces of the TensorFlow shim the reference's example scripts touch (SURVEY.md Appendix C;
tf2_examples/bpr_citeulike.py:1-67, dlrm_criteo.py:1-60): no device needed -- the train step itself is covered on the GPU
(tests/test_gpu_reference_examples.py runs both scripts textually)."""
import sys

import numpy as np
import pytest


def test_install_registers_the_aliases_only_where_nothing_real_exists():
    from openrec_amd.tf2 import compat
    had_tf = "tensorflow" in sys.modules
    installed = compat.install()
    import openrec.tf2.recommenders as rec
    import openrec.tf2.modules as mods
    import openrec_amd.tf2.recommenders as ours
    assert rec.BPR is ours.BPR and rec.DLRM is ours.DLRM and hasattr(mods, "LatentFactor") and hasattr(mods, "PairwiseLogLoss")
    from openrec.tf2.data import Dataset          # noqa: F401  (bpr_citeulike.py:4)
    from openrec.tf2.metrics import AUC, NDCG, Recall, DictMean   # noqa: F401  (bpr_citeulike.py:5)
    if installed:
        import tensorflow as tf
        from tensorflow.keras import optimizers, Model          # noqa: F401
        from tensorflow.data import Dataset as TFDataset        # noqa: F401
        assert tf.keras.optimizers.Adam is optimizers.Adam and callable(tf.function) and tf.GradientTape is compat.GradientTape
        assert compat.install() is True or "tensorflow" in sys.modules       # idempotent
    else:
        assert had_tf or "tensorflow" in sys.modules


def test_tensor_slice_dataset_batches_prefetches_and_shuffles_like_the_script_expects():
    from openrec_amd.tf2.compat import TensorSliceDataset
    n = 23
    data = {"dense_features": np.arange(n * 3, dtype=np.float32).reshape(n, 3), "label": np.arange(n, dtype=np.float32)}
    ds = TensorSliceDataset.from_tensor_slices(data).batch(5).prefetch(8)            # dlrm_criteo.py:18-22
    batches = list(ds)
    assert [len(b["label"]) for b in batches] == [5, 5, 5, 5, 3]                      # the last batch is partial, as in TF
    assert np.array_equal(np.concatenate([b["label"] for b in batches]), data["label"])
    assert np.array_equal(batches[1]["dense_features"], data["dense_features"][5:10])
    assert [len(b["label"]) for b in TensorSliceDataset.from_tensor_slices(data).batch(5, drop_remainder=True)] == [5] * 4
    sh = TensorSliceDataset.from_tensor_slices(data).batch(5).shuffle(3, seed=7)      # batches through a 3-element buffer
    e1, e2 = [b["label"][0] for b in sh], [b["label"][0] for b in sh]
    assert sorted(e1) == [0.0, 5.0, 10.0, 15.0, 20.0] and sorted(e2) == sorted(e1)    # a permutation of the batches, every epoch
    assert e1[0] in (0.0, 5.0, 10.0)                                                  # only what was in the buffer can come first
    with pytest.raises(ValueError):
        TensorSliceDataset.from_tensor_slices({"a": np.zeros(3), "b": np.zeros(4)})
    rows = list(TensorSliceDataset.from_tensor_slices(data))                          # unbatched: one record per element
    assert len(rows) == n and rows[4]["label"] == 4.0


def test_keras_auc_matches_the_exact_auc_to_its_discretisation():
    from sklearn.metrics import roc_auc_score
    from openrec_amd.tf2.compat import AUC
    rng = np.random.default_rng(1)
    y = (rng.random(20000) < 0.3).astype(np.float32)
    p = np.clip(0.35 * y + 0.65 * rng.random(20000), 0, 1).astype(np.float32)
    m = AUC()
    for lo in range(0, 20000, 4096):                                                  # accumulated over batches (dlrm_criteo.py:50-53)
        m.update_state(y[lo:lo + 4096], p[lo:lo + 4096])
    got = float(m.result().numpy())
    assert abs(got - roc_auc_score(y, p)) < 2e-3                                      # 200 thresholds, trapezoids
    m.reset_states()
    m.update_state(y, 1.0 - p)
    assert abs(float(m.result()) - (1.0 - roc_auc_score(y, p))) < 2e-3
    m.reset_states()
    m.update_state(np.array([0, 0, 1, 1]), np.array([0.1, 0.2, 0.8, 0.9]))
    assert float(m.result()) == pytest.approx(1.0, abs=1e-6)


def test_keras_mean_averages_the_loss_tuple_like_the_script():
    from openrec_amd.tf2.compat import Mean
    m = Mean()
    m.update_state((np.float32(2.0), np.float32(4.0)))            # Mean.update_state((loss, l2)) averages both (bpr_citeulike.py:54)
    m.update_state(np.float32(6.0))
    assert float(m.result().numpy()) == pytest.approx(4.0)
    m.reset_states()
    assert float(m.result()) == 0.0


def test_keras_model_collects_variables_in_assignment_order():
    from openrec_amd.tf2.compat import Model

    class Var:
        def __init__(self, name):
            self.name = name

    class Layer:
        def __init__(self, *names):
            self.variables = [Var(n) for n in names]

    class Net(Model):
        def __init__(self):
            super().__init__()
            self.a = Layer("a0")
            self.loss = object()                                   # (a loss module: no variables)
            self.b = Layer("b0", "b1")
            self.again = self.a                                    # the same layer twice: its variables once

        def call(self, x, scale=1):
            return x * scale

    net = Net()
    assert [v.name for v in net.trainable_variables] == ["a0", "b0", "b1"]
    assert net(3, scale=2, training=True) == 6                     # __call__ -> call; keras' `training` kwarg is dropped


def test_host_fallbacks_of_the_tensor_ops():
    from openrec_amd.tf2 import compat
    tf = compat.tf
    a, b = np.arange(6, dtype=np.float32).reshape(2, 3), np.arange(12, dtype=np.float32).reshape(4, 3)
    assert np.array_equal(tf.linalg.matmul(a, b, transpose_b=True), a @ b.T)
    assert np.array_equal(tf.reshape(b, [-1]), b.reshape(-1))
    assert float(tf.nn.l2_loss(a)) == pytest.approx(0.5 * float((a ** 2).sum()))
    assert tf.constant([1, 2], dtype=tf.int32).dtype == np.int32
    f = tf.function(lambda x: x + 1)
    assert f(1) == 2 and tf.function()(lambda x: x)(5) == 5
