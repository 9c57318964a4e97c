"""GPU: the DLRM modules called on PLAIN ARRAYS run on the device: `MLP(...)(x)` ->
orx_mlp_forward, `SecondOrderFeatureInteraction()(list of arrays)` -> orx_interact_forward, held to a NumPy restatement of
openrec/tf2/modules/multi_layer_perceptron.py:5-18 and second_order_feature_interaction.py:12-34 (the reference's triangle
quirk included).  Tolerance 1e-5 relative (fp32 products)."""
import numpy as np
import pytest

from conftest import rel_err, TOL

pytestmark = pytest.mark.gpu


def _mlp_np(x, layers):
    for W, b, act in layers:
        x = x.astype(np.float64) @ W.astype(np.float64)
        if b is not None:
            x = x + b.astype(np.float64)
        if act == "relu":
            x = np.maximum(x, 0)
        elif act == "sigmoid":
            x = 1.0 / (1.0 + np.exp(-x))
    return x


@pytest.mark.parametrize("units,use_bias,act,out_act,in_dim,B", [([512, 256, 128], True, "relu", "relu", 13, 300),
                                                                 ([64, 1], True, "relu", "sigmoid", 479, 1000),
                                                                 ([1], False, "relu", None, 64, 77),
                                                                 ([8, 4], True, "relu", "relu", 13, 5)])
def test_mlp_on_a_plain_array_runs_on_the_device(units, use_bias, act, out_act, in_dim, B):
    from openrec_amd.tf2.modules import MLP
    rng = np.random.default_rng(1)
    x = rng.normal(size=(B, in_dim)).astype(np.float32)
    mlp = MLP(units_list=units, use_bias=use_bias, activation=act, out_activation=out_act)
    y = np.asarray(mlp(x))                                   # an `mlp` node; looking at it evaluates: orx_mlp_forward
    layers = [(l.kernel.read(), l.bias.read() if l.bias is not None else None, l.activation) for l in mlp.layers]
    want = _mlp_np(x, layers)
    assert y.shape == want.shape and y.dtype == np.float32
    assert rel_err(y, want) < TOL
    assert not hasattr(mlp, "host_forward")


def _interact_np(inputs, itself, compat):
    z = np.stack([np.asarray(x, np.float64) for x in inputs], axis=1)
    dots = np.einsum("bfd,bgd->bfg", z, z)
    F = z.shape[1]
    if compat:
        dots = np.tril(dots)                                                    # :21 LinearOperatorLowerTriangular(...).to_dense()
        mask = np.triu(np.ones((F, F), bool), k=0 if itself else 1)              # :23-27 band_part(ones, 0, -1) [- band_part(ones, 0, 0)]
    else:
        mask = np.tril(np.ones((F, F), bool), k=0 if itself else -1)
    return dots[:, mask]


@pytest.mark.parametrize("F,d,B", [(27, 128, 100), (4, 4, 33), (5, 32, 64), (3, 50, 7)])
@pytest.mark.parametrize("itself", [False, True])
@pytest.mark.parametrize("compat", [True, False])
def test_interaction_on_plain_arrays_runs_on_the_device(F, d, B, itself, compat):
    from openrec_amd.tf2.modules import SecondOrderFeatureInteraction
    rng = np.random.default_rng(2)
    inputs = [rng.normal(size=(B, d)).astype(np.float32) for _ in range(F)]
    m = SecondOrderFeatureInteraction(self_interaction=itself, reference_compat=compat)
    got = m(inputs)
    want = _interact_np(inputs, itself, compat)
    assert got.shape == want.shape
    if want.size:
        assert np.abs(got - want).max() <= TOL * max(np.abs(want).max(), 1.0)
    assert not hasattr(m, "host_forward")
