"""The modules that exist for API parity only (openrec/tf2/modules/second_order_feature_interaction.py:4-34; MLP
multi_layer_perceptron.py:5-18 needs device tables and is covered on the GPU) compute on the host: values as in the reference,
and a RuntimeWarning under a GradientTape, where a caller would expect to train through them."""
import warnings

import numpy as np
import pytest


def test_second_order_interaction_values_and_tape_warning():
    from openrec_amd.tf2.modules import SecondOrderFeatureInteraction
    from openrec_amd.tf2.modules import _compose
    from openrec_amd.tf2._lazy import GradientTape
    rng = np.random.default_rng(0)
    xs = [rng.normal(size=(5, 4)).astype(np.float32) for _ in range(3)]
    z = np.stack(xs, 1)
    want = np.stack([(z[:, 1] * z[:, 0]).sum(1), (z[:, 2] * z[:, 0]).sum(1), (z[:, 2] * z[:, 1]).sum(1)], 1)
    got = SecondOrderFeatureInteraction(reference_compat=False)(xs)
    assert np.allclose(got, want, rtol=1e-6)
    # the reference's own output: lower triangle kept, strictly-upper selected -> zeros
    assert not SecondOrderFeatureInteraction()(xs).any()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        SecondOrderFeatureInteraction()(xs)                          # no tape: silent
    _compose._warned.clear()
    with pytest.warns(RuntimeWarning, match="WITHOUT gradients"):
        with GradientTape():
            SecondOrderFeatureInteraction()(xs)
