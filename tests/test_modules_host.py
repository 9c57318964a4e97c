"""SecondOrderFeatureInteraction (openrec/tf2/modules/second_order_feature_interaction.py:4-34) on PLAIN arrays computes on the
host, values as in the reference (with anything lazy among its inputs -- looked-up rows, an MLP output -- it is a node of the DLRM
composition, covered on the GPU: tests/test_gpu_compose.py); a lazy tree that is none of the reference's compositions raises under a
GradientTape, where a caller would expect to train through it."""
import warnings

import numpy as np
import pytest


def test_second_order_interaction_values_and_tape_refusal():
    from openrec_amd.tf2.modules import SecondOrderFeatureInteraction
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
    with GradientTape():
        assert np.allclose(SecondOrderFeatureInteraction(reference_compat=False)(xs), want, rtol=1e-6)     # plain arrays: plain values
        from openrec_amd.tf2.modules._expr import Expr
        lazy = SecondOrderFeatureInteraction()([Expr("neg", xs[0]), xs[1], xs[2]])                         # a lazy input: a tree node ...
        assert isinstance(lazy, Expr) and lazy.op == "interact"
        with pytest.raises(NotImplementedError, match="not one of the compositions"):                     # ... with no device path
            np.asarray(lazy)
    assert np.allclose(np.asarray(SecondOrderFeatureInteraction(reference_compat=False)([Expr("neg", xs[0]), xs[1], xs[2]])),
                       SecondOrderFeatureInteraction(reference_compat=False)([-xs[0], xs[1], xs[2]]), rtol=1e-6)      # outside a tape: its values
