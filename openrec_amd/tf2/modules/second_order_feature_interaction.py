"""SecondOrderFeatureInteraction
(openrec/tf2/modules/second_order_feature_interaction.py:4-34), host arrays.

`reference_compat=True` reproduces the reference bit for bit, INCLUDING its
bug: the lower triangle of Z Z^T is kept (line 21) but the strictly-upper
triangle is selected (lines 23-27, 32), so every selected element is 0 (with
`self_interaction` the diagonal survives).  `reference_compat=False` gives the
evidently intended strictly-lower-triangle pairwise dot products."""
import numpy as np


class SecondOrderFeatureInteraction:

    def __init__(self, self_interaction=False, reference_compat=True):
        self._self_interaction = self_interaction
        self._reference_compat = reference_compat

    def __call__(self, inputs):
        from .._lazy import active_tape
        if active_tape() is not None:          # (the device path of this op lives inside the packaged DLRM's step)
            from ._compose import host_fallback
            host_fallback("SecondOrderFeatureInteraction")
        z = np.stack([np.asarray(x, np.float32) for x in inputs], axis=1)          # [B, F, d]
        dots = np.einsum('bfd,bgd->bfg', z, z)
        F = z.shape[1]
        if self._reference_compat:
            dots = np.tril(dots)                                                   # line 21
            mask = np.triu(np.ones((F, F), bool), k=0 if self._self_interaction else 1)   # lines 23-27
        else:
            mask = np.tril(np.ones((F, F), bool), k=0 if self._self_interaction else -1)
        return dots[:, mask]
