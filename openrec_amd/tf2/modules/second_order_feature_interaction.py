"""SecondOrderFeatureInteraction
(openrec/tf2/modules/second_order_feature_interaction.py:4-34).  On looked-up rows /
MLP outputs it is a node of the DLRM composition (device: `orx_dlrm_step`); on
plain arrays it computes on the host.

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

    def host_forward(self, inputs):
        z = np.stack([np.asarray(x, np.float32) for x in inputs], axis=1)          # [B, F, d]
        dots = np.einsum('bfd,bgd->bfg', z, z)
        F = z.shape[1]
        if self._reference_compat:
            dots = np.tril(dots)                                                   # line 21
            mask = np.triu(np.ones((F, F), bool), k=0 if self._self_interaction else 1)   # lines 23-27
        else:
            mask = np.tril(np.ones((F, F), bool), k=0 if self._self_interaction else -1)
        return dots[:, mask]

    def __call__(self, inputs):
        """A node of a lazy expression when any input is lazy (looked-up rows, an MLP output): inside the composition of
        dlrm.py:87-93 it runs in `orx_dlrm_step` (interact_fwd / interact_bwd kernels); plain arrays compute on the host."""
        from ._expr import Expr, is_lazy
        inputs = list(inputs)
        if any(is_lazy(x) for x in inputs):
            return Expr("interact", self, *inputs)
        return self.host_forward(inputs)
