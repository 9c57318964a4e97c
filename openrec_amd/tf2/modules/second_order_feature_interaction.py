"""SecondOrderFeatureInteraction
(openrec/tf2/modules/second_order_feature_interaction.py:4-34).  On looked-up rows /
MLP outputs it is a node of the DLRM composition (device: `orx_dlrm_step`); on
plain arrays it runs `orx_interact_forward` (the same kernels, values only).

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

    def device_forward(self, inputs, ctx=None):
        """the module on plain arrays, ON THE DEVICE: `orx_interact_forward` (the interaction kernels of the DLRM step)"""
        from ... import _ffi
        from ... import runtime as rt
        z = np.ascontiguousarray(np.stack([np.asarray(x, np.float32) for x in inputs], axis=1))          # [B, F, d] (:19)
        B, F, d = z.shape
        P = F * (F + 1) // 2 if self._self_interaction else F * (F - 1) // 2
        out = np.empty((B, P), np.float32)
        ctx = ctx or rt.default_context()
        if B and P:
            _ffi.check(ctx._lib.orx_interact_forward(ctx._h, z.ctypes.data, B, F, d, int(bool(self._self_interaction)),
                                                     int(bool(self._reference_compat)), 0, out.ctypes.data))
        return out

    def __call__(self, inputs):
        """A node of a lazy expression when any input is lazy (looked-up rows, an MLP output): inside the composition of
        dlrm.py:87-93 it runs in `orx_dlrm_step` (interact_fwd / interact_bwd kernels); plain arrays go through `orx_interact_forward`."""
        from ._expr import Expr, is_lazy
        inputs = list(inputs)
        if any(is_lazy(x) for x in inputs):
            return Expr("interact", self, *inputs)
        return self.device_forward(inputs)
