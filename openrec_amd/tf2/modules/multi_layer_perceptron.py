"""MLP factory (openrec/tf2/modules/multi_layer_perceptron.py:5-18): a stack of
Dense layers.  Only what the hot path needs is device-backed: GMF uses
`MLP(units_list=[1], use_bias=False)`, a single [D, 1] kernel that the fused
pointwise kernel reads and updates in HBM."""
from __future__ import annotations

import numpy as np

from ... import runtime as rt
from .latent_factor import Variable


class Dense:
    def __init__(self, units, activation=None, use_bias=True):
        self.units, self.activation, self.use_bias = units, activation, use_bias
        self.kernel = None          # runtime.Table [in, units], built on first use
        self.bias = None

    def build(self, in_dim, ctx=None, seed=0):
        if self.kernel is None:
            limit = float(np.sqrt(6.0 / (in_dim + self.units)))      # Keras glorot_uniform
            self.kernel = rt.Table(in_dim, self.units, ctx).init_uniform(-limit, limit, 7919 + seed)
            if self.use_bias:
                self.bias = rt.Table(1, self.units, ctx).fill(0.0)
        return self


class MLP:
    def __init__(self, units_list, use_bias=True, activation='relu', out_activation=None):
        self.layers = [Dense(u, activation, use_bias) for u in units_list[:-1]]
        self.layers.append(Dense(units_list[-1], out_activation, use_bias))

    def build(self, in_dim, ctx=None):
        d = in_dim
        for k, layer in enumerate(self.layers):
            layer.build(d, ctx, seed=k)
            d = layer.units
        return self

    @property
    def trainable_variables(self):
        out = []
        for k, layer in enumerate(self.layers):
            if layer.kernel is not None:
                out.append(Variable(layer.kernel, f"dense_{k}/kernel"))
                if layer.bias is not None:
                    out.append(Variable(layer.bias, f"dense_{k}/bias"))
        return out

    def __call__(self, x):
        """Forward on the host (weights read back from HBM), WITHOUT gradients: the device paths are the packaged recommenders
        (GMF's fused pointwise step reads and updates `layers[0].kernel` in HBM; DLRM runs its MLPs inside `orx_dlrm_step`).
        Under a GradientTape -- where the caller expects to train through it -- this says so once."""
        from ._expr import Expr
        if isinstance(x, Expr) and x.op == "mul" and len(self.layers) == 1 and self.layers[0].units == 1 and not self.layers[0].use_bias \
                and self.layers[0].activation is None:
            # gmf.py:28 / :39: Dense(1, no bias) of (user rows * item rows): the fused GMF step / scorer reads this kernel in HBM
            from .latent_factor import GatheredRows
            rows = [a for a in x.args if isinstance(a, GatheredRows)] + [a.args[0] for a in x.args if isinstance(a, Expr) and a.op == "expand_dims"
                                                                          and isinstance(a.args[0], GatheredRows)]
            if rows:
                self.build(rows[0].factor.dim, rows[0].factor.table.ctx)
                return Expr("dense1", self, x)
        from .._lazy import active_tape
        if active_tape() is not None:
            from ._compose import host_fallback
            host_fallback("MLP")
        x = np.asarray(x, np.float32)
        self.build(x.shape[-1])
        for layer in self.layers:
            x = x @ layer.kernel.read()
            if layer.bias is not None:
                x = x + layer.bias.read()
            if layer.activation == 'relu':
                x = np.maximum(x, 0)
            elif layer.activation == 'sigmoid':
                x = 1.0 / (1.0 + np.exp(-x))
        return x
