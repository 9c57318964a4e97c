"""MLP factory (openrec/tf2/modules/multi_layer_perceptron.py:5-18): a stack of
Dense layers whose kernels / biases are HBM tables.  GMF uses
`MLP(units_list=[1], use_bias=False)`, a single [D, 1] kernel that the fused
pointwise kernel reads and updates in HBM; DLRM's bottom and top MLPs
(recommenders/dlrm.py:34-37) become the parameters of `orx_dlrm_step` when the
composition dlrm.py:76-100 is recognised (modules/_compose.py)."""
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

    def device_forward(self, x):
        """the stack applied to a plain array ON THE DEVICE (multi_layer_perceptron.py:5-18 outside a fused composition):
        `orx_mlp_forward` -- the exact-fp32 MFMA products of the DLRM step, bias and activation in their epilogues; the weights
        never leave HBM.  Values only (no gradients: under a tape the caller gets the fused compositions or an error)."""
        import ctypes
        from ... import _ffi
        x = np.ascontiguousarray(np.asarray(x, np.float32))
        lead, in_dim = x.shape[:-1], x.shape[-1]
        self.build(in_dim)
        x2 = x.reshape(-1, in_dim)
        n = len(self.layers)
        if n == 0:
            raise ValueError("MLP.device_forward: the stack has no layers")
        known = {None: 0, "linear": 0, "relu": 1, "sigmoid": 2}
        for i, l in enumerate(self.layers):
            if not (l.activation is None or isinstance(l.activation, str)) or l.activation not in known:
                raise NotImplementedError("MLP.device_forward: layer %d has activation %r; the device path (orx_mlp_forward) has "
                                          "None / 'linear' / 'relu' / 'sigmoid'" % (i, l.activation))
        ctx = self.layers[0].kernel.ctx
        kernels = (ctypes.c_void_p * n)(*[l.kernel._h for l in self.layers])
        biases = (ctypes.c_void_p * n)(*[(l.bias._h if l.bias is not None else None) for l in self.layers])
        acts = (ctypes.c_int32 * n)(*[known[l.activation] for l in self.layers])
        out = np.empty((x2.shape[0], self.layers[-1].units), np.float32)
        if x2.shape[0]:
            _ffi.check(ctx._lib.orx_mlp_forward(ctx._h, n, kernels, biases, acts, x2.ctypes.data, x2.shape[0], in_dim, 0, out.ctypes.data))
        return out.reshape(lead + (self.layers[-1].units,))

    def __call__(self, x):
        """A node of a lazy expression (modules/_expr.py): the compositions of the reference that contain an MLP run as fused
        device steps -- gmf.py:28 (`Dense(1, no bias)` of user rows * item rows: the fused GMF step / scorer reads the kernel in
        HBM) and dlrm.py:87-93 (bottom MLP -> feature interaction -> top MLP: `orx_dlrm_step`, whose parameters this object's
        layers then ARE).  A tree that matches neither runs through `orx_mlp_forward` (device) when somebody looks at its values -- outside a
        tape: under a GradientTape, where the caller expects to train through it, that raises."""
        from ._expr import Expr
        if isinstance(x, Expr) and x.op == "mul" and len(self.layers) == 1 and self.layers[0].units == 1 and not self.layers[0].use_bias \
                and self.layers[0].activation is None:
            from .latent_factor import GatheredRows
            rows = [a for a in x.args if isinstance(a, GatheredRows)] + [a.args[0] for a in x.args if isinstance(a, Expr) and a.op == "expand_dims"
                                                                          and isinstance(a.args[0], GatheredRows)]
            if rows:
                self.build(rows[0].factor.dim, rows[0].factor.table.ctx)
                return Expr("dense1", self, x)
        return Expr("mlp", self, x)
