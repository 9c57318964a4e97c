"""LatentFactor: the embedding table of the reference
(openrec/tf2/modules/latent_factor.py:4-23: a Keras `Embedding` with a
`censor` method), backed by an HBM-resident `runtime.Table`."""
from __future__ import annotations

import numpy as np

from ... import runtime as rt

_seed_counter = [0]


class Variable:
    """What `layer.variables[0]` / `model.trainable_variables` hand out."""

    def __init__(self, table, name):
        self.table, self.name = table, name

    @property
    def shape(self):
        return self.table.shape

    def numpy(self):
        return self.table.read()

    def assign(self, value):
        self.table.write(np.asarray(value, np.float32))
        return self

    def __repr__(self):
        return f"<Variable {self.name} shape={self.table.shape} (HBM)>"


class LatentFactor:

    def __init__(self, num_instances, dim, zero_init=False, name=None, ctx=None, seed=None):
        self.num_instances, self.dim, self.name = int(num_instances), int(dim), name
        self.table = rt.Table(self.num_instances, self.dim, ctx)
        if zero_init:
            self.table.fill(0.0)                       # initializer 'zeros'  (latent_factor.py:8-9)
        else:
            if seed is None:
                _seed_counter[0] += 1
                seed = _seed_counter[0]
            self.table.init_uniform(-0.05, 0.05, seed)  # Keras 'uniform'      (latent_factor.py:10-11)
        self._var = Variable(self.table, (name or "latent_factor") + "/embeddings")

    @property
    def variables(self):
        return [self._var]

    trainable_variables = variables

    def __call__(self, ids):
        """Embedding gather: [*] int ids -> [*, dim] fp32 (host array)."""
        ids = np.asarray(ids)
        out = self.table.gather(ids.reshape(-1))
        return out.reshape(ids.shape + (self.dim,))

    def censor(self, censor_id):
        """latent_factor.py:17-23: rows of the DISTINCT ids are divided by max(norm, 0.1)."""
        self.table.censor(censor_id, 0.1)
        return self._var
