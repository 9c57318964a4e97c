"""CPU compute backend for openrec_amd/sharded_dlrm.py, used ONLY by the tests: the building blocks of
the C ABI (gather_rows / apply_rows / dlrm grads / dense pack + apply) restated with the NumPy oracle, so
that the exchange plan can be exercised with gloo on CPU."""
import numpy as np

from oracle import numpy_oracle as orc
from oracle.dlrm_oracle import DLRMOracle


class Tab:
    def __init__(self, rows, dim):
        self.w = np.zeros((rows, dim), np.float32)


def make_opt(kind, lr):
    return {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7),
            "adam": lambda: orc.AdamTFSparse(lr)}[kind]()


class OracleDLRMBackend:
    def __init__(self, cfg, opt_kind, lr, seed=0):
        self.model = DLRMOracle(seed=seed, **cfg)
        self.model.emb = []                                   # the tables live in the sharded engine
        self.opt = make_opt(opt_kind, lr)
        self.g = None

    def make_table(self, rows, dim, seed):
        return Tab(max(rows, 1), dim)

    def write_table(self, table, values):
        v = np.asarray(values, np.float32)
        table.w[:v.shape[0]] = v

    def read_table(self, table):
        return table.w.copy()

    def gather_rows(self, table, ids, out):
        i = ids.numpy(); m = i >= 0
        out.numpy()[m] = table.w[i[m]]

    def apply_rows(self, table, ids, grads):
        i = ids.numpy(); m = i >= 0
        if hasattr(self.opt, "begin_step") and not getattr(self, "_began", False):
            self.opt.begin_step()
        self._began = False
        self.opt.apply(table.w, i[m], grads.numpy()[m], key=("emb", 0))

    def _params(self):
        out = []
        for name, layers in (("bot", self.model.bot), ("top", self.model.top)):
            for l, (W, b) in enumerate(layers):
                out.append(((name, l, "W"), W)); out.append(((name, l, "b"), b))
        return out

    def grads(self, dense, emb_rows, label, global_b, emb_grads, loss_accum):
        B = label.numel()
        er = emb_rows.numpy().reshape(B, -1, self.model.m_spa)
        loss, g = self.model.loss_and_grads(dense.numpy(), None, label.numpy(), emb_rows=er, global_batch=global_b)
        emb_grads.numpy()[:] = g["emb"].reshape(emb_grads.shape)
        loss_accum += float(loss)
        self.g = g

    def dense_count(self):
        return int(sum(p.size for _, p in self._params()))

    def dense_pack(self, flat):
        parts = []
        for name in ("bot", "top"):
            for gW, gb in self.g[name]:
                parts += [gW.reshape(-1), gb.reshape(-1)]
        flat.numpy()[:] = np.concatenate(parts).astype(np.float32)

    def dense_apply(self, flat):
        f = flat.numpy(); o = 0
        if hasattr(self.opt, "begin_step"):
            self.opt.begin_step(); self._began = True
        for key, p in self._params():
            self.opt.apply_dense(p, f[o:o + p.size].reshape(p.shape).astype(p.dtype), key=key); o += p.size
