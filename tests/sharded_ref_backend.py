"""CPU compute backend for the sharded engine, used ONLY by the tests: the same
three building blocks as the C ABI (gather_rows / pair_grads / apply_rows),
restated with the NumPy oracle, so that the exchange plan in
openrec_amd/sharded.py can be exercised with gloo on CPU."""
import numpy as np
import torch

from oracle import numpy_oracle as orc


class Tab:
    def __init__(self, rows, dim):
        self.w = np.zeros((rows, dim), np.float32)


class OracleBackend:
    def __init__(self, opt_kind, lr):
        self.opt_kind, self.lr = opt_kind, lr
        self.opt = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[opt_kind]()

    def begin_step(self):
        if hasattr(self.opt, "begin_step"):
            self.opt.begin_step()

    def make_table(self, rows, dim, seed):
        return Tab(max(rows, 1), dim)

    def gather_rows(self, table, bias, ids, out):
        i = ids.numpy()
        o = out.numpy()
        m = i >= 0
        D = table.w.shape[1]
        o[m, :D] = table.w[i[m]]
        if bias is not None:
            o[m, D] = bias.w[i[m], 0]

    def pair_grads(self, model, dim, u, p, n, valid, b_global, margin, gu, gp, gn, accum):
        m = valid.numpy() >= 0
        un, pn, nn = u.numpy()[m], p.numpy()[m], n.numpy()[m]
        k = un.shape[0]
        if k == 0:
            return
        # the oracle works on tables + ids: use the gathered rows as 3 private tables
        U, V, b = un[:, :dim].copy(), np.concatenate([pn[:, :dim], nn[:, :dim]]), np.concatenate([pn[:, dim], nn[:, dim]])[:, None]
        ar = np.arange(k)
        if model == "bpr":
            loss, l2, _ = orc.bpr_forward(U, V, b, ar, ar, ar + k)
            gr = orc.bpr_grads(U, V, b, ar, ar, ar + k)
            scale = np.float32(k) / np.float32(b_global)         # mean over the GLOBAL batch
            loss = loss * scale
            for key in ("gu", "gp", "gn"):
                l2part = {"gu": U, "gp": V[:k], "gn": V[k:]}[key]
                gr[key] = (gr[key] - l2part) * scale + l2part     # only the loss part scales
            gr["gbp"] = gr["gbp"] * scale; gr["gbn"] = gr["gbn"] * scale
        else:
            loss, l2, _ = orc.ucml_forward(U, V, b, ar, ar, ar + k, margin)
            gr = orc.ucml_grads(U, V, b, ar, ar, ar + k, margin)
        gu.numpy()[m, :dim] = gr["gu"]; gp.numpy()[m, :dim] = gr["gp"]; gn.numpy()[m, :dim] = gr["gn"]
        gp.numpy()[m, dim] = gr["gbp"]; gn.numpy()[m, dim] = gr["gbn"]
        accum += torch.tensor([float(loss), float(l2)], dtype=torch.float64)

    def apply_rows(self, table, bias, ids, grads):
        i = ids.numpy()
        m = i >= 0
        D = table.w.shape[1]
        g = grads.numpy()[m]
        self.opt.apply(table.w, i[m], g[:, :D], key=id(table))
        if bias is not None:
            self.opt.apply(bias.w, i[m], g[:, D:D + 1], key=id(bias))
