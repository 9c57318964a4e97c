#!/usr/bin/env python3
"""Golden fixtures for the DLRM train step (tests/golden/dlrm_*.npz), minted by
torch CPU autograd in float64 over the graph the reference defines
(openrec/tf2/recommenders/dlrm.py:63-100 + second_order_feature_interaction.py:12-34,
including -- in `compat` cases -- its lower/upper-triangle bug), with the Keras
optimizer rules applied to the dense autograd gradients.

Run:  python tests/golden/make_golden_dlrm.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.dlrm_oracle import DLRMOracle          # only for the parameter init / shapes
from make_golden import Opt, OPTS                  # the independent optimizer restatement

torch.set_default_dtype(torch.float64)

CFG = dict(m_spa=4, ln_emb=[7, 5, 11], ln_bot=[8, 4], ln_top=[16, 8, 1], dense_dim=13)
B = 48


def run(name, optkind, **kw):
    compat = kw.pop("reference_compat")
    o = DLRMOracle(dtype=np.float32, seed=3, reference_compat=compat, **CFG, **kw)
    rng = np.random.default_rng(5)
    dense = np.log1p(rng.integers(0, 50, (B, CFG["dense_dim"]))).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, B) for n in CFG["ln_emb"]], 1).astype(np.int32)
    label = (rng.uniform(size=B) < 0.3).astype(np.float32)
    P = {}
    for f, e in enumerate(o.emb):
        P[f"emb{f}"] = torch.tensor(e.astype(np.float64), requires_grad=True)
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            P[f"{nm}{l}W"] = torch.tensor(W.astype(np.float64), requires_grad=True)
            P[f"{nm}{l}b"] = torch.tensor(b.astype(np.float64), requires_grad=True)
    inputs = {("in_" + k): v.detach().numpy().astype(np.float32) for k, v in P.items()}
    opt = Opt(optkind, **OPTS[optkind])
    F = len(CFG["ln_emb"]) + 1
    itself = kw.get("arch_interaction_itself", False)
    losses = []
    td, tl = torch.tensor(dense.astype(np.float64)), torch.tensor(label.astype(np.float64))
    ts = torch.tensor(sparse.astype(np.int64))

    def mlp(x, nm, n, acts):
        for l in range(n):
            x = x @ P[f"{nm}{l}W"] + P[f"{nm}{l}b"]
            x = torch.relu(x) if acts[l] == "relu" else torch.sigmoid(x)
        return x

    for step in range(2):
        x = mlp(td, "bot", len(o.bot), o.bot_act)
        Z = torch.stack([P[f"emb{f}"][ts[:, f]] for f in range(F - 1)] + [x], 1)
        dots = Z @ Z.transpose(1, 2)
        ones = torch.ones(F, F)
        if compat:
            dots = torch.tril(dots)                                  # LinearOperatorLowerTriangular(..).to_dense()
            mask = torch.triu(ones, 0)                               # band_part(ones, 0, -1)
            if not itself:
                mask = mask - torch.diag(torch.ones(F))              # - band_part(ones, 0, 0)
        else:
            mask = torch.tril(ones, 0 if itself else -1)
        inter = dots[:, mask.bool()]
        p = mlp(torch.cat([x, inter], 1), "top", len(o.top), o.top_act)
        thr = kw.get("loss_threshold", 0.0)
        if 0.0 < thr < 1.0:
            p = torch.clamp(p, thr, 1 - thr)
        p = p.reshape(-1)
        if kw.get("loss_func", "mse") == "mse":
            loss = ((tl - p) ** 2).mean()
        else:
            eps = 1e-7
            pc = torch.clamp(p, eps, 1 - eps)
            loss = -(tl * torch.log(pc + eps) + (1 - tl) * torch.log(1 - pc + eps)).mean()
        names = list(P)
        grads = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
        opt.step_begin()
        for k, g in zip(names, grads):
            g = torch.zeros_like(P[k]) if g is None else g
            nv, _ = opt.apply(k, P[k].detach(), g)
            P[k] = nv.clone().requires_grad_(True)
        losses.append(float(loss.detach()))
    out = dict(inputs)
    out.update({("out_" + k): v.detach().numpy().astype(np.float32) for k, v in P.items()})
    out.update(dense=dense, sparse=sparse, label=label, losses=np.array(losses))
    np.savez_compressed(os.path.join(HERE, f"dlrm_{name}_{optkind}.npz"), **out)
    print(name, optkind, losses)


if __name__ == "__main__":
    run("compat", "sgd", reference_compat=True)
    run("compat", "adam", reference_compat=True)
    run("compatself", "sgd", reference_compat=True, arch_interaction_itself=True)
    run("intended", "sgd", reference_compat=False)
    run("intended", "adagrad", reference_compat=False)
    run("intendedbce", "adam", reference_compat=False, loss_func="bce", loss_threshold=0.05, sigmoid_bot=True)
