#!/usr/bin/env python3
"""Generate the committed golden fixtures in tests/golden/*.npz.

The reference has no golden vectors for this path and TensorFlow is not
installable here, so the pins are minted by an INDEPENDENT second
implementation: torch CPU autograd in float64 over the graph that the
reference defines (openrec/tf2/recommenders/{bpr,ucml,gmf,wrmf}.py,
modules/{pairwise_log_loss,pointwise_mse_loss}.py), with the Keras optimizer
rules applied to the dense (= per-row summed) autograd gradients.

Run:  python tests/golden/make_golden.py     (needs torch; no GPU; ~2 s)
The .npz files hold inputs (tables, ids) and outputs (loss, l2_loss, updated
tables, optimizer slots) so that nothing has to be regenerated on the GPU box.
"""
import os
import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_default_dtype(torch.float64)

NU, NI, B = 24, 40, 96


def make_inputs(seed, D, pointwise=False):
    rng = np.random.default_rng(1000 + seed)
    # inputs are float32-representable so that fp32 and fp64 runs start equal
    U = rng.uniform(-0.05, 0.05, (NU, D)).astype(np.float32)
    V = rng.uniform(-0.05, 0.05, (NI, D)).astype(np.float32)
    b = rng.uniform(-0.05, 0.05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, B).astype(np.int32)
    pid = rng.integers(0, NI, B).astype(np.int32)
    nid = rng.integers(0, NI, B).astype(np.int32)
    # adversarial duplicates: one hot user, p == n collisions, boundary ids
    uid[:16] = 7
    nid[16:24] = pid[16:24]
    uid[24], pid[24], nid[24] = 0, 0, NI - 1
    uid[25], pid[25], nid[25] = NU - 1, NI - 1, 0
    label = (rng.uniform(size=B) < 0.5).astype(np.float32)
    w = rng.uniform(-0.3, 0.3, (D, 1)).astype(np.float32)
    return dict(U=U, V=V, b=b, uid=uid, pid=pid, nid=nid, label=label, w=w)


def t(x, grad=True):
    return torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=grad)


def l2_loss(x):
    return (x * x).sum() / 2


def bpr_loss(U, V, b, uid, pid, nid):
    u, p, n = F.embedding(uid, U), F.embedding(pid, V), F.embedding(nid, V)
    bp, bn = F.embedding(pid, b), F.embedding(nid, b)
    pos = (u * p).sum(1, keepdim=True) + bp
    neg = (u * n).sum(1, keepdim=True) + bn
    loss = -F.logsigmoid(torch.maximum(pos - neg, torch.tensor(-30.0))).mean()
    return loss, l2_loss(u) + l2_loss(p) + l2_loss(n)


def ucml_loss(U, V, b, uid, pid, nid, margin=0.5):
    u, p, n = F.embedding(uid, U), F.embedding(pid, V), F.embedding(nid, V)
    bp, bn = F.embedding(pid, b), F.embedding(nid, b)
    dp = ((u - p) ** 2).sum(-1, keepdim=True)
    dn = ((u - n) ** 2).sum(-1, keepdim=True)
    diff = (-dp + bp) - (-dn + bn)
    loss = torch.clamp(margin - diff, min=0).sum()
    return loss, l2_loss(u) + l2_loss(p) + l2_loss(n)


def gmf_loss(U, V, b, w, uid, iid, label):
    u, i = F.embedding(uid, U), F.embedding(iid, V)
    bi = F.embedding(iid, b)
    logit = ((u * i) @ w + bi).reshape(-1)
    loss = F.binary_cross_entropy_with_logits(logit, label, reduction="mean")
    return loss, l2_loss(u) + l2_loss(i) + l2_loss(w)


def wrmf_loss(U, V, b, uid, iid, label, a, bw):
    u, i = F.embedding(uid, U), F.embedding(iid, V)
    bi = F.embedding(iid, b).reshape(-1)
    pred = (u * i).sum(1) + bi
    c = (a - bw) * label + bw
    loss = (c * (label - pred) ** 2).sum()
    return loss, l2_loss(u) + l2_loss(i)


class Opt:
    def __init__(self, kind, **kw):
        self.kind, self.kw, self.slots, self.t = kind, kw, {}, 0

    def step_begin(self):
        self.t += 1

    def apply(self, name, var, grad):
        """Dense restatement of the TF-2.0 Keras sparse rules: rows with zero
        summed gradient behave exactly like untouched rows."""
        k = self.kind
        if k == "sgd":
            return var - self.kw["lr"] * grad, {}
        if k == "adagrad":
            acc = self.slots.setdefault(name + "/acc", torch.full_like(var, self.kw["init_acc"]))
            acc = acc + grad * grad
            self.slots[name + "/acc"] = acc
            return var - self.kw["lr"] * grad / (acc.sqrt() + self.kw["eps"]), {}
        if k == "adam":
            b1, b2 = self.kw["b1"], self.kw["b2"]
            m = self.slots.setdefault(name + "/m", torch.zeros_like(var))
            v = self.slots.setdefault(name + "/v", torch.zeros_like(var))
            m = b1 * m + (1 - b1) * grad
            v = b2 * v + (1 - b2) * grad * grad
            self.slots[name + "/m"], self.slots[name + "/v"] = m, v
            lr_t = self.kw["lr"] * np.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
            return var - lr_t * m / (v.sqrt() + self.kw["eps"]), {}
        raise ValueError(k)


OPTS = {
    "sgd": dict(lr=0.05),
    "adagrad": dict(lr=0.05, init_acc=0.1, eps=1e-7),
    "adam": dict(lr=0.001, b1=0.9, b2=0.999, eps=1e-7),
}


def run_case(model, D, optkind, seed, steps=2):
    inp = make_inputs(seed, D)
    names = ["U", "V", "b"] + (["w"] if model == "gmf" else [])
    P = {k: t(inp[k]) for k in names}
    opt = Opt(optkind, **OPTS[optkind])
    uid, pid, nid = (torch.tensor(inp[k].astype(np.int64)) for k in ("uid", "pid", "nid"))
    label = torch.tensor(inp["label"].astype(np.float64))
    losses = []
    for s in range(steps):
        # step s uses the ids rolled by s so that the second step sees new pairs
        u_, p_, n_ = uid.roll(s), pid.roll(2 * s), nid.roll(3 * s)
        lab = label.roll(s)
        if model == "bpr":
            loss, l2 = bpr_loss(P["U"], P["V"], P["b"], u_, p_, n_)
        elif model == "ucml":
            loss, l2 = ucml_loss(P["U"], P["V"], P["b"], u_, p_, n_)
        elif model == "gmf":
            loss, l2 = gmf_loss(P["U"], P["V"], P["b"], P["w"], u_, p_, lab)
        elif model == "wrmf":
            loss, l2 = wrmf_loss(P["U"], P["V"], P["b"], u_, p_, lab, 2.0, 0.5)
        grads = torch.autograd.grad(loss + l2, [P[k] for k in names])
        opt.step_begin()
        newP = {}
        for k, g in zip(names, grads):
            nv, _ = opt.apply(k, P[k].detach(), g)
            newP[k] = nv.clone().requires_grad_(True)
        if s == 0:
            grads0 = {k: g.numpy().copy() for k, g in zip(names, grads)}
        P = newP
        losses.append((float(loss), float(l2)))
    out = {("in_" + k): v for k, v in inp.items()}
    # outputs: fp64 truth rounded to fp32 for storage (fixtures stay small)
    out.update({("out_" + k): P[k].detach().numpy().astype(np.float32) for k in names})
    out["grad0_b"] = grads0["b"].astype(np.float32)
    out.update({("slot_" + k.replace("/", "_")): v.numpy().astype(np.float32) for k, v in opt.slots.items()})
    out["losses"] = np.array(losses)
    out["steps"] = np.array(steps)
    return out


def main():
    cases = []
    for D in (50, 64, 128):
        for ok in ("sgd", "adagrad", "adam"):
            cases.append(("bpr", D, ok, 0))
    cases += [("bpr", 64, "sgd", 1), ("bpr", 64, "sgd", 2)]
    cases += [("ucml", 64, "sgd", 0), ("ucml", 128, "sgd", 0), ("ucml", 128, "adagrad", 1)]
    cases += [("gmf", 64, "sgd", 0), ("gmf", 64, "adagrad", 1), ("wrmf", 64, "sgd", 0), ("wrmf", 64, "adagrad", 1)]
    total = 0
    for model, D, ok, seed in cases:
        out = run_case(model, D, ok, seed)
        fn = os.path.join(HERE, f"{model}_d{D}_{ok}_s{seed}.npz")
        np.savez_compressed(fn, **out)
        total += os.path.getsize(fn)
        print(fn, os.path.getsize(fn))
    print("total bytes", total)


if __name__ == "__main__":
    main()
