#!/usr/bin/env python3
"""DLRM on synthetic Criteo-shaped data: the training loop of the reference's tf2_examples/dlrm_criteo.py
(13 dense features, 26 categorical ones, m_spa 4, bottom 8-4, top 128-64-1, Adam, batch 1024) written against
openrec_amd.  The Criteo files are not part of the reference repository, so labels come from a planted logistic
model over a few of the features: the MSE loss falls and the AUC of the predictions rises.

    python examples/dlrm_synthetic.py [--iters 300]

`reference_compat=False` uses the evidently intended pairwise interaction; the default (True) reproduces the
reference's triangle bug, with which the interaction terms (and the embedding gradients) are all zero
(SURVEY.md E.1).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openrec_amd.tf2.compat import tf, optimizers                 # noqa: E402
from openrec_amd.tf2.recommenders import DLRM                     # noqa: E402


def auc(score, label):
    order = np.argsort(score)
    rank = np.empty(len(score)); rank[order] = np.arange(1, len(score) + 1)
    npos = label.sum(); nneg = len(label) - npos
    return (rank[label > 0].sum() - npos * (npos + 1) / 2) / (npos * nneg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--compat", action="store_true", help="reproduce the reference's interaction bug")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    counts = [int(x) for x in rng.integers(3, 5000, 26)]
    wd = rng.normal(size=13) * 0.5
    we = [rng.normal(size=n) for n in counts[:6]]

    def batch(n):
        dense = np.log1p(rng.integers(0, 100, (n, 13))).astype(np.float32)
        sparse = np.stack([rng.integers(0, c, n) for c in counts], 1).astype(np.int32)
        logit = (dense - dense.mean(0)) @ wd + sum(we[f][sparse[:, f]] for f in range(6)) - 1.0
        label = (rng.random(n) < 1 / (1 + np.exp(-logit))).astype(np.float32)
        return dense, sparse, label

    dlrm_model = DLRM(m_spa=4, ln_emb=counts, ln_bot=[8, 4], ln_top=[128, 64, 1], reference_compat=args.compat)
    optimizer = optimizers.Adam(0.01)
    train_loss = tf.keras.metrics.Mean()
    dv, sv, lv = batch(8192)

    @tf.function
    def train_step(dense, sparse, label):
        with tf.GradientTape() as tape:
            loss_value = dlrm_model(dense, sparse, label)
        gradients = tape.gradient(loss_value, dlrm_model.trainable_variables)
        optimizer.apply_gradients(zip(gradients, dlrm_model.trainable_variables))
        return loss_value

    for it in range(1, args.iters + 1):
        train_loss(train_step(*batch(args.batch)))
        if it % 100 == 0 or it == args.iters:
            pred = dlrm_model.inference(dv, sv)
            print(f"iter {it}: train loss {float(train_loss.result()):.4f}  validation AUC {auc(pred, lv):.3f}")
            train_loss.reset_states()


if __name__ == "__main__":
    main()
