#!/usr/bin/env python3
"""BPR on a synthetic CiteULike-shaped dataset (5551 users x 16980 items, ~194k train records):
the training / evaluation loop of the reference's tf2_examples/bpr_citeulike.py, written against
openrec_amd.  Every train step (forward on the pre-step tables, gradients of loss + l2_loss, Adam
sparse apply) is ONE fused device call; evaluation computes AUC / Recall on the device.

    python examples/bpr_synthetic.py [--iters 300] [--eval-interval 100]

The real CiteULike files are not part of the reference repository (README.md:77-79), so the
interactions are drawn from a planted low-rank preference model: AUC starts at 0.5 and rises.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openrec_amd.tf2.compat import tf, optimizers                 # noqa: E402
from openrec_amd.tf2.data import Dataset                          # noqa: E402
from openrec_amd.tf2.metrics import DictMean                      # noqa: E402
from openrec_amd.tf2.recommenders import BPR                      # noqa: E402


def synthetic_citeulike(total_users=5551, total_items=16980, per_user=35, rank=8, seed=0):
    rng = np.random.default_rng(seed)
    pu, qi = rng.normal(size=(total_users, rank)), rng.normal(size=(total_items, rank))
    rec = []
    for u in range(total_users):
        cand = rng.choice(total_items, 400, replace=False)
        top = cand[np.argsort(-(qi[cand] @ pu[u]))[:per_user + 5]]
        rec += [(u, i) for i in top]
    rec = np.array(rec, dtype=[("user_id", np.int32), ("item_id", np.int32)])
    rng.shuffle(rec)
    is_val = np.zeros(len(rec), bool)
    is_val[rng.choice(len(rec), total_users * 5, replace=False)] = True
    return dict(train_data=rec[~is_val], val_data=rec[is_val], total_users=total_users, total_items=total_items)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--eval-interval", type=int, default=100)
    ap.add_argument("--eval-users", type=int, default=1000)
    args = ap.parse_args()
    raw_data = synthetic_citeulike()
    dim_embed, batch_size = 50, 1000                                # tf2_examples/bpr_citeulike.py:12-14
    train_dataset = Dataset(raw_data["train_data"], raw_data["total_users"], raw_data["total_items"], seed=1)
    val_dataset = Dataset(raw_data["val_data"], raw_data["total_users"], raw_data["total_items"], seed=2)
    bpr_model = BPR(total_users=raw_data["total_users"], total_items=raw_data["total_items"],
                    dim_user_embed=dim_embed, dim_item_embed=dim_embed)
    optimizer = optimizers.Adam()

    @tf.function
    def train_step(user_id, p_item_id, n_item_id):
        with tf.GradientTape() as tape:
            loss_value = bpr_model(user_id, p_item_id, n_item_id)
        gradients = tape.gradient(loss_value, bpr_model.trainable_variables)
        optimizer.apply_gradients(zip(gradients, bpr_model.trainable_variables))
        return loss_value

    def eval_step(user_id, pos_mask, excl_mask):
        m = bpr_model.evaluate(user_id, pos_mask, excl_mask, at=[50, 100])
        return {"AUC": m["auc"], "Recall": m["recall"]}

    average_loss = tf.keras.metrics.Mean()
    average_metrics = DictMean({"AUC": [], "Recall": [2]})
    history = []
    for train_iter, batch_data in enumerate(train_dataset.pairwise(batch_size=batch_size, take=args.iters + 1)):
        loss = train_step(**batch_data)
        average_loss.update_state(loss)
        if train_iter % args.eval_interval == 0:
            seen = 0
            for eval_batch in val_dataset.evaluation(batch_size=250, excl_datasets=[train_dataset]):
                average_metrics.update_state(eval_step(**eval_batch))
                seen += len(eval_batch["user_id"])
                if seen >= args.eval_users:
                    break
            result = average_metrics.result()
            history.append(float(result["AUC"]))
            print("Iter: %d, Loss: %.4f, AUC: %.4f, Recall(50, 100): %s" % (
                train_iter, average_loss.result(), result["AUC"], result["Recall"]), flush=True)
            average_loss.reset_states()
            average_metrics.reset_states()
    return history


if __name__ == "__main__":
    main()
