#!/usr/bin/env python3
"""All-item scorer (Recommender.inference, bpr.py:39-43) at evaluation scale: `--users` users x 1 M items x dim 64, the shape
of eval_step (tf2_examples/bpr_citeulike.py:41-46) on BASELINE's synthetic tables.  Kernel time from dispatch-attached
events (ORX_K_GEMM slot); prints one JSON line: achieved HBM rate on the algorithmic bytes (the nq x NI fp32 output
dominates: 4 bytes per user-item pair, plus the item rows and biases once) and the fp32-MFMA rate (2 D flops per pair).
    python scripts/bench_score.py [--users 1000] [--items 1000000] [--dim 64] [--kind dot|l2|gmf] [--simple]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--kind", default="dot")
    ap.add_argument("--simple", action="store_true", help="the round-1 VALU kernel (ORX_SCORE_SIMPLE)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--evaluate", action="store_true", help="also time the evaluation step end to end (scores + AUC + Recall)")
    ap.add_argument("--pos", type=int, default=10); ap.add_argument("--excl", type=int, default=200)
    args = ap.parse_args()
    if args.simple:
        os.environ["ORX_SCORE_SIMPLE"] = "1"
    from openrec_amd import runtime as rt
    ctx = rt.default_context()
    U = rt.Table(1_000_000, args.dim, ctx).init_uniform(seed=0); V = rt.Table(args.items, args.dim, ctx).init_uniform(seed=1)
    b = rt.Table(args.items, 1, ctx).init_uniform(seed=2); w = rt.Table(args.dim, 1, ctx).init_uniform(seed=3)
    uid = np.random.default_rng(0).integers(0, 1_000_000, args.users).astype(np.int32)
    rt.score_all_items(args.kind, U, V, b, uid[:64], w=w)
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        out = rt.score_all_items(args.kind, U, V, b, uid, w=w)
    wall = (time.perf_counter() - t0) / args.reps
    ctx.prof_enable(False)
    p = ctx.prof_get()["gemm"]
    us = p["total_ms"] / p["launches"] * 1e3
    pairs = args.users * args.items
    alg = pairs * 4 + args.items * (args.dim + 1) * 4 + args.users * args.dim * 4
    ev = None
    if args.evaluate:
        # eval_step end to end (tf2_examples/bpr_citeulike.py:41-46): scores + AUC + Recall@{50,100} for `users` users, the
        # masks as item lists (`--pos` positives, `--excl` excluded items per user), wall clock of the host call
        rng = np.random.default_rng(1)
        pos = rt.SparseMask.from_lists([rng.choice(args.items, args.pos, replace=False) for _ in range(args.users)], args.items)
        excl = rt.SparseMask.from_lists([rng.choice(args.items, args.excl, replace=False) for _ in range(args.users)], args.items)
        kw = dict(kind=args.kind, user=U, item=V, bias=b, w=w, uid=uid)
        rt.rank_metrics_csr(pos, excl, [50, 100], **kw)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            r = rt.rank_metrics_csr(pos, excl, [50, 100], **kw)
        ev = {"evaluate_ms": (time.perf_counter() - t0) / args.reps * 1e3, "positives_per_user": args.pos, "excluded_per_user": args.excl,
              "auc_mean": float(np.nanmean(r["auc"])), "masks": "item lists -> device bitmaps (orx_rank_metrics_csr)"}
        if args.users * args.items <= 200_000_000:          # the byte-mask entry point for comparison where the masks fit comfortably
            pd, ed = np.asarray(pos), np.asarray(excl)
            rt.rank_metrics(pd, ed, [50, 100], **kw)
            t0 = time.perf_counter()
            r2 = rt.rank_metrics(pd, ed, [50, 100], **kw)
            ev["evaluate_ms_byte_masks"] = (time.perf_counter() - t0) * 1e3
            ev["same_auc"] = bool(np.array_equal(r["auc"], r2["auc"], equal_nan=True))
    print(json.dumps({"evaluate": ev, "metric": "all-item scoring, user-item pairs/s", "value": pairs / (us * 1e-6), "kernel_us": us,
                      "config": {"workload": f"{args.kind} {args.users} users x {args.items} items x dim {args.dim}",
                                 "kernel": "score_all_kernel (VALU)" if args.simple else "score_mfma_kernel"},
                      "roofline": {"bound": "hbm", "achieved": alg / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": alg / (us * 1e-6) / 1e9 / 8000.0, "algorithmic_bytes": alg},
                      "mfma_tflops": 2.0 * args.dim * pairs / (us * 1e-6) / 1e12, "mfma_peak_fp32_tflops": 157.3,
                      "host_call_s_incl_copy_to_host": wall, "checksum": float(out[:8, :1000].sum())}))


if __name__ == "__main__":
    main()
