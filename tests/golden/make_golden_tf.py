#!/usr/bin/env python3
"""Mint the floating-point golden fixtures from the REFERENCE ITSELF.

    python tests/golden/make_golden_tf.py --backend tf   [--reference /root/reference] [--out tests/golden/tf]
    python tests/golden/make_golden_tf.py --backend stub [--dtype float64]             [--out tests/golden/refstub]

What runs: the reference's own classes, imported from <reference>/openrec/tf2 -- `openrec.tf2.recommenders.{BPR, UCML, GMF,
WRMF, DLRM}` (bpr.py:5-43, ucml.py:7-52, gmf.py:7-40, wrmf.py:7-39, dlrm.py:8-100) -- driven by the train step of the
reference's examples (tf2_examples/bpr_citeulike.py:33-39, dlrm_criteo.py:42-48: tape over the model's output, gradients of
the trainable variables, `optimizer.apply_gradients`); the `ucmlc_*` cases call `censor_vec` after every step (ucml.py:44-48),
the `metrics_*` case runs the reference's AUC / NDCG / Recall (metrics/ranking_metrics.py:8-68).  Nothing of the reference is
copied: it is imported and called.

  --backend tf    real TensorFlow (the reference pins tensorflow==2.0.1, docs_requirements.txt:2).  This is the command that
                  lifts "parity unpinned": on a machine with that wheel it writes tests/golden/tf/*.npz, and
                  tests/test_oracle_golden.py / tests/test_gpu_*.py prefer those files over the torch-autograd ones when present.
                  It cannot run in the build container (no TensorFlow, no network).
  --backend stub  the same script on tests/golden/tf_stub.py (a stand-in `tensorflow` on torch CPU autograd that provides the
                  symbols the reference touches).  The model graphs are then still the reference's text; TensorFlow's own
                  autodiff and the Keras sparse optimizer rules are the stand-in's restatement.  Runs here; its
                  output (tests/golden/refstub/*.npz) is committed and checked by tests/test_reference_goldens.py.

Cases: every committed case of make_golden.py / make_golden_dlrm.py (same inputs, same file names, same .npz schema); `--extra`:
UCML with censor (`ucmlc_*`) and the ranking metrics (`metrics_s0`); `--large`: seeds {0,1,2} at B = 1024, N = 4096,
D in {50, 64, 128}, stored as a selection of rows + checksums of the whole tables.
Schema (per file): in_* inputs, out_* tables after `steps` steps, slot_* optimizer slots, grad0_b (dense bias gradient of step 0),
losses [steps, 2] = (loss, l2_loss), steps, dtype (of the run), backend.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

OPTS = {"sgd": dict(learning_rate=0.05),
        "adagrad": dict(learning_rate=0.05, initial_accumulator_value=0.1, epsilon=1e-7),
        "adam": dict(learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7)}
OPT_CLASS = {"sgd": "SGD", "adagrad": "Adagrad", "adam": "Adam"}
SLOTS = {"sgd": [], "adagrad": [("accumulator", "acc")], "adam": [("m", "m"), ("v", "v")]}

PAIR_CASES = ([("bpr", D, ok, 0) for D in (50, 64, 128) for ok in ("sgd", "adagrad", "adam")] +
              [("bpr", 64, "sgd", 1), ("bpr", 64, "sgd", 2), ("ucml", 64, "sgd", 0), ("ucml", 128, "sgd", 0), ("ucml", 128, "adagrad", 1),
               ("gmf", 64, "sgd", 0), ("gmf", 64, "adagrad", 1), ("wrmf", 64, "sgd", 0), ("wrmf", 64, "adagrad", 1)])
DLRM_CASES = [("compat_sgd", "sgd", dict(reference_compat=True)),
              ("compat_adam", "adam", dict(reference_compat=True)),
              ("compatself_sgd", "sgd", dict(reference_compat=True, arch_interaction_itself=True))]
# (the `intended*` DLRM fixtures describe the interaction the reference MEANT -- lower triangle without its bug, SURVEY.md E.1;
#  the reference's text cannot produce them, so they stay with make_golden_dlrm.py)


def load_backend(backend, dtype, reference):
    if backend == "stub":
        import tf_stub
        tf_stub.uninstall()
        tf = tf_stub.install(dtype)
    else:
        import tensorflow as tf          # noqa: F401  (the real one)
        if not tf.__version__.startswith("2.0."):
            sys.stderr.write("WARNING: the reference pins tensorflow==2.0.1 (docs_requirements.txt:2); this is %s -- Adam's sparse apply "
                             "changed after 2.0.x (no dense decay), expect the Adam cases to differ\n" % tf.__version__)
    if not os.path.isdir(os.path.join(reference, "openrec", "tf2")):
        raise SystemExit("reference not found at %s" % reference)
    sys.path.insert(0, reference)
    from openrec.tf2 import recommenders
    return tf, recommenders


def to_np(x):
    if hasattr(x, "detach"):               # (the stand-in's tensors are torch tensors)
        x = x.detach()
    return np.asarray(x.numpy() if hasattr(x, "numpy") else x)


def train_step(tf, model, opt, *inputs):
    """tf2_examples/bpr_citeulike.py:33-39 in this script's words"""
    with tf.GradientTape() as tape:
        out = model(*inputs)
    tv = model.trainable_variables
    grads = tape.gradient(out, tv)
    opt.apply_gradients(zip(grads, tv))
    return out, grads


def dense_of(tf, g, shape):
    if hasattr(g, "indices"):              # IndexedSlices -> dense (duplicates summed)
        d = np.zeros(shape, np.float64)
        np.add.at(d, to_np(g.indices).astype(np.int64).reshape(-1), to_np(g.values).astype(np.float64).reshape(-1, shape[-1]))
        return d
    return to_np(g).astype(np.float64)


def run_pair_case(tf, rec, model_name, D, optkind, seed, steps=2, inp=None, sel=None, censor=False):
    """one fixture of the BPR / UCML / GMF / WRMF family; `inp` as make_golden.make_inputs returns it"""
    from make_golden import make_inputs
    if inp is None:
        inp = make_inputs(seed, D)
    NU, NI = inp["U"].shape[0], inp["V"].shape[0]
    kw = dict(dim_user_embed=D, dim_item_embed=D, total_users=NU, total_items=NI)
    if model_name == "bpr":
        model = rec.BPR(**kw)
    elif model_name == "ucml":
        model = rec.UCML(margin=0.5, **kw)
    elif model_name == "gmf":
        model = rec.GMF(**kw)
    else:
        model = rec.WRMF(a=2.0, b=0.5, **kw)
    fdt = np.float64 if "float64" in str(getattr(tf, "float32", "")) else np.float32      # (the stub's float type may be float64)
    ids = lambda a: tf.constant(a.astype(np.int32), dtype=tf.int32)
    lab = lambda a: tf.constant(a.astype(fdt))
    pointwise = model_name in ("gmf", "wrmf")
    first = (ids(inp["uid"]), ids(inp["pid"]), lab(inp["label"])) if pointwise else (ids(inp["uid"]), ids(inp["pid"]), ids(inp["nid"]))
    model(*first)                                               # builds every layer (Embedding / Dense create their variables on first use)
    model.user_latent_factor.set_weights([inp["U"].astype(fdt)])
    model.item_latent_factor.set_weights([inp["V"].astype(fdt)])
    model.item_bias.set_weights([inp["b"].astype(fdt)])
    if model_name == "gmf":
        model.mlp.set_weights([inp["w"].astype(fdt)])
    opt = getattr(tf.keras.optimizers, OPT_CLASS[optkind])(**OPTS[optkind])
    losses, grad0_b = [], None
    for s in range(steps):       # step s uses the ids rolled by s (make_golden.py: the second step sees new pairs)
        u_, p_, n_, l_ = np.roll(inp["uid"], s), np.roll(inp["pid"], 2 * s), np.roll(inp["nid"], 3 * s), np.roll(inp["label"], s)
        args = (ids(u_), ids(p_), lab(l_)) if pointwise else (ids(u_), ids(p_), ids(n_))
        out, grads = train_step(tf, model, opt, *args)
        if censor:                                              # ucml.py:44-48: users, then pos items, then neg items
            model.censor_vec(*args)
        if s == 0:
            k = [id(v) for v in model.trainable_variables].index(id(model.item_bias.trainable_variables[0]))
            grad0_b = dense_of(tf, grads[k], (NI, 1))
        losses.append([float(to_np(out[0])), float(to_np(out[1]))])
    names = {"U": model.user_latent_factor, "V": model.item_latent_factor, "b": model.item_bias}
    res = {("in_" + k): v for k, v in inp.items()}
    for k, layer in names.items():
        res["out_" + k] = to_np(layer.variables[0]).astype(np.float32)
        for slot, short in SLOTS[optkind]:
            res["slot_%s_%s" % (k, short)] = to_np(opt.get_slot(layer.variables[0], slot)).astype(np.float32)
    if model_name == "gmf":
        res["out_w"] = to_np(model.mlp.trainable_variables[0]).astype(np.float32)
        for slot, short in SLOTS[optkind]:
            res["slot_w_%s" % short] = to_np(opt.get_slot(model.mlp.trainable_variables[0], slot)).astype(np.float32)
    res["grad0_b"] = grad0_b.astype(np.float32)
    res["losses"] = np.array(losses, np.float64)
    res["steps"] = np.array(steps)
    if sel is not None:          # large cases: only the rows in `sel` are stored (the others are checked through a checksum)
        for k in ("U", "V", "b"):
            for key in [key for key in res if key in ("in_" + k, "out_" + k) or key.startswith("slot_%s_" % k)]:
                full = res[key].astype(np.float64)
                res["sum_" + key] = np.array([full.sum(), np.abs(full).sum()])
                res[key] = res[key][sel["U" if k == "U" else "V"]]
        res["grad0_b"] = res["grad0_b"][sel["V"]]
        res["sel_U"], res["sel_V"] = sel["U"], sel["V"]
    return res


def run_metrics_case(tf, metrics, seed=0, n=6, NI=97):
    """the reference's AUC / NDCG / Recall (ranking_metrics.py:8-68) on random predictions and masks; ties included"""
    rng = np.random.default_rng(300 + seed)
    pred = rng.normal(size=(n, NI)).astype(np.float32)
    pred[:, 5] = pred[:, 6]                                     # tied scores
    pos = rng.uniform(size=(n, NI)) < 0.12
    pos[0, :3] = True
    excl = (rng.uniform(size=(n, NI)) < 0.2) & ~pos
    at = [5, 10, 50]
    fdt = np.float64 if "float64" in str(getattr(tf, "float32", "")) else np.float32
    tp, tpos, tex = tf.constant(pred.astype(fdt)), tf.constant(pos), tf.constant(excl)
    return dict(in_pred=pred, in_pos=pos, in_excl=excl, at=np.array(at),
                auc=to_np(metrics.AUC(pos_mask=tpos, pred=tp, excl_mask=tex)).astype(np.float64),
                ndcg=to_np(metrics.NDCG(pos_mask=tpos, pred=tp, excl_mask=tex, at=at)).astype(np.float64),
                recall=to_np(metrics.Recall(pos_mask=tpos, pred=tp, excl_mask=tex, at=at)).astype(np.float64))


def large_inputs(seed, D, B=1024, N=4096):
    rng = np.random.default_rng(5000 + 10 * seed + D)
    inp = dict(U=rng.uniform(-0.05, 0.05, (N, D)).astype(np.float32), V=rng.uniform(-0.05, 0.05, (N, D)).astype(np.float32),
               b=rng.uniform(-0.05, 0.05, (N, 1)).astype(np.float32), uid=rng.integers(0, N, B).astype(np.int32),
               pid=rng.integers(0, N, B).astype(np.int32), nid=rng.integers(0, N, B).astype(np.int32),
               label=(rng.uniform(size=B) < 0.5).astype(np.float32), w=rng.uniform(-0.3, 0.3, (D, 1)).astype(np.float32))
    inp["uid"][:24] = 11                  # a hot user, p == n collisions
    inp["nid"][24:32] = inp["pid"][24:32]
    return inp


def large_recipe(seed, D):
    """inputs are regenerated from (seed, D) by whoever reads the fixture: only the selection of rows is stored"""
    inp = large_inputs(seed, D)
    touched_u = np.unique(np.concatenate([np.roll(inp["uid"], s) for s in range(2)]))
    touched_v = np.unique(np.concatenate([inp["pid"], inp["nid"]]))
    rng = np.random.default_rng(seed)
    pick = lambda t, n_total: np.unique(np.concatenate([t[:48], rng.integers(0, n_total, 16)])).astype(np.int64)
    return inp, dict(U=pick(touched_u, inp["U"].shape[0]), V=pick(touched_v, inp["V"].shape[0]))


def run_dlrm_case(tf, rec, name, optkind, kw):
    """tests/golden/make_golden_dlrm.py's compat cases (the reference's interaction as written, triangle bug included) through the
    reference's own DLRM class; parameters and batch as that script makes them"""
    from oracle.dlrm_oracle import DLRMOracle                   # parameter init / shapes only
    from make_golden_dlrm import CFG, B
    kw = dict(kw)
    kw.pop("reference_compat")
    o = DLRMOracle(dtype=np.float32, seed=3, reference_compat=True, **CFG, **kw)
    rng = np.random.default_rng(5)
    dense = np.log1p(rng.integers(0, 50, (B, CFG["dense_dim"]))).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, B) for n in CFG["ln_emb"]], 1).astype(np.int32)
    label = (rng.uniform(size=B) < 0.3).astype(np.float32)
    fdt = np.float64 if "float64" in str(getattr(tf, "float32", "")) else np.float32
    model = rec.DLRM(m_spa=CFG["m_spa"], ln_emb=CFG["ln_emb"], ln_bot=CFG["ln_bot"], ln_top=CFG["ln_top"], **kw)
    td, ts, tl = tf.constant(dense.astype(fdt)), tf.constant(sparse, dtype=tf.int32), tf.constant(label.astype(fdt))
    model(td, ts, tl)
    res = {}
    for f, e in enumerate(o.emb):
        model._latent_factors[f].set_weights([e.astype(fdt)]); res["in_emb%d" % f] = e.astype(np.float32)
    for nm, mlp, layers in (("bot", model._mlp_bot, o.bot), ("top", model._mlp_top, o.top)):
        for l, (W, bb) in enumerate(layers):
            mlp.layers[l].set_weights([W.astype(fdt), bb.astype(fdt)])
            res["in_%s%dW" % (nm, l)], res["in_%s%db" % (nm, l)] = W.astype(np.float32), bb.astype(np.float32)
    opt = getattr(tf.keras.optimizers, OPT_CLASS[optkind])(**OPTS[optkind])
    losses = []
    for _ in range(2):
        out, _g = train_step(tf, model, opt, td, ts, tl)
        losses.append(float(to_np(out)))
    for f in range(len(o.emb)):
        res["out_emb%d" % f] = to_np(model._latent_factors[f].variables[0]).astype(np.float32)
    for nm, mlp, layers in (("bot", model._mlp_bot, o.bot), ("top", model._mlp_top, o.top)):
        for l in range(len(layers)):
            W, bb = mlp.layers[l].get_weights()
            res["out_%s%dW" % (nm, l)], res["out_%s%db" % (nm, l)] = np.asarray(W, np.float32), np.asarray(bb, np.float32)
    res.update(dense=dense, sparse=sparse, label=label, losses=np.array(losses, np.float64))
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--backend", choices=("tf", "stub"), default="tf")
    ap.add_argument("--dtype", choices=("float32", "float64"), default="float64", help="stub only (TensorFlow runs float32)")
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=None)
    ap.add_argument("--large", action="store_true", help="also the B=1024, N=4096 cases (SURVEY.md 8c)")
    ap.add_argument("--extra", action="store_true", help="also UCML with censor and the ranking metrics")
    ap.add_argument("--no-base", action="store_true", help="skip the cases make_golden.py / make_golden_dlrm.py already hold")
    ap.add_argument("--only", default=None, help="substring filter on case names")
    args = ap.parse_args(argv)
    out_dir = args.out or os.path.join(HERE, "tf" if args.backend == "tf" else "refstub")
    os.makedirs(out_dir, exist_ok=True)
    tf, rec = load_backend(args.backend, args.dtype, args.reference)
    stamp = dict(dtype=np.array("float32" if args.backend == "tf" else args.dtype), backend=np.array("%s %s" % (args.backend, tf.__version__)))
    total, written = 0, []
    def save(name, res):
        nonlocal total
        fn = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(fn, **res, **stamp)
        total += os.path.getsize(fn); written.append(fn)
        print(fn, os.path.getsize(fn))
    for model, D, ok, seed in ([] if args.no_base else PAIR_CASES):
        name = "%s_d%d_%s_s%d" % (model, D, ok, seed)
        if args.only and args.only not in name:
            continue
        save(name, run_pair_case(tf, rec, model, D, ok, seed))
    for name, ok, kw in ([] if args.no_base else DLRM_CASES):
        if args.only and args.only not in "dlrm_" + name:
            continue
        save("dlrm_" + name, run_dlrm_case(tf, rec, name, ok, kw))
    if args.extra:
        for D, ok, seed in ((64, "sgd", 0), (128, "adagrad", 1), (64, "adam", 2)):
            name = "ucmlc_d%d_%s_s%d" % (D, ok, seed)
            if not (args.only and args.only not in name):
                save(name, run_pair_case(tf, rec, "ucml", D, ok, seed, censor=True))
        if not (args.only and args.only not in "metrics_s0"):
            from openrec.tf2.metrics import ranking_metrics
            save("metrics_s0", run_metrics_case(tf, ranking_metrics))
    if args.large:
        for seed in (0, 1, 2):
            for D in (50, 64, 128):
                name = "large_bpr_d%d_sgd_s%d" % (D, seed)
                if args.only and args.only not in name:
                    continue
                inp, sel = large_recipe(seed, D)
                res = run_pair_case(tf, rec, "bpr", D, "sgd", seed, inp=inp, sel=sel)
                for k in ("in_uid", "in_pid", "in_nid", "in_label", "in_w", "in_U", "in_V", "in_b"):      # regenerated from (seed, D) by the reader
                    res.pop(k)
                res["recipe"] = np.array([seed, D])
                save(name, res)
    print("total bytes", total)
    return written


if __name__ == "__main__":
    main()
