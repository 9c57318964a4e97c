"""Randomized differential test of the exact pairwise step against the fp64 NumPy oracle (GPU box):
python scratch/fuzz_pairwise.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc


OPTS = os.environ.get("FUZZ_OPTS", "sgd,adagrad,adam").split(",")


def case(rng, forced=None):
    c = dict(model=rng.choice(["bpr", "ucml"]), opt=rng.choice(OPTS), D=int(rng.choice([16, 32, 64, 128, 256, 50, 20])),
             NU=int(rng.choice([50, 700, 5000, 40000, 300000])), NI=int(rng.choice([30, 900, 6000, 30000, 500000])),
             B=int(rng.choice([1, 7, 256, 1000, 4096, 8191, 20000])), K=int(rng.choice([1, 2, 3, 5, 9])),
             skew=rng.choice(["uniform", "zipf", "one_hot_item", "one_hot_user", "few"]), censor=bool(rng.random() < 0.2))
    if forced:
        c.update(forced)
    if c["model"] == "bpr":
        c["censor"] = False
    if c["opt"] == "adam" and c["K"] > 130:      # 300 Adam steps amplify fp32-vs-fp64 rounding to 5e-3 (sweep and lazy form alike)
        c["K"] = 130
    return c


def run(c, rng):
    NU, NI, B, K, D = c["NU"], c["NI"], c["B"], c["K"], c["D"]
    scale = 25 if c["censor"] else 1
    U = (rng.uniform(-.05, .05, (NU, D)) * scale).astype(np.float32); V = (rng.uniform(-.05, .05, (NI, D)) * scale).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32)
    if c["skew"] == "zipf":
        w = 1.0 / np.arange(1, NI + 1) ** 1.1; cdf = np.cumsum(w / w.sum()); perm = rng.permutation(NI)
        draw = lambda: perm[np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1)].astype(np.int32)
        pid, nid = draw(), draw()
    elif c["skew"] == "one_hot_item":
        pid = np.full((K, B), int(rng.integers(0, NI)), np.int32); nid = rng.integers(0, NI, (K, B)).astype(np.int32)
        nid[:, ::2] = pid[:, ::2]
    elif c["skew"] == "few":
        pid = rng.integers(0, min(NI, 5), (K, B)).astype(np.int32); nid = rng.integers(0, min(NI, 5), (K, B)).astype(np.int32)
        uid = rng.integers(0, min(NU, 3), (K, B)).astype(np.int32)
    else:
        pid = rng.integers(0, NI, (K, B)).astype(np.int32); nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    if c["skew"] == "one_hot_user":
        uid[:, : max(1, B // 2)] = int(rng.integers(0, NU))
    lr = 0.001 if c["skew"] != "uniform" else 0.02
    if c["opt"] == "adam":
        lr = 0.0005                                  # Adam moves every weight by ~lr per step whatever the gradient
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    opt = {"sgd": lambda: rt.Optimizer.sgd(lr), "adagrad": lambda: rt.Optimizer.adagrad(lr), "adam": lambda: rt.Optimizer.adam(lr)}[c["opt"]]()
    oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr), "adam": lambda: orc.AdamTFSparse(lr)}[c["opt"]]()
    loss, l2 = rt.pairwise_step(c["model"], opt, tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=c["censor"])
    U, V, b = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
    worst = 0.0
    for s in range(K):
        if c["model"] == "bpr":
            lw, l2w = orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)
        else:
            lw, l2w = orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=c["censor"])
        worst = max(worst, abs(loss[s] - lw) / max(abs(lw), 1e-30), abs(l2[s] - l2w) / max(abs(l2w), 1e-30))
    for got, want in ((tU.read(), U), (tV.read(), V), (tb.read(), b)):
        worst = max(worst, np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
    if max(np.abs(U).max(), np.abs(V).max()) > 50 * scale:      # the run diverged (lr x duplicates > 1): rounding is amplified without bound
        return -worst
    return worst


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    forced = [dict(NI=30000, NU=30000, B=65536, K=2, D=64, skew="uniform"),                  # > 6656 tri rows per range: global counters
              dict(NI=5000, NU=5000, B=65536, K=2, D=64, skew="one_hot_item", model="bpr"),  # > 65 k references on one row: 3 tree levels
              dict(NI=2000, NU=2000, B=1000, K=300, D=16, skew="zipf"),                      # more steps than one chunk
              dict(NI=500000, NU=300000, B=20000, K=3, D=128, skew="zipf", model="ucml", censor=True)]
    bad = 0
    t0 = time.time()
    for k in range(n):
        c = case(rng, forced[k] if k < len(forced) else None)
        try:
            w = run(c, rng)
        except Exception as e:                       # noqa: BLE001
            print("EXC", c, repr(e)); bad += 1; continue
        if w < 0:
            print(f"diverged (err {-w:.1e}) {c}"); continue
        flag = "" if w < 5e-5 else "   <-- FAIL"
        if flag or k < len(forced):
            print(f"{w:.2e} {c}{flag}")
        bad += bool(flag)
    print(f"{n} cases, {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
