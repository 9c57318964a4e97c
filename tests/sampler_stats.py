"""Shared statistics of the pairwise-sampler tests (tests/test_sampler_dist.py on CPU, tests/test_gpu_sampler.py on the
GPU): the distributional contract of the reference's generator, as minted by tests/golden/make_golden_sampler.py."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_hist.npz")


def load():
    g = dict(np.load(GOLDEN))
    raw = np.zeros(len(g["raw_user"]), dtype=[("user_id", np.int32), ("item_id", np.int32)])
    raw["user_id"], raw["item_id"] = g["raw_user"], g["raw_item"]
    return g, raw, int(g["NU"]), int(g["NI"]), int(g["E"])


def histograms(u, p, n, NU, NI):
    pos = np.zeros((NU, NI), np.int64); neg = np.zeros((NU, NI), np.int64)
    np.add.at(pos, (u, p), 1); np.add.at(neg, (u, n), 1)
    return pos, neg


def check_against_golden(pos, neg, g, raw, NU, NI, E):
    """pos / neg: [NU, NI] counts over exactly E epochs of the records"""
    mult = np.zeros((NU, NI), np.int64); np.add.at(mult, (raw["user_id"], raw["item_id"]), 1)
    # positives: shuffle-and-pop emits every record exactly once per epoch (utils.py:82-87) -- exact, not statistical
    assert np.array_equal(pos, E * mult) and np.array_equal(g["pos"], E * mult)
    # negatives are never positives of the user (utils.py:110-113)
    assert (neg[mult > 0] == 0).all() and (g["neg"][mult > 0] == 0).all()
    assert np.array_equal(neg.sum(1), g["neg"].sum(1))                 # one negative per emitted record of the user
    chi2_two = df = chi2_one = 0.0
    for u in range(NU):
        free = mult[u] == 0
        tot = neg[u].sum()
        if tot == 0:
            continue
        a, b = neg[u, free].astype(np.float64), g["neg"][u, free].astype(np.float64)
        chi2_two += ((a - b) ** 2 / np.maximum(a + b, 1)).sum()         # two-sample, equal totals
        exp = tot / free.sum()
        chi2_one += ((a - exp) ** 2 / exp).sum()                        # against the uniform law itself
        df += free.sum() - 1
    z_two = (chi2_two - df) / np.sqrt(2 * df)
    z_one = (chi2_one - df) / np.sqrt(2 * df)
    return z_two, z_one, df


def _z(chi2, df):
    return (chi2 - df) / np.sqrt(2 * df)


def check_stratified(uid, iid, label, g, raw, NU, NI):
    """(user, item, label) of `strat_n` samples of the stratified stream against the reference's (dataset.py:18-36)"""
    n, ratio = int(g["strat_n"]), float(g["strat_ratio"])
    assert len(uid) == n
    mult = np.zeros((NU, NI), np.int64); np.add.at(mult, (raw["user_id"], raw["item_id"]), 1)
    isp = label == 1.0
    assert ((label == 0.0) | isp).all()
    out = {}
    for name, pos, neg in (("got", *_hist2(uid, iid, isp, NU, NI)), ("ref", g["strat_pos"].astype(np.int64), g["strat_neg"].astype(np.int64))):
        P = int(pos.sum())
        out[name + "_z_ratio"] = (P - n * ratio) / np.sqrt(n * ratio * (1 - ratio))            # the coin
        full = P // len(raw)                                                                     # shuffle-and-pop: whole epochs, then a part
        assert ((pos >= full * mult) & (pos <= (full + 1) * mult)).all(), name
        assert (neg[mult > 0] == 0).all(), name                                                  # a negative pair is never a positive
        free = mult == 0
        exp = neg.sum() / free.sum()
        out[name + "_z_uniform"] = _z(((neg[free] - exp) ** 2 / exp).sum(), free.sum() - 1)      # uniform over the non-positive PAIRS
        out[name] = neg
    a, b = out["got"][mult == 0].astype(np.float64), out["ref"][mult == 0].astype(np.float64)
    ka, kb = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
    out["z_two"] = _z(((ka * a - kb * b) ** 2 / np.maximum(a + b, 1)).sum(), (mult == 0).sum() - 1)
    return out


def _hist2(uid, iid, isp, NU, NI):
    pos = np.zeros((NU, NI), np.int64); neg = np.zeros((NU, NI), np.int64)
    np.add.at(pos, (uid[isp], iid[isp]), 1); np.add.at(neg, (uid[~isp], iid[~isp]), 1)
    return pos, neg


def check_per_pos(uid, iid, label, g, raw, NU, NI):
    """whole epochs of the per-positive stream (groups of 1 + nneg) against the reference's (dataset.py:38-58)"""
    E2, ratio = int(g["perpos_epochs"]), float(g["perpos_ratio"])
    nneg = int((1 - ratio) / ratio)
    grp = 1 + nneg
    assert len(uid) == E2 * len(raw) * grp
    mult = np.zeros((NU, NI), np.int64); np.add.at(mult, (raw["user_id"], raw["item_id"]), 1)
    U, I, L = uid.reshape(-1, grp), iid.reshape(-1, grp), label.reshape(-1, grp)
    assert (L[:, 0] == 1.0).all() and (L[:, 1:] == 0.0).all() and (U == U[:, :1]).all()        # a record, then its negatives
    pos = np.zeros((NU, NI), np.int64); np.add.at(pos, (U[:, 0], I[:, 0]), 1)
    assert np.array_equal(pos, E2 * mult) and np.array_equal(g["perpos_pos"], E2 * mult)          # every record once per epoch
    assert (I[:, 1:] != I[:, :1]).all()                                                            # never the record's own item
    srt = np.sort(I[:, 1:], axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()                                                      # random.sample: distinct
    hist = np.zeros((NI, NI), np.int64); np.add.at(hist, (np.repeat(I[:, 0], nneg), I[:, 1:].reshape(-1)), 1)
    ref = g["perpos_neg"].astype(np.int64)
    assert np.array_equal(hist.sum(1), ref.sum(1)) and (np.diag(hist) == 0).all() and (np.diag(ref) == 0).all()
    off = ~np.eye(NI, dtype=bool) & (hist.sum(1) > 0)[:, None]
    a, b = hist[off].astype(np.float64), ref[off].astype(np.float64)
    exp = np.broadcast_to(hist.sum(1, keepdims=True) / (NI - 1.0), hist.shape)[off]
    df = off.sum() - (hist.sum(1) > 0).sum()
    # a group draws its nneg items WITHOUT replacement: a cell counts Binomial(groups, q) hits, q = nneg / (NI - 1), whose
    # variance is (1 - q) times the multinomial one the chi-square statistic is normalised with
    shrink = 1.0 - nneg / (NI - 1.0)
    return (_z(((a - b) ** 2 / np.maximum(a + b, 1)).sum() / shrink, df), _z(((a - exp) ** 2 / exp).sum() / shrink, df),
            _z(((b - exp) ** 2 / exp).sum() / shrink, df))
