import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix=""):
    return sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix))


GOLDEN_TF = os.path.join(GOLDEN, "tf")          # written by tests/golden/make_golden_tf.py --backend tf (real TensorFlow 2.0.1)


def golden_source(name):
    """"tf": the fixture was minted by the reference itself under real TensorFlow (tests/golden/tf/, preferred when present);
    "torch": by the torch-autograd restatement (tests/golden/make_golden*.py)"""
    return "tf" if os.path.exists(os.path.join(GOLDEN_TF, name)) else "torch"


def load_golden(name):
    src = os.path.join(GOLDEN_TF, name) if golden_source(name) == "tf" else os.path.join(GOLDEN, name)
    return dict(np.load(src))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def delta_check(W0, got, want, steps=1, what=""):
    """Parity of an UPDATE, not of a table: `got` and `want` are a table after `steps` train steps from `W0`.

    max|got - want| / max|want| cannot see the loss gradient at BASELINE's batch size: with B = 65536 the BPR
    coefficient is g ~ 7.6e-6, the bias moves by lr*g ~ 3.8e-7 on values of 0.05, i.e. 7.6e-6 relative -- a kernel
    that dropped the loss gradient would pass a 1e-5 bound on the table.  Here every element's CHANGE is compared:

        |d_got - d_want| <= 1e-5 * |d_want| + (steps + 1) * ulp32(max(|want|, |W0|, max|d_want| of the table))

    The second term is what fp32 arithmetic allows: two correct fp32 implementations that differ in the last bit of g
    (exp approximation, FMA contraction, summation order of a duplicated row) round w - lr*grad to neighbouring floats,
    once per step, and the terms of lr*grad they round on the way are as large as the largest update of the table (UCML:
    lr * 2(p - n) ~ 0.01 against an l2 term of the opposite sign).  At C2 a dropped loss gradient is ~100 ulp on a bias
    and 5-50 ulp on a row element, a 5 % error in g 5 ulp on a bias.
    Also returns the projection coefficient <d_got, d_want> / <d_want, d_want> over the touched elements (1 for a
    faithful update; rounding noise averages out over millions of elements, so it resolves 1e-3 errors of scale)."""
    W0 = np.asarray(W0); got = np.asarray(got); want = np.asarray(want)
    assert got.dtype == np.float32 and want.dtype == np.float32 and W0.dtype == np.float32, "delta_check works on fp32 tables"
    d_got = got.astype(np.float64) - W0
    d_want = want.astype(np.float64) - W0
    ulp = np.spacing(np.maximum(np.maximum(np.abs(want), np.abs(W0)), np.float32(np.abs(d_want).max())))
    bound = 1e-5 * np.abs(d_want) + (steps + 1) * ulp.astype(np.float64)
    bad = np.abs(d_got - d_want) > bound
    assert not bad.any(), "%s: %d elements beyond the update bound, worst %.3g x bound; max|d_want| %.3g" % (
        what, int(bad.sum()), float((np.abs(d_got - d_want) / bound).max()), float(np.abs(d_want).max()))
    touched = d_want != 0
    den = float((d_want[touched] ** 2).sum())
    return float((d_got[touched] * d_want[touched]).sum() / den) if den > 0 else 1.0


TOL = 1e-5          # BASELINE.json north_star: fp32 loss / gradients within 1e-5 relative
# Adam (TF-2.0 sparse apply).  The update lr_t * m / (sqrt(v) + eps) has magnitude ~lr in EVERY element,
# whatever |g| is, until |g| drops below eps / sqrt(1 - beta_2) ~ 3e-6, where eps takes the denominator over.  There an
# fp32 rounding delta of the element's summed gradient (half an ulp of a 0.05-sized term: 3e-9; summation order of a
# duplicated row, exp / FMA differences) enters the update as lr_1 * (1 - beta_1) * delta / eps with
# lr_1 = lr * sqrt(1 - beta_2) / (1 - beta_1): for the tests' lr = 2e-3 that is 6.3e-4 * 0.1 * 3e-9 / 1e-7 = 1.9e-6, or
# 3.8e-5 of the tables' 0.05 range -- the worst element of a table sits there, against any oracle (fp64 or fp32).
# TF itself is as far from exact arithmetic.  Everything that is not Adam keeps TOL.
TOL_ADAM = 5e-5


def parse_case(fname):
    model, d, opt, s = fname[:-4].split("_")
    return model, int(d[1:]), opt, int(s[1:])


OPT_KW = {
    "sgd": dict(lr=0.05),
    "adagrad": dict(lr=0.05, initial_accumulator_value=0.1, epsilon=1e-7),
    "adam": dict(lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7),
}
