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


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def parse_case(fname):
    model, d, opt, s = fname[:-4].split("_")
    return model, int(d[1:]), opt, int(s[1:])


OPT_KW = {
    "sgd": dict(lr=0.05),
    "adagrad": dict(lr=0.05, initial_accumulator_value=0.1, epsilon=1e-7),
    "adam": dict(lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7),
}
