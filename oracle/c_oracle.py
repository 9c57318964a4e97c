"""ctypes wrapper + build recipe for oracle/orx_oracle.c (TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__ and bench.py's cpu_baseline leg, never by
the product package)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "orx_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "liborx_oracle.so")

_lib = None


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(LIB)
            and (not os.path.exists(SRC) or os.path.getmtime(LIB) >= os.path.getmtime(SRC))):
        return LIB
    cmd = ["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = ctypes.CDLL(LIB)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        L.orc_pairwise_step.argtypes = [ctypes.c_int, ctypes.c_int, fp, fp, fp, fp, fp, fp,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ip, ip, ip,
                                        ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                        fp, ctypes.POINTER(ctypes.c_double)]
        L.orc_pairwise_step.restype = ctypes.c_int
        L.orc_pairwise_step_adam.argtypes = [ctypes.c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ip, ip, ip, ctypes.c_int64,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                             fp, ctypes.POINTER(ctypes.c_double)]
        L.orc_pairwise_step_adam.restype = ctypes.c_int
        L.orc_censor.argtypes = [fp, ctypes.c_int64, ctypes.c_int, ip, ctypes.c_int64, ctypes.c_float]
        L.orc_censor.restype = ctypes.c_int
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_l2w.argtypes = [ctypes.c_float]
        L.orc_set_l2w.restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


class PairwiseCPU:
    """Holds scratch + Adagrad / Adam slots; tables are caller-owned float32 C-contiguous arrays."""

    def __init__(self, model, opt, U, V, b, lr, eps=1e-7, init_acc=0.1, margin=0.5, beta_1=0.9, beta_2=0.999, l2w=1.0):
        self.l2w = float(l2w)       # weight of l2_loss in the differentiated objective (0: the library's ORX_NO_L2)
        self.model = {"bpr": 0, "ucml": 1}[model]
        self.opt = {"sgd": 0, "adagrad": 1, "adam": 2}[opt]
        self.b1, self.b2, self.t = beta_1, beta_2, 0
        self.U, self.V, self.b = U, V, b.reshape(-1)
        assert U.dtype == np.float32 and U.flags.c_contiguous and V.flags.c_contiguous
        self.lr, self.eps, self.margin = lr, eps, margin
        if self.opt == 1:
            self.accU = np.full_like(U, init_acc)
            self.accV = np.full_like(V, init_acc)
            self.accb = np.full_like(self.b, init_acc)
        else:
            self.accU = self.accV = self.accb = None
        if self.opt == 2:
            self.m = [np.zeros_like(x) for x in (U, V, self.b)]
            self.v = [np.zeros_like(x) for x in (U, V, self.b)]
        self.scratch = None

    def step(self, uid, pid, nid):
        B, D = uid.shape[0], self.U.shape[1]
        need = 3 * B * D + B
        if self.scratch is None or self.scratch.size < need:
            self.scratch = np.empty(need, np.float32)
        out = (ctypes.c_double * 2)()
        lib().orc_set_l2w(self.l2w)
        if self.opt == 2:
            self.t += 1
            lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
            rc = lib().orc_pairwise_step_adam(self.model, _fp(self.U), _fp(self.V), _fp(self.b), _fp(self.m[0]), _fp(self.v[0]),
                                              _fp(self.m[1]), _fp(self.v[1]), _fp(self.m[2]), _fp(self.v[2]),
                                              self.U.shape[0], self.V.shape[0], D, _ip(uid), _ip(pid), _ip(nid), B,
                                              lr_t, self.b1, self.b2, self.eps, self.margin, _fp(self.scratch), out)
            assert rc == 0
            return out[0], out[1]
        rc = lib().orc_pairwise_step(self.model, self.opt, _fp(self.U), _fp(self.V), _fp(self.b),
                                     _fp(self.accU), _fp(self.accV), _fp(self.accb),
                                     self.U.shape[0], self.V.shape[0], D, _ip(uid), _ip(pid), _ip(nid), B,
                                     self.lr, self.eps, self.margin, _fp(self.scratch), out)
        assert rc == 0
        return out[0], out[1]


def censor(W, ids, min_norm=0.1):
    rc = lib().orc_censor(_fp(W), W.shape[0], W.shape[1], _ip(np.ascontiguousarray(ids, np.int32)), len(ids), min_norm)
    assert rc == 0


def num_threads():
    return lib().orc_num_threads()
