"""Stress the in-launch duplicate apply / ready-flag hand-off: small tables (almost every row is
duplicated and urgent every step), many steps, compared step by step with the fp64 oracle."""
import sys, numpy as np
sys.path.insert(0, '.')
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
worst = 0
for trial, (NU, NI, B, K, D, optk) in enumerate([(3000, 3000, 8192, 48, 64, 'sgd'), (20000, 30000, 65536, 24, 64, 'sgd'),
                                               (2000, 2500, 4096, 40, 128, 'adagrad'), (50000, 50000, 65536, 16, 32, 'sgd'),
                                               (1000, 1000, 16384, 32, 64, 'adagrad')]):
    rng = np.random.default_rng(trial)
    U = rng.uniform(-.05, .05, (NU, D)); V = rng.uniform(-.05, .05, (NI, D)); b = rng.uniform(-.05, .05, (NI, 1))
    U32, V32, b32 = U.astype(np.float32), V.astype(np.float32), b.astype(np.float32)
    U, V, b = U32.astype(np.float64), V32.astype(np.float64), b32.astype(np.float64)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32); nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    lr = 0.02 / (B / min(NU, NI))            # keep lr * multiplicity < 1 so errors do not amplify
    tU = rt.Table(NU, D).write(U32); tV = rt.Table(NI, D).write(V32); tb = rt.Table(NI, 1).write(b32)
    opt = rt.Optimizer.sgd(lr) if optk == 'sgd' else rt.Optimizer.adagrad(lr, 0.1, 1e-7)
    oo = orc.SGD(lr) if optk == 'sgd' else orc.Adagrad(lr, 0.1, 1e-7)
    for rep in range(3):                     # three calls: also exercises the call boundary
        loss, l2 = rt.pairwise_step('bpr', opt, tU, tV, tb, uid, pid, nid, K=K, B=B)
        for s in range(K):
            lr_, l2r = orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)
            e = abs(loss[s] - lr_) / abs(lr_); worst = max(worst, e)
            assert e < 2e-5, (trial, rep, s, loss[s], lr_)
        eu = np.abs(tU.read() - U).max() / np.abs(U).max(); ev = np.abs(tV.read() - V).max() / np.abs(V).max()
        eb = np.abs(tb.read() - b).max() / np.abs(b).max()
        worst = max(worst, eu, ev, eb)
        assert max(eu, ev, eb) < 5e-5, (trial, rep, eu, ev, eb)
    print('trial', trial, (NU, NI, B, K, D, optk), 'ok  max rel err so far %.2e' % worst, flush=True)
print('STRESS OK', worst)
