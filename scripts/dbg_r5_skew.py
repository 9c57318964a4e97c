"""debug: quiet call -> skewed call (no read-back) -> skewed call, tables against the oracle after EVERY call"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_pairing import _case
from conftest import rel_err
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc

def run(tag):
    U, V, b, uid, pid, nid = _case(9, 30000, 30000, 4096, 64, K=4)
    _, _, _, uz, pz, nz = _case(10, 30000, 30000, 4096, 64, K=4, zipf=1.1)
    ctx = rt.Context(0)
    tU, tV, tb = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b))
    o = rt.Optimizer.sgd(0.05, ctx=ctx)
    Uo, Vo, bo = U.copy(), V.copy(), b.copy(); oo = orc.SGD(lr=0.05)
    f = lambda a: a.reshape(-1)
    for name, (u, p, n) in (("uniform", (uid, pid, nid)), ("zipf-1", (uz, pz, nz)), ("zipf-2", (uz, pz, nz)), ("uniform-2", (uid, pid, nid))):
        l, l2 = rt.pairwise_step("bpr", o, tU, tV, tb, f(u), f(p), f(n), K=4, B=4096)
        lo = [orc.bpr_step(Uo, Vo, bo, u[k], p[k], n[k], oo) for k in range(4)]
        print(tag, name, "nowait_calls", ctx.stat("nowait_calls"), "quiet", ctx.stat("quiet"), "pairs", ctx.stat("pairs"), "max_dup", ctx.stat("max_dup"),
              "| err U %.2e V %.2e b %.2e loss %.2e" % (rel_err(tU.read(), Uo), rel_err(tV.read(), Vo), rel_err(tb.read(), bo),
                                                       rel_err(np.asarray(l, np.float64), np.array([x[0] for x in lo]))), flush=True)

variant = sys.argv[1] if len(sys.argv) > 1 else "default"
run(variant)
