"""where does the fp16-mode top_w0 error of the compat=True 'wide' / 'thinbot' cases sit?"""
import copy, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from openrec_amd import runtime as rt
from oracle import numpy_oracle as orc
from oracle.dlrm_oracle import DLRMOracle
from dlrm_util import draw_batch, load_model, params_of, round_to_fp32, snapshot

for name, cfg in (("wide", dict(m_spa=32, ln_bot=[512, 256, 32], ln_top=[1024, 512, 256, 1], B=2304, dense_dim=13)),
                  ("thinbot", dict(m_spa=64, ln_bot=[64, 16, 64], ln_top=[128, 24, 64, 1], B=600, dense_dim=13))):
    rng = np.random.default_rng(11)
    ln_emb = [50, 300, 7, 1000, 33]
    B = cfg.pop("B"); cfg.update(ln_emb=ln_emb)
    o = DLRMOracle(dtype=np.float64, operand_dtype=np.float16, seed=5, reference_compat=True, **cfg)
    for W, b in o.bot + o.top:
        b[:] = rng.normal(size=b.shape) * 0.1
    round_to_fp32(o)
    o0 = copy.deepcopy(o)
    bt = draw_batch(o, rng, B, ln_emb, label_p=0.3, dense_dim=13)
    o.step(*bt, orc.SGD(0.02))
    m = rt.DLRMModel(reference_compat=True, fp16_mlp=True, **cfg)
    load_model(m, o0)
    m.step(rt.Optimizer.sgd(0.02), *bt)
    got = snapshot(m, o)
    for k in ("top_w0", "top_b0", "top_w1", "bot_w0"):
        w0 = params_of(o0)[k].astype(np.float32).astype(np.float64)
        dg, dw = got[k] - w0, params_of(o)[k] - w0
        E = np.abs(dg - dw)
        r, c = np.unravel_index(E.argmax(), E.shape)
        print(name, k, "shape", E.shape, "max err", E.max() / np.abs(dw).max(), "at", (r, c), "dg", dg[r, c], "dw", dw[r, c], "ulp(w)", np.spacing(np.float32(abs(w0[r, c]))))
        rowmax = E.max(axis=1) / np.abs(dw).max(); colmax = E.max(axis=0) / np.abs(dw).max()
        print("   rows > 1e-4:", np.flatnonzero(rowmax > 1e-4)[:20], "cols > 1e-4:", np.flatnonzero(colmax > 1e-4)[:20], "n", (colmax > 1e-4).sum())
        print("   |w0| at worst", abs(w0[r, c]), "max|w0|", np.abs(w0).max(), "max|dw|", np.abs(dw).max())
