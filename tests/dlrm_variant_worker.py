"""Worker of test_gpu_dlrm.py::test_dlrm_fp16_staging_forms_are_bit_identical: two fp16-mode DLRM steps at shapes that take every tile
configuration of kernels_gemm16.hip (one process per environment variant: the switches are read once per process); prints one digest of
all parameters."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openrec_amd import runtime as rt  # noqa: E402


def main():
    B = 4096
    ln_emb = [1000, 37, 5000, 211, 64, 3000, 9, 700]
    cfg = dict(m_spa=64, ln_emb=ln_emb, ln_bot=[512, 256, 64], ln_top=[1024, 1024, 512, 256, 1], dense_dim=13, reference_compat=False)
    m = rt.DLRMModel(fp16_mlp=True, seed=3, **cfg)
    opt = rt.Optimizer.sgd(0.05)
    rng = np.random.default_rng(17)
    h = hashlib.sha256()
    for step in range(2):
        Bs = B if step == 0 else B - 300                     # (a ragged batch: partial row blocks)
        dense = np.log1p(rng.integers(0, 100, (Bs, 13)).astype(np.float32))
        sparse = np.stack([rng.integers(0, r, Bs) for r in ln_emb], 1).astype(np.int32)
        label = (rng.uniform(size=Bs) < 0.3).astype(np.float32)
        loss = m.step(opt, dense, sparse, label)
        h.update(np.asarray(loss, np.float32).tobytes())
    hp = hashlib.sha256()                                     # the parameters alone (a variant that sums the LOSS in another order keeps these)
    for hh in (h, hp):
        hh.update(m.param("emb").read().tobytes())
        for nm, n in (("bot", 3), ("top", 5)):
            for l in range(n):
                hh.update(m.param(nm + "_w", l).read().tobytes()); hh.update(m.param(nm + "_b", l).read().tobytes())
    print("PARAMS", hp.hexdigest())
    print("DIGEST", h.hexdigest())


if __name__ == "__main__":
    main()
