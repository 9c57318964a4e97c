import os, sys, numpy as np
sys.path.insert(0, '/root/repo/scratch'); sys.path.insert(0, '/root/repo')
import fuzz_pairwise as fz
for K in (60, 130, 300):
    for form in ("lazy", "dense"):
        os.environ.pop("ORX_ADAM_DENSE", None)
        if form == "dense": os.environ["ORX_ADAM_DENSE"] = "1"
        rng = np.random.default_rng(3)
        c = fz.case(rng, dict(NI=2000, NU=2000, B=1000, K=K, D=16, skew="zipf", model="bpr", opt="adam", censor=False))
        print(K, form, fz.run(c, np.random.default_rng(7)))
