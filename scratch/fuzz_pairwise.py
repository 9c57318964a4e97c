"""Randomized differential test of the exact pairwise step against the fp64 NumPy oracle, open-ended (GPU box):

    python scratch/fuzz_pairwise.py [n_cases] [seed]            FUZZ_OPTS=adam restricts the optimizers

The case generator and the runner are tests/test_gpu_fuzz.py's (whose seeded 34-case subset the driver runs)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import FORCED, make_case, run_case      # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    opts = os.environ.get("FUZZ_OPTS")
    rng = np.random.default_rng(seed)
    bad, t0 = 0, time.time()
    for k in range(n):
        c = make_case(rng, FORCED[k] if k < len(FORCED) else None)
        if opts and k >= len(FORCED):
            c["opt"] = str(rng.choice(opts.split(",")))
        try:
            w, diverged = run_case(c, rng)
        except Exception as e:                       # noqa: BLE001
            print("EXC", c, repr(e)); bad += 1; continue
        if diverged:
            print(f"diverged (err {w:.1e}) {c}"); continue
        flag = "" if w < 5e-5 else "   <-- FAIL"
        if flag or k < len(FORCED):
            print(f"{w:.2e} {c}{flag}")
        bad += bool(flag)
    print(f"{n} cases, {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
