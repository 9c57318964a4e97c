import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from openrec_amd import runtime as rt
NU = NI = 1_000_000; NR = 20_000_000
rng = np.random.default_rng(0)
raw = np.zeros(NR, dtype=[("user_id", np.int32), ("item_id", np.int32)])
raw["user_id"] = rng.integers(0, NU, NR); raw["item_id"] = rng.integers(0, NI, NR)
t0 = time.time(); sm = rt.DeviceSampler(raw, NU, NI); print("index build %.1f s" % (time.time() - t0))
dev = torch.device("cuda", 0); n = 65536 * 200
u, p, g = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
sm.pairwise(1, 0, n, u, p, g); sm.ctx.synchronize()
t0 = time.perf_counter(); sm.pairwise(1, n, n, u, p, g); sm.ctx.synchronize(); dt = time.perf_counter() - t0
print("sampler: %.2f G triplets/s (%.1f us per 65536-triplet batch)" % (n / dt / 1e9, dt / 200 * 1e6))
