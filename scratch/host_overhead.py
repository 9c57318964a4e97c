"""Where the host time of a K-step call goes (bench.py's protocol): python-side pieces around orx_pairwise_step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openrec_amd import runtime as rt
dev = torch.device("cuda", 0)
ctx = rt.Context(0)
N, D, B, K, W = 1_000_000, 64, 65536, 20, 5
U = rt.Table(N, D, ctx).init_uniform(seed=0); V = rt.Table(N, D, ctx).init_uniform(seed=1); b = rt.Table(N, 1, ctx).init_uniform(seed=2)
opt = rt.Optimizer.sgd(0.05, ctx=ctx)
g = torch.Generator(device=dev); g.manual_seed(1)
uid, pid, nid = (torch.randint(0, N, (K + W, B), device=dev, dtype=torch.int32, generator=g) for _ in range(3))
rt.pairwise_reserve(opt, U, V, b, K, B)
WK = int(os.environ.get("WARM_K", W))
for _ in range(int(os.environ.get("WARM_REPS", 1))):
    rt.pairwise_step("bpr", opt, U, V, b, uid[:WK], pid[:WK], nid[:WK], K=WK, B=B, want_loss=False)
ctx.synchronize(); torch.cuda.synchronize()
if os.environ.get("SPIN_MS"):
    x = torch.empty(64 << 20, device=dev); y = torch.empty_like(x)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < float(os.environ["SPIN_MS"]):
        y.copy_(x)
    torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    a, bb, c = uid[W:W + K], pid[W:W + K], nid[W:W + K]
    t1 = time.perf_counter()
    rt.pairwise_step("bpr", opt, U, V, b, a, bb, c, K=K, B=B, want_loss=False)
    t2 = time.perf_counter()
    ctx.synchronize()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("slice %.0f  call %.0f  ctx.sync %.0f  torch.sync %.0f  total %.0f us  (%.2f us/step)" % ((t1-t0)*1e6, (t2-t1)*1e6, (t3-t2)*1e6, (t4-t3)*1e6, (t4-t0)*1e6, (t4-t0)*1e6/K))
