import sys, time, numpy as np
sys.path.insert(0, '.')
from openrec_amd import runtime as rt
import torch
N = 1_000_000; D = int(sys.argv[1]) if len(sys.argv) > 1 else 64; B = 65536; K = 100
ctx = rt.default_context()
U = rt.Table(N, D).init_uniform(seed=0); V = rt.Table(N, D).init_uniform(seed=1); b = rt.Table(N, 1).init_uniform(seed=2)
g = torch.Generator(device='cuda'); g.manual_seed(1234)
ids = torch.randint(0, N, (3, K, B), device='cuda', dtype=torch.int32, generator=g)
torch.cuda.synchronize()
bytes_per = 24 * D + 28
for model in ('bpr', 'ucml'):
  for optk, hog in (('sgd', False), ('sgd', True), ('adagrad', False)):
    opt = rt.Optimizer.sgd(0.05) if optk == 'sgd' else rt.Optimizer.adagrad(0.05)
    rt.pairwise_step(model, opt, U, V, b, ids[0], ids[1], ids[2], K=20, B=B, hogwild=hog, want_loss=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    rt.pairwise_step(model, opt, U, V, b, ids[0], ids[1], ids[2], K=K, B=B, hogwild=hog, want_loss=False)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.prof_reset(); ctx.prof_enable(True)
    l, l2 = rt.pairwise_step(model, opt, U, V, b, ids[0], ids[1], ids[2], K=K, B=B, hogwild=hog)
    ctx.prof_enable(False)
    pr = ctx.prof_get()
    bp = bytes_per if optk == 'sgd' else 48 * D + 44
    s = f"{model} {optk} hog={hog} D={D}: {K*B/dt/1e9:.3f} G trip/s, {dt/K*1e6:.1f} us/step, alg {K*B*bp/dt/1e12:.2f} TB/s | loss {l[0]:.5f}->{l[-1]:.5f} |"
    for k, v in pr.items():
      if v['launches']: s += f" {k}: {v['total_ms']/v['launches']*1e3:.1f}us"
    fused = pr['fused']; 
    s += f" | fused alg {B*bp/(fused['total_ms']/fused['launches']*1e-3)/1e12:.2f} TB/s"
    print(s, flush=True)
