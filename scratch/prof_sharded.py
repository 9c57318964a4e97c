import sys, time, torch
sys.path.insert(0, '.')
from openrec_amd import sharded
dev = torch.device('cuda', 0)
N = 1_000_000; B = 65536
eng = sharded.ShardedPairwise('bpr', 'sgd', N, N, 64, lr=0.05, rank=0, world=1, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
ids = torch.randint(0, N, (3, 12, B), device=dev, dtype=torch.int32, generator=g)
for s in range(4): eng.step(ids[0, s], ids[1, s], ids[2, s])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for s in range(4, 12): eng.step(ids[0, s], ids[1, s], ids[2, s])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
