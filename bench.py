#!/usr/bin/env python3
"""Headline benchmark: BPR training triplets/s at dim=64 on a synthetic
1M-user x 1M-item table (BASELINE.json configs[1]), MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one fused train step (forward on the pre-step tables, gradients of
loss + l2_loss, exact TF sparse SGD apply) over one batch of 65536 triplets
per GPU; ids are synthetic and already resident in HBM when the timed region
starts.  N = 1: tables live on one GPU.  N > 1: tables are row-sharded across
the N ranks (row r on rank r % N) and every step exchanges triplets, item rows
and item-row gradients with RCCL all-to-all (openrec_amd/sharded.py);
per-GPU batch is fixed => "scaling": "weak".

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def alg_bytes_per_triplet(dim, opt, model="bpr"):
    # SGD 24*D + 28, Adagrad 48*D + 44 bytes per triplet; GMF / WRMF 16*D + 20 per (user, item, label)
    if model in ("gmf", "wrmf"):
        return 16 * dim + 20 if opt == "sgd" else 32 * dim + 28
    # (TF-2.0 sparse Adam sweeps whole tables: its step is not priced per triplet; the Adagrad figure is reported)
    return 24 * dim + 28 if opt == "sgd" else 48 * dim + 44


def make_ids(torch, n_users, n_items, steps, batch, seed, device, zipf=0.0):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    uid = torch.randint(0, n_users, (steps, batch), device=device, dtype=torch.int32, generator=g)
    if zipf > 0:        # secondary workload: item popularity ~ Zipf(alpha), exposes duplicate handling
        w = 1.0 / torch.arange(1, n_items + 1, device=device, dtype=torch.float64) ** zipf
        cdf = torch.cumsum(w / w.sum(), 0)
        draw = lambda: torch.searchsorted(cdf, torch.rand((steps, batch), device=device, dtype=torch.float64, generator=g)).clamp_(max=n_items - 1).to(torch.int32)
        pid, nid = draw(), draw()
        return uid.contiguous(), pid.contiguous(), nid.contiguous()
    pid = torch.randint(0, n_items, (steps, batch), device=device, dtype=torch.int32, generator=g)
    nid = torch.randint(0, n_items, (steps, batch), device=device, dtype=torch.int32, generator=g)
    for _ in range(4):                      # resample negatives that collide with the positive
        clash = nid == pid
        if not bool(clash.any()):
            break
        fresh = torch.randint(0, n_items, (steps, batch), device=device, dtype=torch.int32, generator=g)
        nid = torch.where(clash, fresh, nid)
    return uid.contiguous(), pid.contiguous(), nid.contiguous()


def cpu_baseline(args, budget_s=15.0, max_steps=2000):
    """The C/OpenMP oracle port (oracle/orx_oracle.c) on the host cores: same
    table shapes, same batch size, a bounded number of steps."""
    from oracle import c_oracle
    rng = np.random.default_rng(0)
    U = rng.uniform(-0.05, 0.05, (args.users, args.dim)).astype(np.float32)
    V = rng.uniform(-0.05, 0.05, (args.items, args.dim)).astype(np.float32)
    b = rng.uniform(-0.05, 0.05, (args.items, 1)).astype(np.float32)
    cpu, cores, what = c_oracle.PairwiseCPU(args.model, args.opt, U, V, b, lr=0.05), c_oracle.num_threads(), "oracle/orx_oracle.c (OpenMP)"
    def draw():       # the same id law as make_ids: uniform users / items, negatives that collide with the positive are redrawn
        u, p, n = (rng.integers(0, hi, args.batch).astype(np.int32) for hi in (args.users, args.items, args.items))
        for _ in range(4):
            clash = n == p
            if not clash.any():
                break
            n = np.where(clash, rng.integers(0, args.items, args.batch).astype(np.int32), n)
        return u, p, n
    ids = [draw() for _ in range(32)]
    for s in range(2):
        cpu.step(*ids[s])
    t0 = time.perf_counter()
    done = 0
    while done < max_steps and time.perf_counter() - t0 < budget_s:      # bounded sample (~15 s)
        cpu.step(*ids[done % len(ids)])
        done += 1
    dt = time.perf_counter() - t0
    return dict(value=done * args.batch / dt, unit="triplets/s", cores=cores, kind="port",
                sample=f"{done} steps of {args.batch} triplets on {args.users}x{args.items}x{args.dim} tables, {what}, {dt:.1f} s")


CRITEO_KAGGLE_COUNTS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10,
                        5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def bench_dlrm(args, torch, rt, world, rank, local_rank, device, dist):
    """Secondary workload (C5): DLRM with the Criteo-Kaggle cardinalities, dim 128, bottom 13-512-256-128,
    top 479-1024-1024-512-256-1, B samples per GPU per step, SGD.  N = 1: orx_dlrm_step on batches resident
    in HBM; N > 1: embedding tables row-sharded, MLPs data-parallel (openrec_amd/sharded_dlrm.py)."""
    K, W, B = args.steps, args.warmup, args.batch
    cfg = dict(m_spa=128, ln_emb=CRITEO_KAGGLE_COUNTS, ln_bot=[512, 256, 128], ln_top=[1024, 1024, 512, 256, 1], dense_dim=13,
               reference_compat=False)
    extra = {}
    g = torch.Generator(device=device); g.manual_seed(99 + rank)
    n = (K + W) * B
    dense = torch.log1p(torch.randint(0, 100, (n, 13), device=device, generator=g).float()).contiguous()
    sparse = torch.stack([torch.randint(0, c, (n,), device=device, generator=g, dtype=torch.int32) for c in CRITEO_KAGGLE_COUNTS], 1).contiguous()
    label = (torch.rand((n,), device=device, generator=g) < 0.25).float().contiguous()
    if world == 1 and not args.sharded:
        ctx = rt.Context(local_rank)
        m = rt.DLRMModel(ctx=ctx, fp16_mlp=args.fp16_mlp, **cfg)
        opt = {"sgd": lambda: rt.Optimizer.sgd(0.01, ctx=ctx), "adagrad": lambda: rt.Optimizer.adagrad(0.01, 0.1, 1e-7, ctx=ctx),
               "adam": lambda: rt.Optimizer.adam(0.001, ctx=ctx)}[args.opt]()
        es = lambda t: t.element_size()
        run = lambda first, count: m.step_device(opt, dense.data_ptr() + first * B * 13 * es(dense), sparse.data_ptr() + first * B * 26 * es(sparse),
                                                 label.data_ptr() + first * B * es(label), count, B)
        torch.cuda.synchronize()
        if W:                                            # two calls (see the pairwise path below: a process's second call is slower)
            run(0, W - W // 2)
            if W // 2:
                run(W - W // 2, W // 2)
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(W, K)
        ctx.synchronize(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        par = "single-gpu"
        # MLP products against the MFMA roofline: 3 products per Dense layer and step (forward, dX, dW;
        # no dX for the first bottom layer), 2*B*in*out flops each; kernel time from dispatch-attached events
        ctx.prof_reset(); ctx.prof_enable(True)
        run(W, K)
        ctx.prof_enable(False)
        gp = ctx.prof_get().get("gemm", {})
        dims = list(zip([13] + cfg["ln_bot"][:-1], cfg["ln_bot"])) + list(zip([128 + 27 * 26 // 2] + cfg["ln_top"][:-1], cfg["ln_top"]))
        flops = sum(2.0 * B * i * o * (2 if k == 0 else 3) for k, (i, o) in enumerate(dims))
        if gp.get("launches"):
            tf = flops * K / (gp["total_ms"] * 1e-3) / 1e12
            peak = 2500.0 if args.fp16_mlp else 157.3         # MI355X_MICROARCH.md: dense fp16 MFMA / fp32 (xf32-less) matrix peak, TFLOP/s
            extra = {"roofline": {"bound": "mfma", "kernel": ("gemm16_nt_dma_kernel + gemm16_tn_dma_kernel + slab_reduce_kernel + head kernels (all MLP products of the step)"
                                                              if args.fp16_mlp else "gemm_f32v_kernel"), "achieved": tf,
                                  "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
                                  "gemm_ms_per_step": gp["total_ms"] / K, "flops_per_step": flops}}
    else:
        from openrec_amd.sharded_dlrm import ShardedDLRM
        eng = ShardedDLRM(rank=rank, world=world, device=device, opt=args.opt, lr=0.001 if args.opt == "adam" else 0.01, seed=0, fp16_mlp=args.fp16_mlp, **cfg)
        eng.force_collectives = dist is not None
        # K steps per host call: the library's engine (orx_sharded_dlrm_steps) where it applies, the per-phase path otherwise
        d3, s3, l3 = dense.view(K + W, B, 13), sparse.view(K + W, B, 26), label.view(K + W, B)
        if W:
            eng.steps(d3[:W], s3[:W], l3[:W])
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.steps(d3[W:], s3[W:], l3[W:])
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        eng.check()
        par = f"embedding tables row-sharded x{world} (all-to-all), MLPs data-parallel (all-reduce); " + \
              ("K-step engine inside the library" if eng._comm is not None else "per-phase path over torch.distributed")
    if rank == 0:
        print(json.dumps({
            **(extra if world == 1 and not args.sharded else {}),
            "metric": "DLRM training samples/sec (Criteo-Kaggle cardinalities, dim 128)", "value": K * B * world / dt,
            "unit": "samples/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 MFMA MLP products, f32 elsewhere" if args.fp16_mlp else "f32",
            "data": "synthetic",
            "config": {"workload": f"dlrm 26 tables (33.8 M rows x 128), bottom 13-512-256-128, top 479-1024-1024-512-256-1, "
                                   f"batch={B} samples/GPU, {args.opt} lr={0.001 if args.opt == 'adam' else 0.01}, mse loss", "parallelism": par}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def secondary_pairwise(torch, rt, ctx, device, model, dim, opt_name, censor, K, W, batch=65536, users=1_000_000, items=1_000_000):
    """one more pairwise workload after the headline run (same protocol: W warm-up steps in two calls, K timed steps in one call,
    then the same K steps again with dispatch-attached events for the kernel time)"""
    lr = 0.05
    U = rt.Table(users, dim, ctx).init_uniform(seed=0); V = rt.Table(items, dim, ctx).init_uniform(seed=1); b = rt.Table(items, 1, ctx).init_uniform(seed=2)
    opt = {"sgd": lambda: rt.Optimizer.sgd(lr, ctx=ctx), "adagrad": lambda: rt.Optimizer.adagrad(lr, ctx=ctx)}[opt_name]()
    uid, pid, nid = make_ids(torch, users, items, K + W, batch, 4321, device)
    torch.cuda.synchronize()
    run = lambda first, count: rt.pairwise_step(model, opt, U, V, b, uid[first:first + count], pid[first:first + count], nid[first:first + count],
                                                K=count, B=batch, margin=0.5, want_loss=False, censor=censor)
    rt.pairwise_reserve(opt, U, V, b, max(K, W, 1), batch)
    w1 = W - W // 2
    run(0, w1)
    if W // 2:
        run(w1, W // 2)
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, K)
    ctx.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.prof_reset(); ctx.prof_enable(True)
    run(W, K)
    ctx.prof_enable(False)
    fused = ctx.prof_get().get("fused", {})
    bpt = alg_bytes_per_triplet(dim, opt_name, model)
    out = {"workload": f"{model} dim={dim} {users}x{items} table, batch={batch}, {opt_name}, exact TF duplicate semantics"
                       f"{', censor after each step' if censor else ''}",
           "value": K * batch / dt, "unit": "triplets/s", "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3}
    if fused.get("launches"):
        dur = fused["total_ms"] / fused["launches"] * 1e-3
        out["roofline"] = {"bound": "hbm", "kernel": "fused_kernel", "kernel_us": dur * 1e6, "bytes_per_triplet": bpt,
                           "achieved": batch * bpt / dur / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": batch * bpt / dur / 1e9 / HBM_PEAK_GBS}
    return out


def secondary_dlrm(torch, rt, ctx, device, fp16, K, W, B=8192):
    """C5 on one GPU after the headline run: orx_dlrm_step on batches resident in HBM (see bench_dlrm for the full version)"""
    cfg = dict(m_spa=128, ln_emb=CRITEO_KAGGLE_COUNTS, ln_bot=[512, 256, 128], ln_top=[1024, 1024, 512, 256, 1], dense_dim=13, reference_compat=False)
    g = torch.Generator(device=device); g.manual_seed(99)
    n = (K + W) * B
    dense = torch.log1p(torch.randint(0, 100, (n, 13), device=device, generator=g).float()).contiguous()
    sparse = torch.stack([torch.randint(0, c, (n,), device=device, generator=g, dtype=torch.int32) for c in CRITEO_KAGGLE_COUNTS], 1).contiguous()
    label = (torch.rand((n,), device=device, generator=g) < 0.25).float().contiguous()
    m = rt.DLRMModel(ctx=ctx, fp16_mlp=fp16, **cfg)
    opt = rt.Optimizer.sgd(0.01, ctx=ctx)
    es = lambda t: t.element_size()
    run = lambda first, count: m.step_device(opt, dense.data_ptr() + first * B * 13 * es(dense), sparse.data_ptr() + first * B * 26 * es(sparse),
                                             label.data_ptr() + first * B * es(label), count, B)
    torch.cuda.synchronize()
    run(0, W - W // 2)
    if W // 2:
        run(W - W // 2, W // 2)
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(W, K)
    ctx.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.prof_reset(); ctx.prof_enable(True)
    run(W, K)
    ctx.prof_enable(False)
    gp = ctx.prof_get().get("gemm", {})
    dims = list(zip([13] + cfg["ln_bot"][:-1], cfg["ln_bot"])) + list(zip([128 + 27 * 26 // 2] + cfg["ln_top"][:-1], cfg["ln_top"]))
    flops = sum(2.0 * B * i * o * (2 if k == 0 else 3) for k, (i, o) in enumerate(dims))
    out = {"workload": f"dlrm 26 Criteo-Kaggle tables (33.8 M rows x 128), bottom 13-512-256-128, top 479-1024-1024-512-256-1, batch={B}, sgd, "
                       f"{'fp16 MFMA MLP products (fp32 master weights, accumulation, embeddings)' if fp16 else 'fp32'}",
           "value": K * B / dt, "unit": "samples/s", "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3}
    if gp.get("launches"):
        tf = flops * K / (gp["total_ms"] * 1e-3) / 1e12
        peak = 2500.0 if fp16 else 157.3
        out["roofline"] = {"bound": "mfma", "kernel": "all MLP products of the step (gemm16_nt_dma / gemm16_tn_dma / slab_reduce / head kernels)" if fp16 else "gemm_f32v_kernel",
                           "kernel_us": gp["total_ms"] / K * 1e3, "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "flops_per_step": flops}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--fp16-mlp", action="store_true", help="dlrm: MLP products on fp16 MFMA (performance mode)")
    ap.add_argument("--model", default="bpr", choices=["bpr", "ucml", "gmf", "wrmf", "dlrm"],
                    help="gmf / wrmf: the pointwise step over B (user, item, label) samples (secondary workloads)")
    ap.add_argument("--opt", default="sgd", choices=["sgd", "adagrad", "adam"])
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=65536, help="triplets per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hogwild", action="store_true", help="racy non-reference mode (never the headline)")
    ap.add_argument("--sharded", action="store_true", help="use the row-sharded engine even on one GPU (debug)")
    ap.add_argument("--shard-dedup", choices=["auto", "on", "off"], default="auto",
                    help="per-destination dedup of the sharded engine's item requests (auto: by list length vs items)")
    ap.add_argument("--zipf", type=float, default=0.0, help="item ids ~ Zipf(alpha) instead of uniform (secondary workload)")
    ap.add_argument("--hot-items", type=int, default=0,
                    help="sharded engine: replicate the H most popular items on every rank (SURVEY.md D.3; use with --zipf)")
    ap.add_argument("--host-ids", action="store_true",
                    help="hand the ids over as host buffers (the C ABI stages them over PCIe inside the timed call); "
                         "reported for DESIGN.md, never the headline")
    ap.add_argument("--censor", action="store_true",
                    help="UCML: LatentFactor.censor of the touched rows after every step (ucml.py:44-48)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the \"secondary\" block of the headline line (C3 = UCML D=128 + censor, C5 = DLRM with fp16 MLP products)")
    args = ap.parse_args()

    import torch
    from openrec_amd import runtime as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or (args.sharded and "RANK" in os.environ):     # torchrun with one rank: exercises the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    K, W = args.steps, args.warmup
    lr = 0.05
    if args.model == "dlrm":
        if args.batch == 65536:
            args.batch = 8192
        return bench_dlrm(args, torch, rt, world, rank, local_rank, device, dist)
    if world == 1 and not args.sharded:
        ctx = rt.Context(local_rank)
        U = rt.Table(args.users, args.dim, ctx).init_uniform(seed=0)
        V = rt.Table(args.items, args.dim, ctx).init_uniform(seed=1)
        b = rt.Table(args.items, 1, ctx).init_uniform(seed=2)
        opt = {"sgd": lambda: rt.Optimizer.sgd(lr, ctx=ctx), "adagrad": lambda: rt.Optimizer.adagrad(lr, ctx=ctx),
               "adam": lambda: rt.Optimizer.adam(0.001, ctx=ctx)}[args.opt]()
        uid, pid, nid = make_ids(torch, args.users, args.items, K + W, args.batch, 1234, device, args.zipf)
        torch.cuda.synchronize()
        if args.host_ids:
            uid, pid, nid = (x.cpu().numpy() for x in (uid, pid, nid))
        pointwise = args.model in ("gmf", "wrmf")
        if pointwise:
            g = torch.Generator(device=device); g.manual_seed(7)
            label = (torch.rand((K + W, args.batch), device=device, generator=g) < 0.5).to(torch.float32).contiguous()
            wk = rt.Table(args.dim, 1, ctx).init_uniform(seed=3) if args.model == "gmf" else None

        views = {}

        def run(first, count, want_loss=False):
            # (the views of the id arrays are made once per range: slicing a torch tensor is host time, not part of a step)
            if (first, count) not in views:
                views[(first, count)] = tuple(x[first:first + count] for x in ((uid, pid, label) if pointwise else (uid, pid, nid)))
            v = views[(first, count)]
            if pointwise:
                return rt.pointwise_step(args.model, opt, U, V, b, wk, v[0], v[1], v[2], K=count, B=args.batch, hogwild=args.hogwild,
                                         want_loss=want_loss)
            return rt.pairwise_step(args.model, opt, U, V, b, v[0], v[1], v[2], K=count, B=args.batch, margin=0.5,
                                    hogwild=args.hogwild, want_loss=want_loss, censor=args.censor)

        if pointwise:
            run(0, max(K, W, 1))                                          # sizes every grow-only buffer
        else:
            rt.pairwise_reserve(opt, U, V, b, max(K, W, 1), args.batch)   # allocations stay out of the timed region
        if W:
            # the W warm-up steps go in two calls: the second call a process makes is ~40 us slower on the host than any
            # later one (scratch/host_overhead.py: 200 vs 165 us inside the call, whatever the K of either), and with one
            # warm-up call the timed call would be that second call -- 2 us per step of the driver's 20-step protocol
            w1 = W - W // 2
            run(0, w1)
            if W // 2:
                run(w1, W // 2)
        views[(W, K)] = tuple(x[W:W + K] for x in ((uid, pid, label) if pointwise else (uid, pid, nid)))
        ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(W, K)
        ctx.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # kernel-level timing of the same K steps (HIP events on the library's stream)
        ctx.prof_reset()
        ctx.prof_enable(True)
        losses = run(W, K, want_loss=True)
        ctx.prof_enable(False)
        prof = ctx.prof_get()
        parallelism = "single-gpu"
    else:
        from openrec_amd import sharded
        uid, pid, nid = make_ids(torch, args.users, args.items, K + W, args.batch, 1234 + rank, device, args.zipf)
        cold = 1.0
        if args.hot_items:
            # the exchanged buckets are sized for the share of item references that are NOT replicated (measured on this rank's ids,
            # the largest share of any step, plus a margin; the ranks agree on the largest)
            per_step = ((pid >= args.hot_items).float().mean(1) + (nid >= args.hot_items).float().mean(1)) / 2
            ct = (per_step.max() * 1.05 + 0.002).clamp(max=1.0)
            if dist is not None:
                dist.all_reduce(ct, op=dist.ReduceOp.MAX)
            cold = float(ct.item())
        eng = sharded.ShardedPairwise(args.model, args.opt, args.users, args.items, args.dim, lr=lr,
                                      rank=rank, world=world, device=device, seed=0,
                                      dedup={"auto": None, "on": True, "off": False}[args.shard_dedup],
                                      hot_items=args.hot_items, hot_cold_fraction=cold)
        eng.force_collectives = dist is not None
        if W:
            w1 = W - W // 2
            eng.steps(uid[:w1], pid[:w1], nid[:w1])
            if W // 2:
                eng.steps(uid[w1:W], pid[w1:W], nid[w1:W])
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.steps(uid[W:W + K], pid[W:W + K], nid[W:W + K])          # one K-step call, like the single-GPU path
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        eng.check()
        # ---- diagnosis of the sharded step (after the timed region): the same K steps again with dispatch-attached events on the
        # kernels and HIP events around every exchange, and what a link delivers to the engine's own ncclSend / ncclRecv groups
        sharded_extra = {}

        def all_ranks_ok(ok):
            # the diagnosis below makes COLLECTIVE calls: a rank that failed must take the others out with it, not leave them
            # waiting inside an exchange
            if dist is None:
                return ok
            t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))
        try:
            ready = True
            try:
                eng.check()
            except Exception as e:                          # noqa: BLE001
                ready = False
                sharded_extra["diagnosis_error"] = repr(e)
            if not all_ranks_ok(ready):
                raise RuntimeError(sharded_extra.get("diagnosis_error", "another rank is not ready for the diagnosis pass"))
            ping = eng.comm_ping(32 << 20, 5)              # (first: connections are up before anything is timed)
            eng.be.ctx.prof_reset(); eng.be.ctx.prof_enable(True)
            eng.comm_stats_start()
            tq0 = time.perf_counter()
            eng.steps(uid[W:W + K], pid[W:W + K], nid[W:W + K])
            torch.cuda.synchronize()
            tq = time.perf_counter() - tq0
            st = eng.comm_stats_stop()
            eng.be.ctx.prof_enable(False)
            prof = eng.prof()
            ovf = int(eng._ovf.item()) if eng._ovf is not None else 0
            ker = {k: v for k, v in prof.items() if v.get("launches")}
            ker_ms = sum(v["total_ms"] for v in ker.values())
            bpt_alg = alg_bytes_per_triplet(args.dim, args.opt, args.model)
            sharded_extra["phases_us"] = {
                **{"kernels:" + k: v["total_ms"] / K * 1e3 for k, v in ker.items()},
                "exchanges": (st["exchange_ms"] / K * 1e3) if st else None,
                "step_profiled": tq / K * 1e6}
            if ker_ms > 0:
                # per rank: the rows of its B triplets are read and written once somewhere in the job -- the single-GPU step's algorithmic
                # bytes -- against the time of this rank's kernels (gathers, gradient kernels, applies; the exchanges are not HBM work)
                ach = args.batch * bpt_alg * K / (ker_ms * 1e-3) / 1e9
                sharded_extra["roofline"] = {"bound": "hbm", "kernel": "this rank's gather / gradient / apply kernels of the sharded step",
                                             "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                             "bytes_per_triplet": bpt_alg, "kernel_us_per_step": ker_ms / K * 1e3}
            if st:
                sharded_extra["link_GBps"] = {
                    "ping_out_all_links": ping["GBps_out"] if ping else None, "ping_per_link": ping["GBps_per_link"] if ping else None,
                    "ping_us_per_all_to_all_of_32MiB": ping["us_per_all_to_all"] if ping else None,
                    "in_step_out_all_links": (st["wire_bytes"] / (st["exchange_ms"] * 1e-3) / 1e9) if st["exchange_ms"] > 0 else None,
                    "wire_bytes_per_step": st["wire_bytes"] / K, "self_bytes_per_step": st["self_bytes"] / K,
                    "exchanges_per_step": st["exchanges"] / K}
            sharded_extra["overflow"] = ovf
            if args.hot_items:
                sharded_extra["hot_items"] = {"H": args.hot_items, "cold_fraction_of_item_references": cold,
                                              "allreduce_bytes_per_step": 2.0 * (world - 1) / world * args.hot_items * (args.dim + 4) * 4}
        except Exception as e:                              # the diagnosis must never hide the number
            sharded_extra["diagnosis_error"] = repr(e)
            prof = eng.prof()
        losses = None
        parallelism = f"row-sharded x{world}, all-to-all"

    if rank == 0:
        total = K * args.batch * world
        bpt = alg_bytes_per_triplet(args.dim, args.opt, args.model)
        unit = "samples/s" if args.model in ("gmf", "wrmf") else "triplets/s"
        out = {
            "metric": "BPR training triplets/sec at dim=64, 1Mx1M table" if (args.model, args.dim, args.users, args.items) == ("bpr", 64, 1_000_000, 1_000_000)
                      else f"{args.model.upper()} training {unit[:-2]}/sec at dim={args.dim}",
            "value": total / dt, "unit": unit, "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} dim={args.dim} {args.users}x{args.items} table, "
                                   f"batch={args.batch} triplets/GPU, {args.opt} lr={lr}, objective loss+l2_loss, "
                                   f"{'HOGWILD (non-reference)' if args.hogwild else 'exact TF duplicate semantics'}"
                                   f"{', censor after each step' if args.censor else ''}"
                                   f"{', ids handed over as HOST buffers (PCIe inside the timed call)' if args.host_ids else ''}"
                                   f"{', items ~ Zipf(%g)' % args.zipf if args.zipf else ''}",
                       "parallelism": parallelism},
        }
        fused = prof.get("pointwise" if args.model in ("gmf", "wrmf") else "fused", {})
        if fused.get("launches"):
            dur = fused["total_ms"] / fused["launches"] * 1e-3
            achieved = args.batch * bpt / dur / 1e9
            # HBM bytes per launch from the PMC passes of scripts/collect_profiles.sh -- only if they were measured on THIS
            # build (source hash) and THIS workload; otherwise null, never a stale constant
            traffic, traffic_src = None, None
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if world == 1 and os.path.exists(tfile):
                try:
                    from openrec_amd.build import source_hash
                    tj = json.load(open(tfile))
                    key = {("bpr", 64, "sgd"): "c2", ("ucml", 128, "sgd"): "c3_ucml128_censor", ("bpr", 64, "adagrad"): "bpr_adagrad",
                           ("wrmf", 64, "sgd"): "wrmf"}.get((args.model, args.dim, args.opt))
                    default_shape = (args.users, args.items, args.batch, args.zipf, args.hogwild) == (1_000_000, 1_000_000, 65536, 0.0, False)
                    if key == "c3_ucml128_censor" and not args.censor:
                        key = None
                    if key in tj.get("workloads", {}) and default_shape:
                        if tj.get("source_hash") == source_hash():
                            traffic = tj["workloads"][key]["bytes_per_launch"]
                            traffic_src = f"profiles/{tj['tag']}_pmc_summary.csv"
                        else:
                            traffic_src = f"profiles/{tj['tag']}_pmc_summary.csv was measured on another build: not reported"
                except Exception:
                    traffic = None
            # second yardstick: the streaming-copy rate of THIS device, measured in this process after the
            # timed region by a float4 copy kernel over 1 GiB buffers (orx_copy_bandwidth); and the whole step -- algorithmic bytes
            # over ms_per_step -- against the peak: how much of the step is not the dominant kernel
            copy_gbs = None
            if world == 1 and not args.sharded:
                try:
                    copy_gbs = ctx.copy_bandwidth(1 << 30, 10)
                except Exception:
                    copy_gbs = None
            step_achieved = args.batch * bpt / (dt / K) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "point_fused_kernel" if args.model in ("gmf", "wrmf") else "fused_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                               "copy_peak_measured": copy_gbs, "frac_of_copy_peak": (achieved / copy_gbs) if copy_gbs else None,
                               "step_frac": step_achieved / HBM_PEAK_GBS,
                               "bytes_per_triplet": bpt, "kernel_us": dur * 1e6,
                               "other_kernels_us": {k: v["total_ms"] / v["launches"] * 1e3
                                                    for k, v in prof.items() if v.get("launches") and k not in ("fused", "pointwise")}}
        if world > 1 or args.sharded:
            out.update(sharded_extra)
        if losses is not None:
            out["loss_first_last"] = [float(losses[0][0]), float(losses[0][-1])]
        headline = (args.model, args.dim, args.users, args.items, args.batch, args.opt, args.zipf, args.hogwild, args.host_ids, args.censor) == \
                   ("bpr", 64, 1_000_000, 1_000_000, 65536, "sgd", 0.0, False, False, False)
        if world == 1 and not args.sharded and headline and not args.no_secondary:
            # BASELINE.json configs[2] and configs[4] at the same protocol, after the timed C2 region (the headline tables are
            # released first); a failure here must never hide the headline
            sec = {}
            del U, V, b, uid, pid, nid
            torch.cuda.empty_cache()
            for key, fn in (("c3", lambda: secondary_pairwise(torch, rt, ctx, device, "ucml", 128, "sgd", True, 20, 10)),
                            ("c5_fp16", lambda: secondary_dlrm(torch, rt, ctx, device, True, 20, 10))):
                try:
                    sec[key] = fn()
                except Exception as e:
                    sec[key] = {"error": repr(e)}
            out["secondary"] = sec
        if world == 1 and not args.sharded and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:                      # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
