"""CPU compute backend for the sharded engine, used ONLY by the tests: the same
three building blocks as the C ABI (gather_rows / pair_grads / apply_rows),
restated with the NumPy oracle, so that the exchange plan in
openrec_amd/sharded.py can be exercised with gloo on CPU."""
import numpy as np
import torch

from oracle import numpy_oracle as orc


class Tab:
    def __init__(self, rows, dim):
        self.w = np.zeros((rows, dim), np.float32)


class OracleBackend:
    def __init__(self, opt_kind, lr):
        self.opt_kind, self.lr = opt_kind, lr
        self.opt = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[opt_kind]()

    def begin_step(self):
        if hasattr(self.opt, "begin_step"):
            self.opt.begin_step()

    def make_table(self, rows, dim, seed):
        return Tab(max(rows, 1), dim)

    # checkpoint hooks of ShardedPairwise.save / load: this rank's tables and the oracle optimizer's state for them
    def save_tables(self, path, tables, shard):
        import os, pickle
        os.makedirs(path, exist_ok=True)
        state = {k: {n: (v.get(id(t)) if isinstance(v, dict) else None) for n, t in tables.items()}
                 for k, v in vars(self.opt).items() if isinstance(v, dict)}
        scal = {k: v for k, v in vars(self.opt).items() if not isinstance(v, dict)}
        with open(os.path.join(path, "ref.rank%dof%d.pkl" % shard), "wb") as f:
            pickle.dump(dict(tables={n: t.w for n, t in tables.items()}, state=state, scal=scal), f)

    def load_tables(self, path, tables, shard):
        import os, pickle
        with open(os.path.join(path, "ref.rank%dof%d.pkl" % shard), "rb") as f:
            z = pickle.load(f)
        for n, t in tables.items():
            t.w[:] = z["tables"][n]
        for k, per in z["state"].items():
            d = getattr(self.opt, k)
            for n, t in tables.items():
                if per[n] is not None:
                    d[id(t)] = per[n]
        for k, v in z["scal"].items():
            setattr(self.opt, k, v)

    def gather_rows(self, table, bias, ids, out):
        i = ids.numpy()
        o = out.numpy()
        m = i >= 0
        D = table.w.shape[1]
        o[m, :D] = table.w[i[m]]
        if bias is not None:
            o[m, D] = bias.w[i[m], 0]

    def pair_grads(self, model, dim, u, p, n, valid, b_global, margin, gu, gp, gn, accum):
        m = valid.numpy() >= 0
        un, pn, nn = u.numpy()[m], p.numpy()[m], n.numpy()[m]
        k = un.shape[0]
        if k == 0:
            return
        # the oracle works on tables + ids: use the gathered rows as 3 private tables
        U, V, b = un[:, :dim].copy(), np.concatenate([pn[:, :dim], nn[:, :dim]]), np.concatenate([pn[:, dim], nn[:, dim]])[:, None]
        ar = np.arange(k)
        if model == "bpr":
            loss, l2, _ = orc.bpr_forward(U, V, b, ar, ar, ar + k)
            gr = orc.bpr_grads(U, V, b, ar, ar, ar + k)
            scale = np.float32(k) / np.float32(b_global)         # mean over the GLOBAL batch
            loss = loss * scale
            for key in ("gu", "gp", "gn"):
                l2part = {"gu": U, "gp": V[:k], "gn": V[k:]}[key]
                gr[key] = (gr[key] - l2part) * scale + l2part     # only the loss part scales
            gr["gbp"] = gr["gbp"] * scale; gr["gbn"] = gr["gbn"] * scale
        else:
            loss, l2, _ = orc.ucml_forward(U, V, b, ar, ar, ar + k, margin)
            gr = orc.ucml_grads(U, V, b, ar, ar, ar + k, margin)
        gu.numpy()[m, :dim] = gr["gu"]; gp.numpy()[m, :dim] = gr["gp"]; gn.numpy()[m, :dim] = gr["gn"]
        gp.numpy()[m, dim] = gr["gbp"]; gn.numpy()[m, dim] = gr["gbn"]
        accum += torch.tensor([float(loss), float(l2)], dtype=torch.float64)

    def apply_rows(self, table, bias, ids, grads):
        i = ids.numpy()
        m = i >= 0
        D = table.w.shape[1]
        g = grads.numpy()[m]
        self.opt.apply(table.w, i[m], g[:, :D], key=id(table))
        if bias is not None:
            self.opt.apply(bias.w, i[m], g[:, D:D + 1], key=id(bias))


class FastOracleBackend(OracleBackend):
    """The DEVICE-SIDE exchange plan of the engine's K-step paths (`ShardedPairwise._steps_planned` / `_steps_overlapped`,
    what `bench.py --gpus N` runs) restated on the CPU, contract by contract, from the kernels it stands in for
    (openrec_amd/csrc/kernels_sharded.hip: shard_route_kernel, shard_request_kernel, shard_localize_kernel,
    shard_grads_kernel, and the dedup request plan shard_keys / shard_dedup_slots; padding = -1, slots inside a bucket in arrival
    order, or in ascending row order with dedup), so that those paths run under gloo with
    world > 1 -- the real collectives, asynchronous ones included.  No `rows_dupflags`: the engine then applies every list
    through `apply_rows` (the optimizer's own duplicate rule), which is the path Adagrad / Adam take on the GPU too."""
    fast = True

    def stream_ctx(self):
        import contextlib
        return contextlib.nullcontext()

    def check(self):
        pass

    @staticmethod
    def _bucket(dest, world, cap):
        from openrec_amd.sharded import bucket_slots
        slot, ov = bucket_slots(dest.to(torch.int64), world, cap)
        return slot, bool(ov)

    def shard_route_steps(self, uid, pid, nid, n_users, n_items, world, cap, send, counters, overflow):
        K, B = uid.shape
        send.fill_(-1); counters.zero_()
        for k in range(K):
            u = uid[k].to(torch.int64)
            slot, ov = self._bucket(u % world, world, cap)
            ok = slot >= 0
            send[k, slot[ok], 0] = uid[k][ok]; send[k, slot[ok], 1] = pid[k][ok]; send[k, slot[ok], 2] = nid[k][ok]
            counters[k] = torch.bincount(u % world, minlength=world).to(torch.int32)
            if ov:
                overflow.fill_(1)

    def shard_request_steps(self, trip, world, cap, send_ids, slot, u_loc, counters, overflow):
        K, T = trip.shape[0], trip.shape[1]
        send_ids.fill_(-1); counters.zero_()
        for k in range(K):
            u, p, n = (trip[k, :, c].to(torch.int64) for c in range(3))
            live = u >= 0
            ids = torch.cat([p, n]); alive = torch.cat([live, live])
            dest = torch.where(alive, ids % world, torch.full_like(ids, -1))
            s, ov = self._bucket(dest, world, cap)
            ok = s >= 0
            send_ids[k, s[ok]] = ids[ok].to(torch.int32)
            slot[k] = s.to(torch.int32)
            gp, gn = s[:T], s[T:]
            u_loc[k] = torch.where(live & (gp >= 0) & (gn >= 0), torch.div(u, world, rounding_mode="floor"), torch.full_like(u, -1)).to(torch.int32)
            if ov:
                overflow.fill_(1)

    def shard_request_dedup_steps(self, trip, world, cap, n_items, send_ids, slot, u_loc, dupref, overflow):
        """kernels_sharded.hip: shard_keys_kernel + orx_rows_sort + shard_dedup_slots_kernel.  The distinct items a list asks an
        owner for fill that owner's bucket in ascending local-row order; every reference of an item gets that one slot;
        dupref = 1 on references whose item is asked for more than once; a triplet lives if both its requests found a slot."""
        K, T = trip.shape[0], trip.shape[1]
        send_ids.fill_(-1)
        for k in range(K):
            u, p, n = (trip[k, :, c].numpy().astype(np.int64) for c in range(3))
            live = u >= 0
            ids = np.concatenate([p, n]); alive = np.concatenate([live, live])
            s = np.full(2 * T, -1, np.int64); d = np.zeros(2 * T, np.uint8)
            for o in range(world):
                sel = alive & (ids % world == o)
                uniq, inv, cnt = np.unique(ids[sel] // world, return_inverse=True, return_counts=True)      # ascending local row
                if uniq.size > cap:
                    overflow.fill_(1)
                ok = inv < cap
                refs = np.nonzero(sel)[0]
                s[refs[ok]] = o * cap + inv[ok]
                d[refs] = (cnt[inv] > 1).astype(np.uint8)
                m = min(uniq.size, cap)
                send_ids[k, o * cap:o * cap + m] = torch.from_numpy((uniq[:m] * world + o).astype(np.int32))
            slot[k] = torch.from_numpy(s.astype(np.int32)); dupref[k] = torch.from_numpy(d)
            keep = live & (s[:T] >= 0) & (s[T:] >= 0)
            u_loc[k] = torch.from_numpy(np.where(keep, u // world, -1).astype(np.int32))
        return None                                          # (the GPU plan's opaque outputs: the sums below do not need them)

    def shard_localize(self, ids, world, out):
        out.copy_(torch.where(ids >= 0, torch.div(ids, world, rounding_mode="floor"), torch.full_like(ids, -1)))

    def shard_grads(self, model, user, rows_in, u_loc, slot, b_global, margin, gu, send_g, accum, dupref=None):
        T = u_loc.numel()
        D = user.w.shape[1]
        ul, sp, sn = u_loc.numpy(), slot.numpy()[:T], slot.numpy()[T:]
        live = ul >= 0
        g_out = send_g.numpy()
        if dupref is not None:                               # dedup: shared slots receive the SUM of their references
            g_out[:] = 0.0                                   # (dupref = (flags, plan, list index): only "dedup is on" matters here)
        else:
            for s_ in (sp, sn):                              # surviving requests of dead triplets get zero gradients
                dead = (~live) & (s_ >= 0)
                g_out[s_[dead]] = 0.0
        if not live.any():
            return
        k = int(live.sum())
        rows = rows_in.numpy()
        U = user.w[ul[live]].copy()
        P, Nn = rows[sp[live]], rows[sn[live]]
        V = np.concatenate([P[:, :D], Nn[:, :D]]); b = np.concatenate([P[:, D], Nn[:, D]])[:, None]
        ar = np.arange(k)
        if model == "bpr":
            loss, l2, _ = orc.bpr_forward(U, V, b, ar, ar, ar + k)
            gr = orc.bpr_grads(U, V, b, ar, ar, ar + k)
            scale = np.float32(k) / np.float32(b_global)
            loss = loss * scale
            for key, l2part in (("gu", U), ("gp", V[:k]), ("gn", V[k:])):
                gr[key] = (gr[key] - l2part) * scale + l2part
            gr["gbp"] = gr["gbp"] * scale; gr["gbn"] = gr["gbn"] * scale
        else:
            loss, l2, _ = orc.ucml_forward(U, V, b, ar, ar, ar + k, margin)
            gr = orc.ucml_grads(U, V, b, ar, ar, ar + k, margin)
        gu.numpy()[live] = gr["gu"]
        if dupref is not None:
            full = np.zeros((k, g_out.shape[1]), np.float32)
            for sl_, g_, gb_ in ((sp[live], gr["gp"], gr["gbp"]), (sn[live], gr["gn"], gr["gbn"])):
                full[:, :D] = g_; full[:, D] = np.asarray(gb_).reshape(-1)
                np.add.at(g_out, sl_, full)
        else:
            g_out[sp[live], :D] = gr["gp"]; g_out[sp[live], D] = gr["gbp"]
            g_out[sn[live], :D] = gr["gn"]; g_out[sn[live], D] = gr["gbn"]
        accum += torch.tensor([float(loss), float(l2)], dtype=torch.float64)


class FlaggedOracleBackend(FastOracleBackend):
    """... plus the SGD specialisations the GPU engine takes (kernels_sharded.hip: rows_dupflags, shard_grads_kernel<APPLY>,
    apply_rows_sgd_flagged_kernel): duplicate flags per apply list, user rows referenced once updated by the gradient kernel
    itself, flagged applies for the rest.  SGD only (the engine asks for flags only then)."""

    def rows_dupflags(self, table, ids2d, out):
        ids = ids2d.numpy()
        o = out.numpy().reshape(ids.shape)
        for k in range(ids.shape[0]):
            live = ids[k] >= 0
            uniq, cnt = np.unique(ids[k][live], return_counts=True)
            dup = np.isin(ids[k], uniq[cnt > 1]) & live
            o[k] = dup.astype(np.uint8)

    def apply_rows_flagged(self, table, bias, ids, grads, dflag):
        self.apply_rows(table, bias, ids, grads)          # scatter_add is the same sum with or without the flags

    def shard_grads_sgd(self, model, user, rows_in, u_loc, slot, dup_u, b_global, margin, gu, u_apply, send_g, accum, dupref=None):
        assert self.opt_kind == "sgd"
        self.shard_grads(model, user, rows_in, u_loc, slot, b_global, margin, gu, send_g, accum, dupref=dupref)
        ul, dup = u_loc.numpy(), dup_u.numpy() != 0
        once = (ul >= 0) & ~dup
        user.w[ul[once]] -= np.float32(self.lr) * gu.numpy()[once]
        ua = u_apply.numpy()
        ua[:] = np.where((ul >= 0) & dup, ul, -1)
