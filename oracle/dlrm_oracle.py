"""CPU oracle for the reference's DLRM train step (NumPy; TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED (no reference tests / golden vectors; TensorFlow absent) -- pinned
against torch autograd fixtures (tests/golden/make_golden_dlrm.py).

Restates (paths relative to /root/reference):
  openrec/tf2/recommenders/dlrm.py:8-100          model wiring, loss, clipping
  openrec/tf2/modules/multi_layer_perceptron.py:5-18   Dense stacks (hidden relu)
  openrec/tf2/modules/second_order_feature_interaction.py:12-34
  tf2_examples/dlrm_criteo.py:42-48               tape.gradient(loss) + apply_gradients

`reference_compat=True` reproduces second_order_feature_interaction.py bit for
bit, including its bug: line 21 keeps the LOWER triangle of
Z Z^T, lines 23-27/32 select the strictly UPPER triangle, so every selected
element is 0 (only the diagonal survives with self_interaction) and the
embedding tables receive zero gradients.  `reference_compat=False` is the
evidently intended strictly-lower-triangle pairwise dot product.

`operand_dtype=np.float16` restates the library's ORX_DLRM_FP16_MLP mode (north_star: "dense top-MLP on fp16 MFMA"): every
MLP product -- X W, dZ W^T, X^T dZ -- rounds BOTH operands to fp16 once and accumulates in `dtype`; biases, activations,
bias gradients (column sums of the unrounded dZ), the feature interaction, the loss and the optimizer stay in `dtype`.
The gradients are rounded under a static LOSS SCALE, as the library does (csrc/dlrm.hip: loss_scale): S = the power of two
>= the batch the loss mean runs over (at most 2^15); fp16(S dZ) / S -- dLoss/dPred is O(1 / B), which at B = 8192 lies in
fp16's subnormal range.

`tie_margin` reports how close a sample comes to a discontinuity of the network (a relu pre-activation near zero, the
prediction near a clipping threshold): two correct fp32 implementations that add the same terms in a different order can
land on different sides of it, and the sample's share of every gradient below then differs at first order.  Parity tests
draw their batches so that no sample lies within summation noise of such a point (tests/dlrm_util.py), and hold everything
else to 1e-5.
"""
from __future__ import annotations

import numpy as np

from . import numpy_oracle as orc

KERAS_EPS = 1e-7


def glorot(rng, fan_in, fan_out, dtype):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, (fan_in, fan_out)).astype(dtype)


def interaction_pairs(F, self_interaction, reference_compat):
    """Returns (I, J) index arrays of the selected (row, col) elements in
    tf.boolean_mask (row-major) order, and whether off-diagonal values are live."""
    if reference_compat:
        mask = np.triu(np.ones((F, F), bool), k=0 if self_interaction else 1)
    else:
        mask = np.tril(np.ones((F, F), bool), k=0 if self_interaction else -1)
    I, J = np.nonzero(mask)
    return I, J


class DLRMOracle:
    def __init__(self, m_spa, ln_emb, ln_bot, ln_top, dense_dim, arch_interaction_itself=False,
                 sigmoid_bot=False, sigmoid_top=True, loss_func="mse", loss_threshold=0.0,
                 reference_compat=True, dtype=np.float32, seed=0, operand_dtype=None):
        rng = np.random.default_rng(seed)
        self.dt = np.dtype(dtype)
        self.op_dt = None if operand_dtype is None else np.dtype(operand_dtype)
        self.m_spa, self.ln_emb = m_spa, list(ln_emb)
        self.emb = [rng.uniform(-0.05, 0.05, (n, m_spa)).astype(dtype) for n in ln_emb]   # dlrm.py:32-33
        self.bot, d = [], dense_dim
        for u in ln_bot:
            self.bot.append([glorot(rng, d, u, dtype), np.zeros(u, dtype)]); d = u
        assert d == m_spa, "the bottom MLP must end at m_spa (its output is stacked with the embeddings)"
        self.F = len(ln_emb) + 1
        self.I, self.J = interaction_pairs(self.F, arch_interaction_itself, reference_compat)
        self.compat = reference_compat
        self.top, d = [], m_spa + len(self.I)
        for u in ln_top:
            self.top.append([glorot(rng, d, u, dtype), np.zeros(u, dtype)]); d = u
        self.bot_act = ["relu"] * (len(ln_bot) - 1) + ["sigmoid" if sigmoid_bot else "relu"]   # dlrm.py:34-35
        self.top_act = ["relu"] * (len(ln_top) - 1) + ["sigmoid" if sigmoid_top else "relu"]   # dlrm.py:36-37
        self.loss_func, self.thr = loss_func, loss_threshold
        self._gscale = self.dt.type(1.0)

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _act(x, a):
        if a == "relu":
            return np.maximum(x, 0)
        if a == "sigmoid":
            return orc._sigmoid(x)
        return x

    def _op(self, a):
        """operand of an MLP product: rounded to `operand_dtype` once (fp16 mode), kept in `dtype` otherwise"""
        return a if self.op_dt is None else a.astype(self.op_dt).astype(self.dt)

    def _mlp(self, x, layers, acts, probe=None, x_noise=None):
        """probe (dict, optional) collects what tie_margin / fp16_flip_risk report; x_noise: absolute uncertainty of the
        input activations between two correct implementations (None: the inputs are given data, identical on both sides)"""
        outs = [x]
        for (W, b), a in zip(layers, acts):
            xo, Wo = self._op(x), self._op(W)
            z = xo @ Wo + b
            S = None
            if probe is not None:
                S = np.abs(xo) @ np.abs(Wo) + np.abs(b) + np.finfo(self.dt).tiny      # the magnitudes z was added up from
                move = None
                if self.op_dt is not None and x_noise is not None:
                    move = self._flip_move(x, xo, Wo, x_noise)
                if a == "relu":
                    probe["margin"].append((np.abs(z) / S).min(axis=1))
                    if "ties" in probe:          # (net, layer, sample, unit) of every relu unit within delta of zero: tie_units
                        for bi, ji in np.argwhere(np.abs(z) / S < probe["delta"]):
                            probe["ties"].append((probe["net"], len(outs) - 1, int(bi), int(ji)))
                    if move is not None:
                        probe["risky"] |= (np.abs(z) < move + probe["delta"] * S).any(axis=1)
            x = self._act(z, a)
            if probe is not None:       # what the next layer's inputs may differ by: the summation noise of this layer.  (The doubtful
                # roundings themselves are NOT passed on as certain movement: taken as certain and coherent they mark every sample
                # within three layers, while in fact a handful per step come true -- tests/test_gpu_dlrm.py budgets for those.)
                x_noise = probe["kappa"] * S * (1.0 if a != "sigmoid" else 0.25)
            outs.append(x)
        if probe is not None:
            probe["out_noise"] = x_noise
        return outs

    def _flip_move(self, x, xo, Wo, x_noise):
        """fp16 mode, [B, units]: how far the input activations that another correct implementation may round to the
        neighbouring fp16 value (they sit within the implementations' disagreement `x_noise` of a rounding boundary) can move
        each pre-activation of this layer.  Each such operand moves z_j by one fp16 ulp of x_k times |W_kj|; all of a sample's
        doubtful operands are taken to move together (a 1024-term product has several).  A relu unit closer to zero than
        that may come out on the other side there; the next layer inherits the movement as input uncertainty."""
        ulp = np.spacing(np.abs(xo).astype(self.op_dt)).astype(self.dt)
        dist = 0.5 * ulp - np.abs(x - xo)                       # distance of the unrounded activation from the rounding boundary
        doubt = np.where((dist < x_noise) & (np.abs(x) > 0), ulp, 0.0)
        return doubt @ np.abs(Wo)

    def forward(self, dense, sparse, emb_rows=None, probe=None):
        """emb_rows [B, n_emb, d]: embedding vectors handed in instead of looked up (the hybrid-parallel
        step of openrec_amd/sharded_dlrm.py exchanges them between ranks first)."""
        dense = dense.astype(self.dt)
        if probe is not None:
            probe["net"] = "bot"
        bot = self._mlp(dense, self.bot, self.bot_act, probe, None)                     # dlrm.py:87
        if emb_rows is not None:
            vecs = [emb_rows[:, f, :].astype(self.dt) for f in range(emb_rows.shape[1])] + [bot[-1]]
        else:
            vecs = [self.emb[f][sparse[:, f]] for f in range(len(self.emb))] + [bot[-1]]    # dlrm.py:83-85, :91
        Z = np.stack(vecs, 1)                                                           # [B, F, d]
        dots = np.einsum("bfd,bgd->bfg", Z, Z)
        if self.compat:
            dots = np.tril(dots)                                                        # interaction.py:21
        inter = dots[:, self.I, self.J]
        R = np.concatenate([bot[-1], inter], 1)                                         # dlrm.py:90-92
        R_noise = None
        if probe is not None:       # the interaction's outputs are fp32 dot products of the (fp32) feature rows
            aZ = np.abs(Z)
            Sd = np.einsum("bfd,bgd->bfg", aZ, aZ)[:, self.I, self.J]
            R_noise = np.concatenate([probe["out_noise"] if probe["out_noise"] is not None else np.zeros_like(bot[-1]), probe["kappa"] * Sd], 1)
        if probe is not None:
            probe["net"] = "top"
        top = self._mlp(R, self.top, self.top_act, probe, R_noise)
        p = top[-1]
        clip_mask = np.ones_like(p)
        if 0.0 < self.thr < 1.0:                                                        # dlrm.py:97-98
            lo, hi = self.dt.type(self.thr), self.dt.type(1.0 - self.thr)
            if probe is not None:
                probe["margin"].append(np.minimum(np.abs(p - lo), np.abs(p - hi)).reshape(-1))
            clip_mask = ((p >= lo) & (p <= hi)).astype(self.dt)
            p = np.clip(p, lo, hi)
        return dict(bot=bot, Z=Z, R=R, top=top, pred=p.reshape(-1), clip_mask=clip_mask.reshape(-1))

    def inference(self, dense, sparse):
        return self.forward(dense, sparse)["pred"]

    def tie_margin(self, dense, sparse, emb_rows=None, kappa=2e-7, delta=1e-6):
        """[B]: the smallest relative distance of any relu pre-activation of the sample from zero, |z| / (|x| |W| + |b|)
        (and of the prediction from a clipping threshold).  fp16 mode: -1 for samples where an fp16 rounding of a hidden
        activation that two correct implementations may resolve differently could flip a relu unit (_flip_move; `kappa`
        = the relative disagreement of fp32 dot products, in units of the summed magnitudes)."""
        probe = dict(margin=[], risky=np.zeros(dense.shape[0], bool), kappa=kappa, delta=delta, out_noise=None)
        self.forward(dense, sparse, emb_rows, probe)
        m = np.min(np.stack(probe["margin"], 0), axis=0) if probe["margin"] else np.full(dense.shape[0], np.inf)
        return np.where(probe["risky"], -1.0, m)

    def tie_units(self, dense, sparse, emb_rows=None, delta=1e-6):
        """[(net, layer, sample, unit)]: the relu units whose pre-activation lies within `delta` of zero, relative to the sum of
        the magnitudes it was added up from -- where two correct fp32 implementations (another order of the same additions) may
        put the unit on different sides.  The forward value barely notices (|z| ~ 0); the BACKWARD mask `y > 0` is 0 on one side
        and 1 on the other: `loss_and_grads(flip=...)` gives the gradients with chosen units on the other side."""
        probe = dict(margin=[], risky=np.zeros(dense.shape[0], bool), kappa=2e-7, delta=delta, out_noise=None, ties=[])
        self.forward(dense, sparse, emb_rows, probe)
        return probe["ties"]

    # ----------------------------------------------------------- loss + backward
    def loss_and_grads(self, dense, sparse, label, emb_rows=None, global_batch=None, flip=None):
        """global_batch: the loss mean runs over that many samples (this call sees a slice of them);
        the returned loss is then this slice's share of the global mean.
        flip: {(net, layer): bool [B, units]} -- relu units whose backward mask is inverted (tie_units)."""
        c = self.forward(dense, sparse, emb_rows)
        p, y = c["pred"], label.astype(self.dt)
        B = p.shape[0] if global_batch is None else int(global_batch)
        if self.loss_func == "mse":                                                     # Keras MeanSquaredError
            loss = ((y - p) ** 2).sum(dtype=self.dt) / self.dt.type(B)
            dp = 2 * (p - y) / self.dt.type(B)
        else:                                                                           # Keras BinaryCrossentropy
            eps = self.dt.type(KERAS_EPS)
            pc = np.clip(p, eps, 1 - eps)
            loss = -(y * np.log(pc + eps) + (1 - y) * np.log(1 - pc + eps)).sum(dtype=self.dt) / self.dt.type(B)
            inside = ((p >= eps) & (p <= 1 - eps)).astype(self.dt)
            dp = -(y / (pc + eps) - (1 - y) / (1 - pc + eps)) * inside / self.dt.type(B)
        dp = (dp * c["clip_mask"]).reshape(-1, 1)
        S = 1.0
        if self.op_dt is not None:
            while S < B and S < 32768.0:
                S *= 2.0
        self._gscale = self.dt.type(S)
        fl = flip or {}
        g_top, dR = self._mlp_backward(c["top"], self.top, self.top_act, dp, {l: m for (n, l), m in fl.items() if n == "top"})
        d = self.m_spa
        dZ = np.zeros_like(c["Z"])
        dZ[:, self.F - 1, :] += dR[:, :d]
        G = dR[:, d:]                                                                   # [B, P]
        Z = c["Z"]
        for k, (i, j) in enumerate(zip(self.I, self.J)):
            live = (i >= j) if self.compat else True        # compat: only the kept lower triangle carries values
            if not live:
                continue
            g = G[:, k:k + 1]
            if i == j:
                dZ[:, i, :] += 2 * g * Z[:, i, :]
            else:
                dZ[:, i, :] += g * Z[:, j, :]
                dZ[:, j, :] += g * Z[:, i, :]
        g_bot, _ = self._mlp_backward(c["bot"], self.bot, self.bot_act, dZ[:, self.F - 1, :], {l: m for (n, l), m in fl.items() if n == "bot"})
        return loss, dict(emb=dZ[:, :self.F - 1, :], bot=g_bot, top=g_top)

    def _mlp_backward(self, outs, layers, acts, dy, flip=None):
        grads = [None] * len(layers)
        for l in range(len(layers) - 1, -1, -1):
            y = outs[l + 1]
            if acts[l] == "relu":
                dz = dy * ((y > 0) ^ flip[l] if flip and l in flip else (y > 0))
            elif acts[l] == "sigmoid":
                dz = dy * y * (1 - y)
            else:
                dz = dy
            dzo = self._op(dz * self._gscale) / self._gscale
            grads[l] = (self._op(outs[l]).T @ dzo, dz.sum(0))
            dy = dzo @ self._op(layers[l][0]).T
        return grads, dy

    # ------------------------------------------------------------------- step
    def step(self, dense, sparse, label, opt):
        loss, g = self.loss_and_grads(dense, sparse, label)
        if hasattr(opt, "begin_step"):
            opt.begin_step()
        for f in range(len(self.emb)):
            opt.apply(self.emb[f], sparse[:, f], g["emb"][:, f, :], key=("emb", f))
        for name, layers, gl in (("bot", self.bot, g["bot"]), ("top", self.top, g["top"])):
            for l, ((W, b), (gW, gb)) in enumerate(zip(layers, gl)):
                opt.apply_dense(W, gW.astype(self.dt), key=(name, l, "W"))
                opt.apply_dense(b, gb.astype(self.dt), key=(name, l, "b"))
        return loss
