"""Training step of the conditional WGAN-GP of cWGAN_my.ipynb on the HIP kernels of libcgrna.

    Generator      (cell 1)  [z | embeddings | numeric] -> (Linear-BatchNorm-ReLU) x3 -> Linear -> library-size norm
    Discriminator  (cell 2)  [x | embeddings | numeric] -> (Linear-LeakyReLU(0.2)-Dropout(0.3)) x3 -> Linear
    train_gan      (cell 3)  nb_critic critic updates (Wasserstein + lambda_gp * gradient penalty, double backward),
                             then one generator update; Adam with amsgrad for both

Same building blocks as the cycle engine: flat parameter arenas with one fused Adam launch, hand-derived backward,
the critic's three calls of a critic update (real | generated | interpolated) stacked along the rows, the penalty's
double-backward chain writing into a second gradient buffer so that it runs beside the Wasserstein chain, losses
reduced on the device, HIP-graph replay of both steps.
"""
from __future__ import annotations

from typing import Dict

import torch

from .engine import EPI_LEAKY, EPI_MUL, Arena, EngineBase, Net, split_state

S_REAL, S_FAKE, S_GP, S_GADV = 0, 1, 2, 3
N_SLOTS, MAX_PARTS = 4, 1024
D_WIDTHS = (1024, 512, 256)
G_WIDTHS = (256, 512, 1024)


def _pad4(n):
    return (n + 3) // 4 * 4


class GenW:
    """One call of the generator on a batch (workspace, forward, backward)."""

    def __init__(self, eng, net: Net, z_dim, emb_dims, n_num, x_dim, max_b):
        ops = eng.ops
        self.ops, self.net = ops, net
        self.z_dim, self.emb_dims, self.n_num, self.x_dim = z_dim, list(emb_dims), n_num, x_dim
        self.in_dim = z_dim + sum(emb_dims) + n_num
        Z, B = ops.zeros, max_b
        self.inp = Z(B, _pad4(self.in_dim))
        self.h0 = [Z(B, w) for w in G_WIDTHS]
        self.a = [Z(B, w) for w in G_WIDTHS]
        self.stats = [(Z(w), Z(w)) for w in G_WIDTHS]
        self.raw, self.out, self.rowsum = Z(B, x_dim), Z(B, x_dim), Z(B)
        self.d_raw = Z(B, x_dim)
        self.d_act = [Z(B, w) for w in G_WIDTHS]
        self.d_h0 = [Z(B, w) for w in G_WIDTHS]
        self.d_inp = Z(B, _pad4(self.in_dim))
        self.cat = None
        self.b = 0

    def forward(self, z, cat, num, library_size, train=True):
        ops, P, Bf = self.ops, self.net.P, self.net.B
        b = z.shape[0]
        self.b, self.cat = b, cat
        inp = self.inp[:b, :self.in_dim]
        ops.axpby(inp[:, :self.z_dim], 1.0, z)
        off = self.z_dim
        for i, e in enumerate(self.emb_dims):
            ops.embed_gather(inp[:, off:off + e], P[f"embeddings.{i}.weight"], cat[:, i:i + 1])
            off += e
        if self.n_num:
            ops.axpby(inp[:, off:off + self.n_num], 1.0, num)
        a = inp
        for li in range(3):
            lin, bn = f"network.{3 * li}", f"network.{3 * li + 1}"
            ops.gemm(self.h0[li][:b], a, P[lin + ".weight"], False, True, bias=P[lin + ".bias"])
            if train:
                ops.bn_fwd(self.a[li][:b], self.h0[li][:b], P[bn + ".weight"], P[bn + ".bias"], self.stats[li][0],
                           self.stats[li][1], Bf[bn + ".running_mean"], Bf[bn + ".running_var"], 0.0)
                self.net.nbt[bn + ".num_batches_tracked"] += 1
            else:
                ops.bn_eval(self.a[li][:b], self.h0[li][:b], P[bn + ".weight"], P[bn + ".bias"],
                            Bf[bn + ".running_mean"], Bf[bn + ".running_var"], 0.0)
            a = self.a[li][:b]
        ops.gemm(self.raw[:b], a, P["network.9.weight"], False, True, bias=P["network.9.bias"])
        ops.libsize_fwd(self.out[:b], self.raw[:b], self.rowsum[:b], library_size)
        return self.out[:b]

    def backward(self, d_out, library_size):
        """Accumulates the parameter gradients of <d_out, output> into the generator's gradient arena."""
        ops, P, G, b = self.ops, self.net.P, self.net.G, self.b
        d = self.d_raw[:b]
        ops.libsize_bwd(d, d_out, self.raw[:b], self.rowsum[:b], library_size)
        ops.gemm(G["network.9.weight"], d, self.a[2][:b], True, False, accumulate=True)
        ops.colsum(G["network.9.bias"], d, 1.0, True)
        ops.gemm(self.d_act[2][:b], d, P["network.9.weight"], False, False)
        inp = self.inp[:b, :self.in_dim]
        for li in (2, 1, 0):
            lin, bn = f"network.{3 * li}", f"network.{3 * li + 1}"
            ops.bn_bwd(self.d_h0[li][:b], self.d_act[li][:b], self.h0[li][:b], self.a[li][:b], None, self.stats[li][0],
                       self.stats[li][1], P[bn + ".weight"], G[bn + ".weight"], G[bn + ".bias"], True, 0.0)
            a_prev = self.a[li - 1][:b] if li > 0 else inp
            ops.gemm(G[lin + ".weight"], self.d_h0[li][:b], a_prev, True, False, accumulate=True)
            # the Linear bias feeds a BatchNorm: its gradient is identically zero and is left at zero
            d_prev = self.d_act[li - 1][:b] if li > 0 else self.d_inp[:b, :self.in_dim]
            if li > 0 or self.emb_dims:
                ops.gemm(d_prev, self.d_h0[li][:b], P[lin + ".weight"], False, False)
        off = self.z_dim
        for i, e in enumerate(self.emb_dims):
            ops.embed_scatter(G[f"embeddings.{i}.weight"], self.d_inp[:b, off:off + e], self.cat[:, i:i + 1], True)
            off += e


class CriticW:
    """`nseg` calls of the critic stacked along the rows (no BatchNorm in this critic, so a stack is just a taller
    batch).  Row segment s holds call s; segments are packed at stride b."""

    def __init__(self, eng, net: Net, x_dim, emb_dims, n_num, max_b, nseg):
        ops = eng.ops
        self.ops, self.net = ops, net
        self.x_dim, self.emb_dims, self.n_num, self.S = x_dim, list(emb_dims), n_num, nseg
        self.E = sum(emb_dims)
        self.in_dim = x_dim + self.E + n_num
        Z, R = ops.zeros, nseg * max_b
        self.inp = Z(R, _pad4(self.in_dim))
        self.act = [Z(R, w) for w in D_WIDTHS]
        self.pm = [Z(R, w) for w in D_WIDTHS]
        self.val, self.dval = Z(R, 1), Z(R, 1)
        self.g = [Z(R, w) for w in D_WIDTHS]          # first-order chains (both branches use disjoint row ranges)
        self.h = [Z(max_b, w) for w in D_WIDTHS]      # second-order chain of the penalty
        self.demb = Z(R, max(self.E, 1))
        self.ux = Z(max_b, x_dim)
        self.rowpen = Z(max_b)
        self.ones = ops.to_device(torch.ones(max_b, 1))
        self.cat = ops.to_device(torch.zeros(R, max(len(emb_dims), 1), dtype=torch.int64))
        self.b = 0

    def x_seg(self, s, b):
        return self.inp[s * b:(s + 1) * b, :self.x_dim]

    def load(self, s, b, x, cat, num):
        P = self.net.P
        rows = slice(s * b, (s + 1) * b)
        if x is not None:
            self.inp[rows, :self.x_dim].copy_(x)
        off = self.x_dim
        if self.emb_dims:
            self.cat[rows].copy_(cat)
        for i, e in enumerate(self.emb_dims):
            self.ops.embed_gather(self.inp[rows, off:off + e], P[f"embeddings.{i}.weight"], cat[:, i:i + 1])
            off += e
        if self.n_num:
            self.inp[rows, off:off + self.n_num].copy_(num)

    def forward(self, b, nseg, masks):
        """masks: three keep masks (nseg*b, width) or None (eval).  Returns the validity (nseg*b, 1)."""
        ops, P = self.ops, self.net.P
        self.b = b
        R = nseg * b
        a = self.inp[:R, :self.in_dim]
        for li, w in enumerate(D_WIDTHS):
            lin = f"main_network.{3 * li}"
            ops.gemm(self.act[li][:R], a, P[lin + ".weight"], False, True, bias=P[lin + ".bias"], epi=EPI_LEAKY,
                     E=None if masks is None else masks[li], aux=self.pm[li][:R], slope=0.2)
            a = self.act[li][:R]
        ops.gemm(self.val[:R], a, P["main_network.9.weight"], False, True, bias=P["main_network.9.bias"])
        return self.val[:R]

    def _chain(self, rows, dval, G, param_grads, want_dx, dx_out=None):
        """Backward of sum(dval * validity) over `rows`: optional parameter gradients into G (accumulated), optional
        input gradient into dx_out (overwritten)."""
        ops, P = self.ops, self.net.P
        if param_grads:
            ops.gemm(G["main_network.9.weight"], dval, self.act[2][rows], True, False, accumulate=True)
            ops.colsum(G["main_network.9.bias"], dval, 1.0, True)
        d = self.g[2][rows]
        ops.gemm(d, dval, P["main_network.9.weight"], False, False, epi=EPI_MUL, E=self.pm[2][rows])
        for li in (2, 1, 0):
            lin = f"main_network.{3 * li}"
            a_prev = self.act[li - 1][rows] if li > 0 else self.inp[rows, :self.in_dim]
            if param_grads:
                ops.gemm(G[lin + ".weight"], d, a_prev, True, False, accumulate=True)
                ops.colsum(G[lin + ".bias"], d, 1.0, True)
            if li > 0:
                nd = self.g[li - 1][rows]
                ops.gemm(nd, d, P[lin + ".weight"], False, False, epi=EPI_MUL, E=self.pm[li - 1][rows])
                d = nd
        W1 = P["main_network.0.weight"]
        if param_grads and self.emb_dims:
            demb = self.demb[rows, :self.E]
            ops.gemm(demb, d, W1[:, self.x_dim:self.x_dim + self.E], False, False)
            off = 0
            for i, e in enumerate(self.emb_dims):
                ops.embed_scatter(G[f"embeddings.{i}.weight"], demb[:, off:off + e], self.cat[rows, i:i + 1], True)
                off += e
        if want_dx:
            ops.gemm(dx_out, d, W1[:, :self.x_dim], False, False)

    def backward_params(self, b, scales):
        """Critic update: gradients of sum_s scales[s] * sum(validity of segment s) over the first len(scales)
        segments."""
        R = len(scales) * b
        for k, sc in enumerate(scales):          # on the device at every call, so that a captured step carries the writes
            self.dval[k * b:(k + 1) * b].fill_(sc)
        self._chain(slice(0, R), self.dval[:R], self.net.G, True, False)

    def backward_input(self, b, seg, scale, dx_out):
        """Generator update: dx_out = d(scale * sum(validity of segment seg))/d(input batch)."""
        rows = slice(seg * b, (seg + 1) * b)
        self.dval[rows].fill_(scale)
        self._chain(rows, self.dval[rows], None, False, True, dx_out)

    def gp_backward(self, b, seg, lambda_gp, red, slot):
        """Segment `seg` holds the call on the interpolates.  First-order chain to the input gradient, penalty (row
        terms reduced into `slot`), second-order chain: d(lambda*penalty)/d(params) into the second gradient buffer,
        every tensor overwritten on first touch (the biases and the embedding tables get no penalty gradient: the
        input gradient does not depend on them except through the fixed activation pattern)."""
        ops, P, G = self.ops, self.net.P, self.net.G2
        rows = slice(seg * b, (seg + 1) * b)
        pm1, pm2, pm3 = self.pm[0][rows], self.pm[1][rows], self.pm[2][rows]
        W1, W2, W3, w4 = (P[f"main_network.{i}.weight"] for i in (0, 3, 6, 9))
        u3, u2, u1 = self.g[2][rows], self.g[1][rows], self.g[0][rows]
        ops.gemm(u3, self.ones[:b], w4, False, False, epi=EPI_MUL, E=pm3)
        ops.gemm(u2, u3, W3, False, False, epi=EPI_MUL, E=pm2)
        ops.gemm(u1, u2, W2, False, False, epi=EPI_MUL, E=pm1)
        ux = self.ux[:b]
        ops.gemm(ux, u1, W1[:, :self.x_dim], False, False)
        ops.gp_penalty(ux, self.rowpen[:b], ux, lambda_gp)          # ux now holds V = d(lambda*penalty)/d(ux)
        ops.reduce_sum(red, slot, self.rowpen[:b])
        gW1 = G["main_network.0.weight"]
        ops.gemm(gW1[:, :self.x_dim], u1, ux, True, False)
        t1 = self.h[0][:b]
        ops.gemm(t1, ux, W1[:, :self.x_dim], False, True, epi=EPI_MUL, E=pm1)
        ops.gemm(G["main_network.3.weight"], u2, t1, True, False)
        t2 = self.h[1][:b]
        ops.gemm(t2, t1, W2, False, True, epi=EPI_MUL, E=pm2)
        ops.gemm(G["main_network.6.weight"], u3, t2, True, False)
        t3 = self.h[2][:b]
        ops.gemm(t3, t2, W3, False, True, epi=EPI_MUL, E=pm3)
        ops.colsum(G["main_network.9.weight"], t3, 1.0, False)


class CwganEngine(EngineBase):
    """State of a cWGAN training run on one device + its two steps."""

    def __init__(self, g_state: Dict[str, torch.Tensor], d_state: Dict[str, torch.Tensor], cfg: dict, ops, max_batch):
        self.ops, self.cfg, self.max_b = ops, dict(cfg), max_batch
        gp, gb = split_state(g_state)
        dp, db = split_state(d_state)
        self.arena_g = Arena(ops, [(("g", k), tuple(v.shape)) for k, v in gp.items()])
        self.arena_d = Arena(ops, [(("d", k), tuple(v.shape)) for k, v in dp.items()], second_grad=True)
        self.nets = {"g": Net("g", self.arena_g, list(gp.keys()), gb, ops), "d": Net("d", self.arena_d, list(dp.keys()), db, ops)}
        for name, params in (("g", gp), ("d", dp)):
            for k, v in params.items():
                self.nets[name].P[k].copy_(ops.to_device(v.detach().float()))
        emb_g = [v.shape[1] for k, v in gp.items() if k.startswith("embeddings.")]
        emb_d = [v.shape[1] for k, v in dp.items() if k.startswith("embeddings.")]
        self.z_dim = int(cfg["latent_dim"])
        self.x_dim = gp["network.9.weight"].shape[0]
        self.n_num = gp["network.0.weight"].shape[1] - self.z_dim - sum(emb_g)
        assert self.n_num >= 0 and dp["main_network.0.weight"].shape[1] == self.x_dim + sum(emb_d) + self.n_num
        assert not any(k.startswith("negative_detector") for k in dp), "use_neg_detector=True is not supported"
        self.lib = float(cfg.get("library_size", 10000))
        self.lr_g = self.lr_d = float(cfg["lr"])
        self.betas = (float(cfg.get("beta1", 0.5)), float(cfg.get("beta2", 0.9)))
        self.eps = float(cfg.get("eps", 1e-8))
        B = max_batch
        self.gen = GenW(self, self.nets["g"], self.z_dim, emb_g, self.n_num, self.x_dim, B)
        self.crit = CriticW(self, self.nets["d"], self.x_dim, emb_d, self.n_num, B, 3)
        Z = ops.zeros
        self.red = (Z(N_SLOTS, MAX_PARTS), Z(N_SLOTS, dtype=torch.int32))
        self.tot = Z(N_SLOTS, dtype=torch.float64)
        self.slots = ops.to_device(torch.arange(N_SLOTS, dtype=torch.int32))
        self.scales = Z(N_SLOTS, dtype=torch.float64)
        self._scales_b = None
        self.d_fake = Z(B, self.x_dim)
        from .ddp import GradSync
        self.sync = GradSync(ops.device, stream=(lambda: ops.stream("comm", scratch=False)) if hasattr(ops, "stream") else None)
        assert not self.sync.enabled, "the cWGAN engine is single-rank"

    def _adam(self, arena):
        kw = {} if arena.p16 is None else dict(p16=arena.p16)
        self.ops.adam(arena.p, arena.g, arena.m, arena.v, arena.vmax, arena.lr_step, self.betas[0], self.betas[1], self.eps,
                      **kw)

    def _set_scales(self, b):
        if self._scales_b != b:
            lam = float(self.cfg.get("lambda_gp", 10))
            self.scales.copy_(torch.tensor([-1.0 / b, 1.0 / b, lam / b, -1.0 / b], dtype=torch.float64))
            self._scales_b = b

    @staticmethod
    def _masks(noise, b, nseg, alpha_before=None):
        """Keep masks of nseg stacked critic calls (+ alpha): one draw per layer from a device source, call by call
        in the reference's order (three masks per call; alpha before the call on the interpolates) otherwise."""
        alpha = None
        if getattr(noise, "stacked_draws", False):
            if alpha_before is not None:
                alpha = noise.rand((b, 1))
            return [noise.keep((nseg * b, w), 0.3) for w in D_WIDTHS], alpha
        per = []
        for s in range(nseg):
            if s == alpha_before:
                alpha = noise.rand((b, 1))
            per.append([noise.keep((b, w), 0.3) for w in D_WIDTHS])
        return [torch.cat([per[s][li] for s in range(nseg)]) for li in range(3)], alpha

    # ---- critic update (cell 3, "Train Discriminator") ----------------------------------------
    def d_step(self, real, cat, num, noise):
        b = real.shape[0]
        assert b <= self.max_b
        self._set_scales(b)
        self._begin_opt(self.arena_d, self.lr_d)
        ins = (real, cat) + ((num,) if self.n_num else ())
        if self.use_graphs and b == self.max_b:
            self._graphed("d", self._d_body, ins, noise)
        else:
            self._d_body(*ins, noise)

    def _d_body(self, real, cat, *rest):
        num, noise = (rest[0], rest[1]) if self.n_num else (None, rest[0])
        ops, cfg, cr = self.ops, self.cfg, self.crit
        b = real.shape[0]
        self.arena_d.g.zero_()
        z = noise.randn((b, self.z_dim))
        fake = self.gen.forward(z, cat, num, self.lib, True)
        cr.load(0, b, real, cat, num)
        cr.load(1, b, fake, cat, num)
        cr.load(2, b, None, cat, num)
        masks, alpha = self._masks(noise, b, 3, alpha_before=2)
        ops.gp_interp(cr.x_seg(2, b), real, fake, alpha)
        val = cr.forward(b, 3, masks)
        ops.reduce_sum(self.red, S_REAL, val[:b])
        ops.reduce_sum(self.red, S_FAKE, val[b:2 * b])
        self._many(lambda: cr.backward_params(b, (-1.0 / b, 1.0 / b)),
                   lambda: cr.gp_backward(b, 2, float(cfg.get("lambda_gp", 10)), self.red, S_GP))
        self.arena_d.g.add_(self.arena_d.g2)
        self._adam(self.arena_d)
        ops.sum_parts(self.tot, self.red, self.slots, self.scales)

    # ---- generator update (cell 3, "Train Generator") ------------------------------------------
    def g_step(self, b, cat, num, noise):
        self._set_scales(b)
        self._begin_opt(self.arena_g, self.lr_g)
        ins = (cat,) + ((num,) if self.n_num else ())
        if self.use_graphs and b == self.max_b:
            self._graphed("g", self._g_body, ins, noise)
        else:
            self._g_body(*ins, noise)

    def _g_body(self, cat, *rest):
        num, noise = (rest[0], rest[1]) if self.n_num else (None, rest[0])
        ops, cr = self.ops, self.crit
        b = cat.shape[0]
        self.arena_g.g.zero_()
        z = noise.randn((b, self.z_dim))
        fake = self.gen.forward(z, cat, num, self.lib, True)
        cr.load(0, b, fake, cat, num)
        masks, _ = self._masks(noise, b, 1)
        val = cr.forward(b, 1, masks)
        ops.reduce_sum(self.red, S_GADV, val[:b])
        dfake = self.d_fake[:b]
        cr.backward_input(b, 0, -1.0 / b, dfake)
        self.gen.backward(dfake, self.lib)
        self._adam(self.arena_g)
        ops.sum_parts(self.tot, self.red, self.slots, self.scales)
        self.fake = fake

    def losses(self):
        """d_loss and gradient penalty of the last critic update, g_loss of the last generator update.  The
        generator's output is non-negative by construction (relu, then a positive scale), so the notebook's
        negative-value penalty and its two metrics are identically zero."""
        t = self.tot.cpu()
        lam = float(self.cfg.get("lambda_gp", 10))
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))  # noqa: E731
        return dict(d_loss=f32(float(t[0]) + float(t[1]) + float(t[2])), gp=float(t[2]) / lam, g_loss_main=float(t[3]),
                    g_loss=float(t[3]), neg_prop=0.0, neg_mag=0.0)

    def d_loss_now(self):
        """d_loss of the critic update just issued (reads the device)."""
        return self.losses()["d_loss"]

    # ---- per-update losses kept on the device (train_gan averages them per epoch, cell 3) -----------------
    LOG_ROWS = 4096

    def begin_log(self):
        if getattr(self, "_hist", None) is None:
            self._hist = self.ops.zeros(self.LOG_ROWS, N_SLOTS, dtype=torch.float64)
        self._kinds, self._done = [], []

    def _flush_log(self):
        t = self._hist[:len(self._kinds)].cpu()
        f32 = lambda v: float(torch.tensor(float(v), dtype=torch.float32))  # noqa: E731
        for row, kind in zip(t, self._kinds):
            self._done.append((kind, f32(row[0] + row[1] + row[2]) if kind == "d" else f32(row[3])))
        self._kinds = []

    def log(self, kind):
        """Record the loss of the update just issued ("d": critic, "g": generator) without waiting for it."""
        if len(self._kinds) == self.LOG_ROWS:
            self._flush_log()
        self._hist[len(self._kinds)].copy_(self.tot)
        self._kinds.append(kind)

    def end_log(self):
        """(critic losses, generator losses) recorded since begin_log(), in order."""
        self._flush_log()
        return [v for k, v in self._done if k == "d"], [v for k, v in self._done if k == "g"]

    def iteration(self, real, cat, num, noise):
        for _ in range(int(self.cfg["nb_critic"])):
            self.d_step(real, cat, num, noise)
        self.g_step(real.shape[0], cat, num, noise)

    def generate(self, z, cat, num, train=False):
        return self.gen.forward(z, cat, num, self.lib, train)

    def state_dict(self, name):
        return self.nets[name].state_dict()
