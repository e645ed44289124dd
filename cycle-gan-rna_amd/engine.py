"""Explicit forward / backward / update of one CycleGAN training iteration.

This is the host side of the hot path: the loop body of
train_enhanced_cycle_gan (reference GAN_functions.py:298-419) written as a fixed
sequence of kernel launches through `ops` (ops.py:HipOps), with hand-derived
backward passes instead of an autograd tape:

  * GenPass   - GeneExpressionGenerator.forward (:132-150) and its backward
  * DiscPass  - GeneExpressionDiscriminator.forward (:178-191), its backward,
                and the gradient penalty's first- and second-order chains
                (compute_gradient_penalty :194-204 followed by .backward())
  * StepEngine.d_step / g_step - :322-340 and :343-419, including the loss
                bookkeeping (:380-406, on the device) and Adam-amsgrad (:223-225)

Two sweeps the reference's autograd performs and then throws away are not
executed: the double backward from the penalty into the generators during the
critic step (their grads are zeroed at :343 before use) and the critic parameter
gradients during the generator step (zeroed at :322-323).  Losses, parameter
updates and BatchNorm running statistics are unaffected.

Parameters live in flat float32 arenas (one per optimizer) so that one fused
Adam launch updates a whole optimizer; each named parameter is a view.
"""
from __future__ import annotations

import gc
import math
import os
import sys
import warnings
import weakref
from collections import OrderedDict
from typing import Dict, Optional

import torch

EPI_NONE, EPI_LEAKY, EPI_MUL = 0, 1, 2
SEQ_LEN = 16
BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")

# reduction slots
(S_DM_REAL, S_DM_FAKE, S_DM_GP, S_DH_REAL, S_DH_FAKE, S_DH_GP,
 S_ADV_H, S_ADV_M, S_CYC_M, S_CYC_H, S_CT_H, S_CT_M, S_FM_H, S_FM_M, S_KL_H, S_KL_M, S_IDT_M, S_IDT_H,
 N_SLOTS) = range(19)
N_GSLOTS = 12           # slots of the generator step (S_ADV_H ... S_IDT_H)
MAX_PARTS = 1024
PIN_SLOTS = 64          # staging slots for the {lr, step} uploads of optimizer steps in flight


def _pad4(n):
    return (n + 3) // 4 * 4


class Arena:
    """Flat float32 storage for the parameters of one optimizer (+ grads and
    Adam state), with named views.  Every tensor starts on a 16-byte boundary;
    the padding stays zero (zero gradient keeps Adam's update at zero)."""

    def __init__(self, ops, named_shapes, second_grad=False):
        self.ops = ops
        self.offsets = OrderedDict()
        off = 0
        for key, shape in named_shapes:
            n = int(math.prod(shape)) if len(shape) else 1
            self.offsets[key] = (off, n, tuple(shape))
            off += _pad4(n)
        self.n = off
        self.p = ops.zeros(off)
        self.g = ops.zeros(off)
        # second gradient buffer: lets two backward chains of one network run side by side without both
        # accumulating into the same tensors; added to `g` before the optimizer step
        self.g2 = ops.zeros(off) if second_grad else None
        self.m = ops.zeros(off)
        self.v = ops.zeros(off)
        self.vmax = ops.zeros(off)
        self.lr_step = ops.zeros(2, dtype=torch.float64)
        self.step = 0
        # bf16 mode: a bf16 image of the parameters, rewritten by every Adam step, that the bf16 products read the
        # weights from (call sync_shadow() after writing parameters any other way)
        self.p16 = None
        if getattr(ops, "precision", "fp32") == "bf16" and hasattr(ops, "register_shadow"):
            self.p16 = ops.zeros(off, dtype=torch.bfloat16)
            weakref.finalize(self, ops.unregister_shadow, ops.register_shadow(self.p, self.p16))
        self.net_range = {}
        for (net, _), (o, n, _) in self.offsets.items():
            lo, hi = self.net_range.get(net, (o, o))
            self.net_range[net] = (min(lo, o), max(hi, o + _pad4(n)))

    def sync_shadow(self):
        if self.p16 is not None:
            self.ops.refresh_shadow(self.p, self.p16)

    def net_grad(self, net):
        lo, hi = self.net_range[net]
        return self.g[lo:hi]

    def view(self, buf, key):
        off, n, shape = self.offsets[key]
        return buf[off:off + n].view(shape)

    def flat(self, buf, key_first, n):
        off = self.offsets[key_first][0]
        return buf[off:off + n]


def split_state(sd):
    params = OrderedDict((k, v) for k, v in sd.items() if not k.endswith(BUFFER_SUFFIXES))
    buffers = OrderedDict((k, v) for k, v in sd.items() if k.endswith(BUFFER_SUFFIXES))
    return params, buffers


def _arena_order(keys):
    """state_dict order, with the (never used) sparse-attention gate moved to the
    end so that each attention block's six tensors stay contiguous, and log_var's weight / bias placed right
    after mu's: the two heads of the encoder (GAN_functions.py:97-98,:139) then form one (2L, H) matrix and one
    2L bias, and each of their three products (forward, weight gradient, input gradient) is a single GEMM."""
    gate = [k for k in keys if ".gate." in k]
    out = []
    for k in keys:
        if ".gate." in k or k in ("log_var.weight", "log_var.bias"):
            continue
        if k == "mu.bias" and "log_var.weight" in keys:
            out.append("log_var.weight")
        out.append(k)
        if k == "mu.bias" and "log_var.bias" in keys:
            out.append("log_var.bias")
    return out + gate


def _emb_tables(P):
    """(key, width) of the embedding tables, one per categorical covariate, in the order the reference
    concatenates them (GAN_functions.py:85,:135,:160,:182)."""
    out, i = [], 0
    while f"embeddings.{i}.weight" in P:
        out.append((f"embeddings.{i}.weight", P[f"embeddings.{i}.weight"].shape[1]))
        i += 1
    return out


def _zero(ops, view):
    """view <- 0 through the backend when it keeps bf16 images of its workspaces (HipOps.zero), else in place."""
    if hasattr(ops, "zero"):
        ops.zero(view)
    else:
        view.zero_()


def _embed_gather(ops, out, P, labels):
    """out (b, sum of widths) <- [table_i[labels[:, i]] for every covariate i]."""
    off = 0
    for i, (k, w) in enumerate(_emb_tables(P)):
        ops.embed_gather(out[:, off:off + w], P[k], labels[:, i:i + 1])
        off += w


def _embed_scatter(ops, G, dout, labels, accumulate):
    off = 0
    for i, (k, w) in enumerate(_emb_tables(G)):
        ops.embed_scatter(G[k], dout[:, off:off + w], labels[:, i:i + 1], accumulate)
        off += w


class Net:
    """Device-side parameters (arena views), gradients and buffers of one network."""

    def __init__(self, name, arena, keys, buffers, ops):
        self.name = name
        self.P = OrderedDict((k, arena.view(arena.p, (name, k))) for k in keys)
        self.G = OrderedDict((k, arena.view(arena.g, (name, k))) for k in keys)
        self.G2 = OrderedDict((k, arena.view(arena.g2, (name, k))) for k in keys) if arena.g2 is not None else None
        self.B = OrderedDict()
        self.nbt = OrderedDict()
        for k, v in buffers.items():
            if k.endswith("num_batches_tracked"):
                self.nbt[k] = int(v)
            else:
                self.B[k] = ops.to_device(v.detach().clone().float())
        self.arena = arena

    def state_dict(self):
        sd = OrderedDict()
        for k, v in self.P.items():
            sd[k] = v.detach().cpu().clone()
        for k, v in self.B.items():
            sd[k] = v.detach().cpu().clone()
        for k, v in self.nbt.items():
            sd[k] = torch.tensor(v, dtype=torch.long)
        return sd

    def load(self, sd):
        for k, v in sd.items():
            if k in self.P:
                self.P[k].copy_(v.reshape(self.P[k].shape))
            elif k in self.B:
                self.B[k].copy_(v)
            elif k in self.nbt:
                self.nbt[k] = int(v)
        self.arena.sync_shadow()


# --------------------------------------------------------------------------- #
class GenPass:
    """Workspace + forward/backward of one call of a generator on a batch."""

    def __init__(self, engine, net: Net, in_dim, out_dim, max_b):
        ops, cfg = engine.ops, engine.cfg
        # no reference back to the engine: an engine must die by reference count when its owner drops it, not in a
        # later garbage-collection pass (which could run inside another engine's graph capture and destroy HIP graphs there)
        self.ops, self.net, self.cfg = ops, net, cfg
        self.in_dim, self.out_dim = in_dim, out_dim
        self.hidden = list(cfg["hidden_dims"])
        self.L = cfg["latent_dim"]
        self.nb = cfg["residual_blocks"]
        self.heads = cfg["num_attention_heads"]
        self.E = engine.emb_dim
        self.counts = cfg.get("preprocess_type", "counts") == "counts"
        self.D = self.L // SEQ_LEN          # token width (GAN_functions.py:82)
        assert self.D * SEQ_LEN == self.L, "latent_dim must be divisible by seq_len"
        # 16-wide tokens run the fused token kernels; other widths take the ResidualBlock through the GEMM and
        # BatchNorm kernels
        self.fused_tok = self.D == 16
        Z, B, H, L, D = ops.zeros, max_b, self.hidden, self.L, self.D
        # workspaces that only gemm / bn_fwd* / bn_eval / bn_bwd write and that products read next: in bf16 mode the
        # HIP backend keeps a bf16 image of them (ops.shadowed), which those products read instead
        Zs = getattr(ops, "shadowed", Z)
        T = B * SEQ_LEN
        self.enc_h0 = [Z(B, h) for h in H]
        self.enc_a = [Zs(B, h) for h in H]
        self.enc_stats = [(Z(h), Z(h)) for h in H]
        self.ml = Z(B, 2 * L)
        self.z = [Z(B, L) for _ in range(self.nb + 1)]
        self.t1 = [Z(T, D) for _ in range(self.nb)]
        self.r = [Z(T, D) for _ in range(self.nb)]
        self.t2 = [Z(T, D) for _ in range(self.nb)]
        self.y = [Z(T, D) for _ in range(self.nb)]
        self.u = Z(T, D)
        self.rb_stats = [((Z(D), Z(D)), (Z(D), Z(D))) for _ in range(self.nb)]
        self.keep = [None] * self.nb
        # 16-wide tokens with 4 heads: the block's last step, the attention and the next block's first step in one launch
        self.fused_block = self.fused_tok and hasattr(ops, "attn_block_ok") and ops.attn_block_ok(self.D, self.heads)
        if self.fused_tok:
            nblk = max(ops.tok_blocks(T), 1)
            if self.fused_block:
                nblk = max(nblk, ops.attn_block_blocks(B), ops.attn_bwd_blocks(B))
            self.tok_stat = [Z(nblk * 32, dtype=torch.float64) for _ in range(4)]     # per-workgroup column sums
            self.tok_wpart = [Z(nblk * 256), Z(nblk * 256)]                           # per-workgroup 16x16 weight grads
        self.dec_in = Z(B, L + self.E)
        self.dec_h0 = [Z(B, h) for h in H]
        self.dec_a = [Zs(B, h) for h in H]
        self.dec_stats = [(Z(h), Z(h)) for h in H]
        # without library-size normalisation the last product's result is the generated batch itself, which the
        # other generator's first layer (cycle pass) and its weight gradient read
        self.out_raw = Z(B, out_dim) if self.counts else Zs(B, out_dim)
        self.out = Z(B, out_dim) if self.counts else self.out_raw
        self.rowsum = Z(B)
        # backward scratch
        self.d_out_raw = Z(B, out_dim) if self.counts else None
        self.d_act = [Z(B, h) for h in H]
        self.d_h0 = [Zs(B, h) for h in H]
        self.d_dec_in = Z(B, L + self.E)
        self.dz = [Z(B, L), Z(B, L)]
        self.d16 = [Z(T, D), Z(T, D)]
        self.dml = Z(B, 2 * L)
        self.attn_ws = Z(max(ops.attn_bwd_ws(B, D, self.heads), 1))   # raises for a geometry that is not built
        # the parameter-gradient partials of every block kept apart and reduced in two launches at the end of the
        # latent chain's backward instead of two per block (needs the blocks' tensors at equal distances in the arena)
        self.defer_reduce = False
        if self.fused_block and hasattr(ops, "latent_param_reduce") and not os.environ.get("CGRNA_NO_DEFERRED_REDUCE"):
            def steps(ts):
                d = {ts[k + 1].data_ptr() - ts[k].data_ptr() for k in range(len(ts) - 1)}
                return len(d) <= 1 and all(x % 16 == 0 for x in d)
            pres = [f"residual_attention_blocks.{bi}" for bi in range(self.nb)]
            w2 = [net.G[q + ".0.block.3.weight"] for q in pres]
            w1 = [net.G[q + ".0.block.0.weight"] for q in pres]
            same = len({w2[k + 1].data_ptr() - w2[k].data_ptr() for k in range(self.nb - 1)} |
                       {w1[k + 1].data_ptr() - w1[k].data_ptr() for k in range(self.nb - 1)}) <= 1
            if steps([self._attn("g", bi) for bi in range(self.nb)]) and steps(w2) and steps(w1) and same:
                self.defer_reduce = True
                self.attn_ws_b = Z(self.nb, self.attn_ws.numel())
                self.tok_wpart_b = [Z(self.nb, self.tok_wpart[0].numel()), Z(self.nb, self.tok_wpart[1].numel())]
        self.eps = None
        self.labels = None
        self.x = None
        self.b = 0
        # bf16 mode: a bf16 image of the input batch for the first layer's product and its weight gradient (unless the
        # batch already has one: the other generator's output in a cycle pass)
        self.x16 = None
        if getattr(ops, "precision", "fp32") == "bf16" and hasattr(ops, "image_of"):
            self.x16 = ops.zeros(B * _pad4(in_dim), dtype=torch.bfloat16)
        self.x_img = None
        self._own_encoder = (self.enc_h0, self.enc_a, self.enc_stats, self.ml)
        # [mu; log_var] as one matrix (see _arena_order)
        ar, hl = net.arena, H[-1]
        self.W_ml = ar.flat(ar.p, (net.name, "mu.weight"), 2 * L * hl).view(2 * L, hl)
        self.b_ml = ar.flat(ar.p, (net.name, "mu.bias"), 2 * L)
        self.GW_ml = ar.flat(ar.g, (net.name, "mu.weight"), 2 * L * hl).view(2 * L, hl)
        self.Gb_ml = ar.flat(ar.g, (net.name, "mu.bias"), 2 * L)
        assert self.W_ml[L:].data_ptr() == net.P["log_var.weight"].data_ptr() and \
            self.b_ml[L:].data_ptr() == net.P["log_var.bias"].data_ptr()

    # attention parameter block of residual block `bi` (params or grads)
    def _attn(self, which, bi):
        first = f"residual_attention_blocks.{bi}.1.attention.in_proj_weight"
        return self.net.arena.flat(getattr(self.net.arena, which), (self.net.name, first), 4 * self.D * self.D + 6 * self.D)

    def _bn(self, prefix):
        P, Bf = self.net.P, self.net.B
        return (P[prefix + ".weight"], P[prefix + ".bias"], Bf[prefix + ".running_mean"], Bf[prefix + ".running_var"])

    def _linear_bn_relu(self, out_a, out_h0, stats, a, lin, bn, train, a16=None):
        ops, P = self.ops, self.net.P
        gamma, beta, rm, rv = self._bn(bn)
        # when the product splits K, the BatchNorm adds the partial sums itself while it loads its input: one launch
        # and one pass over the pre-activation less
        slabs = train and ops.bn_slabs_ok(*out_h0.shape)
        kw = {} if a16 is None else dict(a16=a16)
        n = ops.gemm(out_h0, a, P[lin + ".weight"], False, True, bias=P[lin + ".bias"], keep_slabs=slabs, **kw)
        if slabs and n > 1:
            ops.bn_fwd_slabs(out_a, out_h0, n, P[lin + ".bias"], gamma, beta, stats[0], stats[1], rm, rv, 0.0)
            self.net.nbt[bn + ".num_batches_tracked"] += 1
        elif train:
            ops.bn_fwd(out_a, out_h0, gamma, beta, stats[0], stats[1], rm, rv, 0.0)
            self.net.nbt[bn + ".num_batches_tracked"] += 1
        else:
            ops.bn_eval(out_a, out_h0, gamma, beta, rm, rv, 0.0)

    def forward(self, x, labels, noise, train=True, encoder_of=None):
        """encoder_of: a pass of the SAME network (this one included) whose last train-mode forward saw the same
        batch with the same weights - the critic step's call of the generator, when the generator step of the same
        iteration calls it again (GAN_functions.py:326-327, :344-345; only the critics change in between).  The
        encoder and the two heads are deterministic, so their activations, pre-activations and batch statistics are
        taken over from that pass (its buffers become this pass's, for the backward pass too) instead of being
        computed again; what the second call would do to the running statistics of the encoder's BatchNorm layers
        is applied from the stored batch statistics.  Everything after the heads draws its own noise as usual."""
        ops, P, L = self.ops, self.net.P, self.L
        b = x.shape[0]
        nh = len(self.hidden)
        if encoder_of is not None:
            src = encoder_of
            assert train and src.train and src.b == b and src.net is self.net
            self.enc_h0, self.enc_a, self.enc_stats, self.ml = src.enc_h0, src.enc_a, src.enc_stats, src.ml
            self.b, self.x, self.labels, self.train, self.x_img = b, src.x, labels, train, src.x_img
            for li in range(nh):
                bn = f"encoder.{3 * li + 1}"
                _, _, rm, rv = self._bn(bn)
                ops.bn_running_update(rm, rv, self.enc_stats[li][0], self.enc_stats[li][1], b)
                self.net.nbt[bn + ".num_batches_tracked"] += 1
            ml = self.ml[:b]
        else:
            self.enc_h0, self.enc_a, self.enc_stats, self.ml = self._own_encoder
            self.b, self.x, self.labels, self.train = b, x, labels, train
            a = x
            self.x_img = None
            if self.x16 is not None and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and \
                    x.stride(0) <= _pad4(self.in_dim):
                self.x_img = ops.image_of(x, self.x16)
            for li in range(nh):
                self._linear_bn_relu(self.enc_a[li][:b], self.enc_h0[li][:b], self.enc_stats[li], a,
                                     f"encoder.{3 * li}", f"encoder.{3 * li + 1}", train,
                                     a16=self.x_img if li == 0 else None)
                a = self.enc_a[li][:b]
            ml = self.ml[:b]
            ops.gemm(ml, a, self.W_ml, False, True, bias=self.b_ml)            # mu | log_var, :139
        self.eps = noise.randn((b, L))
        ops.reparam_fwd(self.z[0][:b], ml, self.eps)
        t = b * SEQ_LEN
        all_keep = None
        if train and getattr(noise, "stacked_draws", False):     # one draw for the attention dropout of every block
            all_keep = noise.keep((self.nb, b * self.heads, SEQ_LEN, SEQ_LEN), 0.1)
        for bi in range(self.nb):
            pre = f"residual_attention_blocks.{bi}"
            zin = self.z[bi][:b].view(t, self.D)
            t1, r, t2, y, u = self.t1[bi][:t], self.r[bi][:t], self.t2[bi][:t], self.y[bi][:t], self.u[:t]
            (s1, s2) = self.rb_stats[bi]
            g1, b1, rm1, rv1 = self._bn(pre + ".0.block.1")
            g2, b2, rm2, rv2 = self._bn(pre + ".0.block.4")
            W1, c1 = P[pre + ".0.block.0.weight"], P[pre + ".0.block.0.bias"]
            W2, c2 = P[pre + ".0.block.3.weight"], P[pre + ".0.block.3.bias"]
            if train and not self.fused_tok:
                ops.gemm(t1, zin, W1, False, True, bias=c1)
                ops.bn_fwd(r, t1, g1, b1, s1[0], s1[1], rm1, rv1, 0.0)
                ops.gemm(t2, r, W2, False, True, bias=c2)
                ops.bn_fwd(u, t2, g2, b2, s2[0], s2[1], rm2, rv2, 1.0)
                ops.axpby(y, 1.0, zin, 1.0, u)
                self.net.nbt[pre + ".0.block.1.num_batches_tracked"] += 1
                self.net.nbt[pre + ".0.block.4.num_batches_tracked"] += 1
            elif train and self.fused_block:
                # two launches per block: BatchNorm+ReLU+Linear | BatchNorm+residual, attention+LayerNorm and the next
                # block's first Linear (the first block's own first Linear is a launch of its own)
                sa, sb = self.tok_stat[0], self.tok_stat[1]
                n_cells_blocks = ops.attn_block_blocks(b)
                if bi == 0:
                    ops.tok_lin_fwd(t1, sa, zin, W1, c1)
                    ops.tok_bn_lin_fwd(r, t2, sb, t1, sa, g1, b1, s1[0], s1[1], rm1, rv1, W2, c2)
                else:
                    ops.tok_bn_lin_fwd_from(r, t2, sb, t1, sa, n_cells_blocks, g1, b1, s1[0], s1[1], rm1, rv1, W2, c2)
                self.net.nbt[pre + ".0.block.1.num_batches_tracked"] += 1
                self.net.nbt[pre + ".0.block.4.num_batches_tracked"] += 1
                if all_keep is not None:
                    self.keep[bi] = all_keep[bi]
                else:
                    self.keep[bi] = noise.keep((b * self.heads, SEQ_LEN, SEQ_LEN), 0.1)
                nxt = None
                if bi + 1 < self.nb:
                    npre = f"residual_attention_blocks.{bi + 1}"
                    nxt = (self.t1[bi + 1][:t], sa, P[npre + ".0.block.0.weight"], P[npre + ".0.block.0.bias"])
                ops.attn_block_fwd(self.z[bi + 1][:b].view(t, self.D), y, t2, zin, sb, g2, b2, s2[0], s2[1], rm2, rv2,
                                   self.keep[bi], self._attn("p", bi), self.heads, nxt=nxt)
                continue
            elif train:     # three fused launches: Linear | BatchNorm+ReLU+Linear | BatchNorm+residual
                sa, sb = self.tok_stat[0], self.tok_stat[1]
                ops.tok_lin_fwd(t1, sa, zin, W1, c1)
                ops.tok_bn_lin_fwd(r, t2, sb, t1, sa, g1, b1, s1[0], s1[1], rm1, rv1, W2, c2)
                ops.tok_bn_res_fwd(y, t2, sb, g2, b2, s2[0], s2[1], rm2, rv2, zin)
                self.net.nbt[pre + ".0.block.1.num_batches_tracked"] += 1
                self.net.nbt[pre + ".0.block.4.num_batches_tracked"] += 1
            else:
                ops.gemm(t1, zin, W1, False, True, bias=c1)
                ops.bn_eval(r, t1, g1, b1, rm1, rv1, 0.0)
                ops.gemm(t2, r, W2, False, True, bias=c2)
                ops.bn_eval(u, t2, g2, b2, rm2, rv2, 1.0)
                ops.axpby(y, 1.0, zin, 1.0, u)
            if all_keep is not None:
                self.keep[bi] = all_keep[bi]
            else:
                self.keep[bi] = noise.keep((b * self.heads, SEQ_LEN, SEQ_LEN), 0.1) if train else None
            ops.attn_fwd(self.z[bi + 1][:b].view(t, self.D), y, self.keep[bi], self._attn("p", bi), self.heads)
        dec_in = self.dec_in[:b]
        ops.axpby(dec_in[:, :L], 1.0, self.z[self.nb][:b])
        if labels is not None:
            _embed_gather(ops, dec_in[:, L:], P, labels)
        else:
            _zero(ops, dec_in[:, L:])
        a = dec_in
        for li in range(nh):
            self._linear_bn_relu(self.dec_a[li][:b], self.dec_h0[li][:b], self.dec_stats[li], a,
                                 f"decoder.{3 * li}", f"decoder.{3 * li + 1}", train)
            a = self.dec_a[li][:b]
        last = f"decoder.{3 * nh}"
        ops.gemm(self.out_raw[:b], a, P[last + ".weight"], False, True, bias=P[last + ".bias"])
        if self.counts:
            ops.libsize_fwd(self.out[:b], self.out_raw[:b], self.rowsum[:b], self.cfg["library_size"])
        return self.out[:b]

    def mu(self):
        return self.ml[:self.b, :self.L]

    def log_var(self):
        return self.ml[:self.b, self.L:]

    def _dgrad_to_bn(self, d_prev, d, W):
        """d_prev = d W, the gradient that the BatchNorm backward of the layer below reads next on this stream (nothing
        else reads it).  When the product splits K its partial sums stay in the stream's scratch and that BatchNorm
        adds them while it loads (`_slabs_for` tells it): one launch and one pass over the gradient less."""
        ops = self.ops
        ok = getattr(ops, "bn_bwd_slabs_ok", None)
        keep = ok is not None and ok(*d_prev.shape) and not os.environ.get("CGRNA_NO_BWD_SLABS")
        n = ops.gemm(d_prev, d, W, False, False, keep_slabs=True) if keep else ops.gemm(d_prev, d, W, False, False)
        self._slabs_for = (d_prev.data_ptr(), n) if keep and n > 1 else None

    _slabs_for = None

    def _bn_linear_bwd(self, d_act, d_h0, h0, act, stats, a_prev, lin, bn, d_prev, prev16=None):
        """Backward of Linear -> BatchNorm -> ReLU; d_prev (may be None) receives the input gradient."""
        ops, P, G = self.ops, self.net.P, self.net.G
        pending, self._slabs_for = self._slabs_for, None
        if pending is not None and pending[0] == d_act.data_ptr():
            ops.bn_bwd_slabs(d_h0, pending[1], h0, act, None, stats[0], stats[1], P[bn + ".weight"], G[bn + ".weight"],
                             G[bn + ".bias"], True, 0.0)
        else:
            assert pending is None, "a product left its partial sums for a BatchNorm backward that did not follow"
            ops.bn_bwd(d_h0, d_act, h0, act, None, stats[0], stats[1], P[bn + ".weight"], G[bn + ".weight"],
                       G[bn + ".bias"], True, 0.0)
        kw = {} if prev16 is None else dict(b16=prev16)
        ops.gemm(G[lin + ".weight"], d_h0, a_prev, True, False, accumulate=True, **kw)
        # the Linear bias feeds a BatchNorm: its gradient is identically zero and is left at zero
        if d_prev is not None:
            if d_prev.data_ptr() in self._bn_inputs:
                self._dgrad_to_bn(d_prev, d_h0, P[lin + ".weight"])
            else:
                ops.gemm(d_prev, d_h0, P[lin + ".weight"], False, False)

    def backward(self, d_out, need_dx=False, dx_out=None, dx_accumulate=False, kl=None, d_mu=None, d_lv=None,
                 hook=None):
        """d_out: gradient w.r.t. the returned output.  kl = (coef, idx, weight)
        adds the KL term's gradient; d_mu / d_lv add gradients arriving at the
        returned mu / log_var.  Parameter gradients are accumulated into the
        arena; returns dx (gradient w.r.t. the input) when need_dx.
        hook (optional) is called with "decoder" once every decoder.* gradient (and the embedding table's) has
        received this pass's contribution, with "middle" once everything but the encoder's first layer has, and with
        "first" at the end - for a caller that starts the optimizer step on the parameters that are done."""
        self.backward_decoder(d_out)
        if hook is not None:
            hook("decoder")
        dx = self.backward_rest(need_dx, dx_out, dx_accumulate, kl, d_mu, d_lv, hook=hook)
        if hook is not None:
            hook("first")
        return dx

    def backward_decoder(self, d_out):
        """First half of backward(): the decoder layers and the embedding table.  After it the gradients of every
        decoder.* tensor are complete (a data-parallel run starts averaging them while backward_rest() runs)."""
        ops, P, G, L, b = self.ops, self.net.P, self.net.G, self.L, self.b
        nh = len(self.hidden)
        d = d_out
        if self.counts:
            ops.libsize_bwd(self.d_out_raw[:b], d_out, self.out_raw[:b], self.rowsum[:b], self.cfg["library_size"])
            d = self.d_out_raw[:b]
        last = f"decoder.{3 * nh}"
        ops.gemm(G[last + ".weight"], d, self.dec_a[nh - 1][:b], True, False, accumulate=True)
        ops.colsum(G[last + ".bias"], d, 1.0, True)
        # gradients that only a BatchNorm backward reads (d_act of every hidden layer; not the decoder input's)
        self._bn_inputs = {self.d_act[li][:b].data_ptr() for li in range(nh)}
        self._dgrad_to_bn(self.d_act[nh - 1][:b], d, P[last + ".weight"])
        for li in reversed(range(nh)):
            a_prev = self.dec_a[li - 1][:b] if li > 0 else self.dec_in[:b]
            d_prev = self.d_act[li - 1][:b] if li > 0 else self.d_dec_in[:b]
            self._bn_linear_bwd(self.d_act[li][:b], self.d_h0[li][:b], self.dec_h0[li][:b], self.dec_a[li][:b],
                                self.dec_stats[li], a_prev, f"decoder.{3 * li}", f"decoder.{3 * li + 1}", d_prev)
        ddi = self.d_dec_in[:b]
        if self.labels is not None:
            _embed_scatter(ops, G, ddi[:, L:], self.labels, True)

    def backward_rest(self, need_dx=False, dx_out=None, dx_accumulate=False, kl=None, d_mu=None, d_lv=None, hook=None):
        """Second half of backward(): latent blocks, the two heads, the encoder."""
        ops, P, G, L, b = self.ops, self.net.P, self.net.G, self.L, self.b
        nh = len(self.hidden)
        t = b * SEQ_LEN
        ddi = self.d_dec_in[:b]
        cur = 0
        ops.axpby(self.dz[cur][:b], 1.0, ddi[:, :L])
        dz = self.dz[cur][:b].view(t, self.D)
        for bi in reversed(range(self.nb)):
            pre = f"residual_attention_blocks.{bi}"
            (s1, s2) = self.rb_stats[bi]
            dy = self.dz[1 - cur][:b].view(t, self.D)
            zin = self.z[bi][:b].view(t, self.D)
            defer = self.defer_reduce
            if self.fused_block:        # the attention's backward also sums what the BatchNorm backward below needs
                s3 = self.tok_stat[2]
                ops.attn_block_bwd(dy, self._attn("g", bi), self.attn_ws_b[bi] if defer else self.attn_ws, dz,
                                   self.y[bi][:t], self.keep[bi], self._attn("p", bi), self.heads, True,
                                   self.t2[bi][:t], s2[0], s2[1], s3, reduce_now=not defer)
            else:
                ops.attn_bwd(dy, self._attn("g", bi), self.attn_ws, dz, self.y[bi][:t], self.keep[bi], self._attn("p", bi),
                             self.heads, True)
            if self.fused_tok:
                s3, s4 = self.tok_stat[2], self.tok_stat[3]
                wp2, wp1 = (self.tok_wpart_b[0][bi], self.tok_wpart_b[1][bi]) if defer else self.tok_wpart
                g1buf = self.d16[0][:t]
                mid = (self.t2[bi][:t], self.r[bi][:t], self.t1[bi][:t])
                if self.fused_block:
                    ops.tok_bwd_mid_from(g1buf, s4, wp2, G[pre + ".0.block.4.weight"], G[pre + ".0.block.4.bias"], dy,
                                         *mid, s3, ops.attn_bwd_blocks(b), s2[0], s2[1],
                                         P[pre + ".0.block.4.weight"], P[pre + ".0.block.3.weight"], s1[0], s1[1])
                else:
                    ops.tok_bn_bwd_stats(s3, dy, self.t2[bi][:t], s2[0], s2[1])
                    ops.tok_bwd_mid(g1buf, s4, wp2, G[pre + ".0.block.4.weight"], G[pre + ".0.block.4.bias"], dy,
                                    *mid, s3, s2[0], s2[1],
                                    P[pre + ".0.block.4.weight"], P[pre + ".0.block.3.weight"], s1[0], s1[1])
                ops.tok_bwd_in(dy, wp1, G[pre + ".0.block.1.weight"], G[pre + ".0.block.1.bias"], g1buf,
                               self.t1[bi][:t], zin, dy, s4, s1[0], s1[1], P[pre + ".0.block.1.weight"],
                               P[pre + ".0.block.0.weight"])
                if not defer:
                    ops.tok_wgrad_reduce(G[pre + ".0.block.3.weight"], G[pre + ".0.block.0.weight"], wp2, wp1, t)
            else:           # the same chain through the general kernels; dy ends as the gradient w.r.t. zin
                dt, dr = self.d16[0][:t], self.d16[1][:t]
                ops.bn_bwd(dt, dy, self.t2[bi][:t], None, None, s2[0], s2[1], P[pre + ".0.block.4.weight"],
                           G[pre + ".0.block.4.weight"], G[pre + ".0.block.4.bias"], True, 1.0)
                ops.gemm(G[pre + ".0.block.3.weight"], dt, self.r[bi][:t], True, False, accumulate=True)
                ops.gemm(dr, dt, P[pre + ".0.block.3.weight"], False, False)
                ops.bn_bwd(dt, dr, self.t1[bi][:t], self.r[bi][:t], None, s1[0], s1[1], P[pre + ".0.block.1.weight"],
                           G[pre + ".0.block.1.weight"], G[pre + ".0.block.1.bias"], True, 0.0)
                ops.gemm(G[pre + ".0.block.0.weight"], dt, zin, True, False, accumulate=True)
                ops.gemm(dy, dt, P[pre + ".0.block.0.weight"], False, False, accumulate=True)
            cur = 1 - cur
            dz = dy
        if self.defer_reduce:
            pres = [f"residual_attention_blocks.{bi}" for bi in range(self.nb)]
            done = ops.latent_param_reduce([self._attn("g", bi) for bi in range(self.nb)], self.attn_ws_b,
                                           [G[q + ".0.block.3.weight"] for q in pres],
                                           [G[q + ".0.block.0.weight"] for q in pres], self.tok_wpart_b[0],
                                           self.tok_wpart_b[1], b)
            assert done
        dz0 = self.dz[cur][:b]
        dml = self.dml[:b]
        coef, idx, w = kl if kl is not None else (None, 0, 0.0)
        ops.reparam_bwd(dml, dz0, self.ml[:b], self.eps, coef, idx, w)
        if d_mu is not None:
            ops.axpby(dml[:, :L], 1.0, dml[:, :L], 1.0, d_mu)
        if d_lv is not None:
            ops.axpby(dml[:, L:], 1.0, dml[:, L:], 1.0, d_lv)
        a_enc = self.enc_a[nh - 1][:b]
        ops.gemm(self.GW_ml, dml, a_enc, True, False, accumulate=True)
        ops.colsum(self.Gb_ml, dml, 1.0, True)
        da = self.d_act[nh - 1][:b]
        self._bn_inputs = {self.d_act[li][:b].data_ptr() for li in range(nh)}
        self._dgrad_to_bn(da, dml, self.W_ml)
        dx = None
        for li in reversed(range(nh)):
            if li == 0 and hook is not None:
                hook("middle")
            a_prev = self.enc_a[li - 1][:b] if li > 0 else self.x
            d_prev = self.d_act[li - 1][:b] if li > 0 else None
            self._bn_linear_bwd(self.d_act[li][:b], self.d_h0[li][:b], self.enc_h0[li][:b], self.enc_a[li][:b],
                                self.enc_stats[li], a_prev, f"encoder.{3 * li}", f"encoder.{3 * li + 1}", d_prev,
                                prev16=self.x_img if li == 0 else None)
        if need_dx:
            dx = dx_out if dx_out is not None else ops.zeros(b, self.in_dim)
            ops.gemm(dx, self.d_h0[0][:b], P["encoder.0.weight"], False, False, accumulate=dx_accumulate)
        return dx


# --------------------------------------------------------------------------- #
class DiscPass:
    """Workspace + forward/backward of one call of a critic on a batch."""

    def __init__(self, engine, net: Net, in_dim, max_b):
        ops = engine.ops
        self.ops, self.net = ops, net
        self.Gd, self.E = in_dim, engine.emb_dim
        Z, B = ops.zeros, max_b
        self.emb = Z(B, self.E)
        self.h0, self.a1, self.pm1 = Z(B, 1024), Z(B, 1024), Z(B, 1024)
        self.stats = (Z(1024), Z(1024))
        self.a2, self.pm2 = Z(B, 512), Z(B, 512)
        self.a3, self.pm3 = Z(B, 256), Z(B, 256)
        self.val = Z(B, 1)
        # backward scratch
        self.dval = Z(B, 1)
        self.ones = ops.to_device(torch.ones(B, 1))
        self.g256, self.g512, self.g1024a, self.g1024b = Z(B, 256), Z(B, 512), Z(B, 1024), Z(B, 1024)
        self.h256, self.h512, self.h1024a, self.h1024b = Z(B, 256), Z(B, 512), Z(B, 1024), Z(B, 1024)
        self.demb = Z(B, self.E)
        self.ux = Z(B, in_dim)
        self.rowpen = Z(B)
        self.ubar, self.mbar = Z(1024), Z(1024)
        self.x = None
        self.labels = None
        self.b = 0

    def forward(self, x, labels, noise, train=True):
        ops, P, Bf, Gd = self.ops, self.net.P, self.net.B, self.Gd
        b = x.shape[0]
        self.b, self.x, self.labels = b, x, labels
        W1 = P["model.0.weight"]
        emb = self.emb[:b]
        if labels is not None:
            _embed_gather(ops, emb, P, labels)
        else:
            _zero(ops, emb)
        h0 = self.h0[:b]
        ops.gemm(h0, x, W1[:, :Gd], False, True, bias=P["model.0.bias"])
        ops.gemm(h0, emb, W1[:, Gd:], False, True, accumulate=True)
        if train:
            m1 = noise.keep((b, 1024), 0.3)
            ops.bn_fwd(self.a1[:b], h0, P["model.1.weight"], P["model.1.bias"], self.stats[0], self.stats[1],
                       Bf["model.1.running_mean"], Bf["model.1.running_var"], 0.2, m1, self.pm1[:b])
            self.net.nbt["model.1.num_batches_tracked"] += 1
            m2 = noise.keep((b, 512), 0.3)
        else:
            ops.bn_eval(self.a1[:b], h0, P["model.1.weight"], P["model.1.bias"], Bf["model.1.running_mean"],
                        Bf["model.1.running_var"], 0.2)
            m2 = None
        ops.gemm(self.a2[:b], self.a1[:b], P["model.4.weight"], False, True, bias=P["model.4.bias"], epi=EPI_LEAKY,
                 E=m2, aux=self.pm2[:b], slope=0.2)
        ops.gemm(self.a3[:b], self.a2[:b], P["model.7.weight"], False, True, bias=P["model.7.bias"], epi=EPI_LEAKY,
                 E=None, aux=self.pm3[:b], slope=0.2)
        ops.gemm(self.val[:b], self.a3[:b], P["model.9.weight"], False, True, bias=P["model.9.bias"])
        return self.val[:b], self.a3[:b]

    def backward(self, coef=None, idx=0, scale=0.0, dfeat=None, param_grads=True, dx_out=None):
        """Gradient of  coef[idx]*scale*sum(validity) + <dfeat, features>.
        param_grads: accumulate into the critic's gradient arena (critic step);
        dx_out: accumulate the input gradient there (generator step)."""
        ops, P, G, Gd, b = self.ops, self.net.P, self.net.G, self.Gd, self.b
        W1 = P["model.0.weight"]
        dh3 = self.g256[:b]
        if coef is not None:
            dval = self.dval[:b]
            ops.fill_coef(dval, coef, idx, scale)
            if param_grads:
                ops.gemm(G["model.9.weight"], dval, self.a3[:b], True, False, accumulate=True)
                ops.colsum(G["model.9.bias"], dval, 1.0, True)
            ops.gemm(dh3, dval, P["model.9.weight"], False, False, epi=EPI_MUL, E=self.pm3[:b])
            if dfeat is not None:
                tmp = self.h256[:b]
                ops.mul(tmp, dfeat, self.pm3[:b])
                ops.axpby(dh3, 1.0, dh3, 1.0, tmp)
        else:
            ops.mul(dh3, dfeat, self.pm3[:b])
        if param_grads:
            ops.gemm(G["model.7.weight"], dh3, self.a2[:b], True, False, accumulate=True)
            ops.colsum(G["model.7.bias"], dh3, 1.0, True)
        dh2 = self.g512[:b]
        ops.gemm(dh2, dh3, P["model.7.weight"], False, False, epi=EPI_MUL, E=self.pm2[:b])
        if param_grads:
            ops.gemm(G["model.4.weight"], dh2, self.a1[:b], True, False, accumulate=True)
            ops.colsum(G["model.4.bias"], dh2, 1.0, True)
        da1 = self.g1024a[:b]
        ops.gemm(da1, dh2, P["model.4.weight"], False, False)
        dh0 = self.g1024b[:b]
        ops.bn_bwd(dh0, da1, self.h0[:b], None, self.pm1[:b], self.stats[0], self.stats[1], P["model.1.weight"],
                   G["model.1.weight"] if param_grads else None, G["model.1.bias"] if param_grads else None, True, 0.2)
        if param_grads:
            gW1 = G["model.0.weight"]
            ops.gemm(gW1[:, :Gd], dh0, self.x, True, False, accumulate=True)
            ops.gemm(gW1[:, Gd:], dh0, self.emb[:b], True, False, accumulate=True)
            if self.labels is not None:
                ops.gemm(self.demb[:b], dh0, W1[:, Gd:], False, False)
                _embed_scatter(ops, G, self.demb[:b], self.labels, True)
        if dx_out is not None:
            ops.gemm(dx_out, dh0, W1[:, :Gd], False, False, accumulate=True)

    def gp_backward(self, lambda_gp, red, slot):
        """After forward(interpolates, labels): the first-order chain to the input
        gradient, the penalty (row terms reduced into `slot`), and the
        second-order chain accumulating d(lambda*penalty)/d(params)."""
        ops, P, G, Gd, b = self.ops, self.net.P, self.net.G, self.Gd, self.b
        W1, W2, W3, w4 = P["model.0.weight"], P["model.4.weight"], P["model.7.weight"], P["model.9.weight"]
        u3, u2, u1, u0 = self.g256[:b], self.g512[:b], self.g1024a[:b], self.g1024b[:b]
        ops.gemm(u3, self.ones[:b], w4, False, False, epi=EPI_MUL, E=self.pm3[:b])
        ops.gemm(u2, u3, W3, False, False, epi=EPI_MUL, E=self.pm2[:b])
        ops.gemm(u1, u2, W2, False, False, epi=EPI_MUL, E=self.pm1[:b])
        ops.bn_bwd(u0, u1, self.h0[:b], None, None, self.stats[0], self.stats[1], P["model.1.weight"], None, None,
                   False, 1.0, self.ubar, self.mbar)
        ux = self.ux[:b]
        ops.gemm(ux, u0, W1[:, :Gd], False, False)
        ops.gp_penalty(ux, self.rowpen[:b], ux, lambda_gp)          # ux now holds V
        ops.reduce_sum(red, slot, self.rowpen[:b])
        gW1 = G["model.0.weight"]
        ops.gemm(gW1[:, :Gd], u0, ux, True, False, accumulate=True)
        w = self.h1024a[:b]
        ops.gemm(w, ux, W1[:, :Gd], False, True)
        vu1, vh0 = self.h1024b[:b], self.h1024a[:b]
        ops.bn_bwd_bwd(vu1, vh0, G["model.1.weight"], True, w, u1, self.h0[:b], self.stats[0], self.stats[1],
                       P["model.1.weight"], self.ubar, self.mbar)
        t1 = vu1
        ops.mul(t1, vu1, self.pm1[:b])
        ops.gemm(G["model.4.weight"], u2, t1, True, False, accumulate=True)
        t2 = self.h512[:b]
        ops.gemm(t2, t1, W2, False, True, epi=EPI_MUL, E=self.pm2[:b])
        ops.gemm(G["model.7.weight"], u3, t2, True, False, accumulate=True)
        vu3 = self.h256[:b]
        ops.gemm(vu3, t2, W3, False, True, epi=EPI_MUL, E=self.pm3[:b])
        ops.colsum(G["model.9.weight"], vu3, 1.0, True)
        ops.gemm(gW1[:, :Gd], vh0, self.x, True, False, accumulate=True)
        ops.gemm(gW1[:, Gd:], vh0, self.emb[:b], True, False, accumulate=True)
        if self.labels is not None:
            ops.gemm(self.demb[:b], vh0, W1[:, Gd:], False, False)
            _embed_scatter(ops, G, self.demb[:b], self.labels, True)


class CriticStack:
    """Several calls of one critic evaluated at once.  The reference calls a critic on the real, the generated and
    the interpolated batch one after the other (GAN_functions.py:326-336), and three times per species in the
    generator step (:347, :366-371).  Here the batches are stacked along the rows ("segments"): every linear layer
    runs once over all rows, BatchNorm normalises each segment with its own statistics and updates the running
    statistics once per segment in the reference's call order, dropout masks are per row anyway.

    nseg: segments of the hidden activations; nx: distinct input batches.  With nx < nseg the first input feeds
    segments 0..nseg-nx (the generator step looks at the same generated batch twice), the others one segment each.
    Segments are packed at stride b (the current batch size), so ragged batches stay contiguous.

    The critic step's stack (nseg = nx = 3: real | generated | interpolated) keeps one row block more in two
    workspaces, laid out so that the whole gradient of the first layer's weight is ONE product over three row blocks
    (w1_grad): the stacked input holds [x_real; x_generated; V; x_interp] and the hidden-gradient workspace
    [dh0_real; dh0_generated; u0; vh0]."""

    def __init__(self, engine, net: Net, in_dim, max_b, nseg, nx):
        ops = engine.ops
        self.ops, self.net = ops, net
        self.Gd, self.E, self.S, self.nx = in_dim, engine.emb_dim, nseg, nx
        Z, B, S = ops.zeros, max_b, nseg
        # the first layer's input [x | emb] as the reference concatenates it (GAN_functions.py:184), so that the
        # layer and its weight gradient are one product each; x and emb are column views
        self.W = in_dim + self.E
        Zs = getattr(ops, "shadowed", Z)        # see GenPass: written only through ops wrappers that keep the bf16 image
        # writers: critic_inputs / copy / gp_penalty (gene columns), embed_gather / copy (emb columns)
        self.gp_layout = nseg == 3 and nx == 3      # the critic step's stack: see the class docstring
        self.xe = Zs((nx + 1) * B, _pad4(self.W))
        self.x = self.xe[:, :in_dim]
        self.emb = self.xe[:, in_dim:self.W]
        self.lab = ops.to_device(torch.zeros((nx + 1) * B, getattr(engine, "n_cov", 1), dtype=torch.int64))
        self.h0, self.a1, self.pm1 = Z(S * B, 1024), Zs(S * B, 1024), Z(S * B, 1024)
        self.stats = (Z(S, 1024), Z(S, 1024))
        self.a2, self.pm2 = Zs(S * B, 512), Z(S * B, 512)
        self.a3, self.pm3 = Zs(S * B, 256), Z(S * B, 256)
        self.val = Z(S * B, 1)
        self.dval = Z(S * B, 1)
        self.ones = ops.to_device(torch.ones(B, 1))
        self.g256, self.g512, self.g1024a, self.g1024b = Z(S * B, 256), Zs(S * B, 512), Z(S * B, 1024), Zs((S + 1) * B, 1024)
        self.h256, self.h512, self.h1024a, self.h1024b = Z(B, 256), Zs(B, 512), Z(B, 1024), Z(B, 1024)
        self.dsum = Z(B, 1024)
        self.demb = Z((nx + 1) * B, self.E)
        self.zero_emb = Z(B, max(self.E, 1))
        self.demb_diff = Z(B, max(self.E, 1))
        self._lsum = {}
        self.alpha = None                       # the interpolation weights of the current critic step (forward)
        self._dval_const = {}
        self.rowpen = Z(B)
        self.ubar, self.mbar = Z(1024), Z(1024)
        self.b, self.have_labels = 0, True
        self._late_colsums = []

    def x_seg(self, i, b):
        return self.x[i * b:(i + 1) * b]

    def load(self, i, b, x, labels):
        """Input batch i (x may be None when the rows were written in place through x_seg)."""
        if x is not None:
            if hasattr(self.ops, "copy"):
                self.ops.copy(self.x[i * b:(i + 1) * b], x)
            else:
                self.x[i * b:(i + 1) * b].copy_(x)
        self.have_labels = labels is not None
        if labels is not None:
            self.lab[i * b:(i + 1) * b].copy_(labels)

    def forward(self, b, m1, m2, order, interp=None):
        """m1 (nseg*b, 1024), m2 (nseg*b, 512): dropout keep masks.  Returns validity (nseg*b, 1) and the features
        (nseg*b, 256), segment-major.
        interp (three input batches only): the per-row alpha with which the third input batch was interpolated from
        the first two (GAN_functions.py:196-199, with the second batch's labels).  The first layer is linear, so its
        output on the third batch is then combined from its outputs on the first two instead of being multiplied
        out again - a third of the layer's product, the largest of the critic step."""
        ops, P, Bf, Gd, S, nx = self.ops, self.net.P, self.net.B, self.Gd, self.S, self.nx
        self.b = b
        R, Rx, off = S * b, nx * b, (S - nx) * b
        W1 = P["model.0.weight"]
        if interp is not None:
            self.alpha = interp
        # row blocks of the stacked input that hold input batches: 0, 1 and - in the critic step's layout - 3
        # (with labels, the critic step's layout takes all four blocks in one launch: the rows of block 2 carry the
        # label -1, no cell, and get the zeros that V needs under the embedding columns)
        blocks = [(0, 2 * b), (3 * b, 4 * b)] if self.gp_layout else [(0, Rx)]
        if self.gp_layout and self.have_labels:
            blocks = [(0, 4 * b)]
        for lo, hi in blocks:
            if self.have_labels:
                _embed_gather(ops, self.emb[lo:hi], P, self.lab[lo:hi])
            else:
                _zero(ops, self.emb[lo:hi])
        h0 = self.h0[:R]
        if interp is not None:
            assert self.gp_layout
            ops.gemm(h0[:2 * b], self.xe[:2 * b, :self.W], W1, False, True, bias=P["model.0.bias"])
            ops.interp_hidden(h0, b, interp, self.emb[:2 * b] if self.have_labels else None,
                              W1[:, Gd:self.W] if self.have_labels else None)
        elif self.gp_layout:        # the interpolates' first layer multiplied out, as the reference does
            ops.gemm(h0[:2 * b], self.xe[:2 * b, :self.W], W1, False, True, bias=P["model.0.bias"])
            ops.gemm(h0[2 * b:], self.xe[3 * b:4 * b, :self.W], W1, False, True, bias=P["model.0.bias"])
        else:
            ops.gemm(h0[off:], self.xe[:Rx, :self.W], W1, False, True, bias=P["model.0.bias"])
        for k in range(S - nx):
            h0[k * b:(k + 1) * b].copy_(h0[off:off + b])
        ops.bn_fwd_seg(self.a1[:R], h0, S, order, P["model.1.weight"], P["model.1.bias"], self.stats[0], self.stats[1],
                       Bf["model.1.running_mean"], Bf["model.1.running_var"], 0.2, m1, self.pm1[:R])
        self.net.nbt["model.1.num_batches_tracked"] += S
        ops.gemm(self.a2[:R], self.a1[:R], P["model.4.weight"], False, True, bias=P["model.4.bias"], epi=EPI_LEAKY,
                 E=m2, aux=self.pm2[:R], slope=0.2)
        ops.gemm(self.a3[:R], self.a2[:R], P["model.7.weight"], False, True, bias=P["model.7.bias"], epi=EPI_LEAKY,
                 E=None, aux=self.pm3[:R], slope=0.2)
        ops.gemm(self.val[:R], self.a3[:R], P["model.9.weight"], False, True, bias=P["model.9.bias"])
        return self.val[:R], self.a3[:R]

    _stages_done = False
    merged_stages = not os.environ.get("CGRNA_NO_MERGED_STAGES")

    def first_order_stages(self, b, scales):
        """The first two backward stages (head, third layer) of the critic step's two first-order chains - the penalty's
        on the interpolates (seed 1) and the Wasserstein terms' on the real and the generated rows (seeds `scales`) -
        as one chain over all three row blocks: the same weights, adjacent rows of the same workspaces, two launches
        instead of four.  gp_backward() and backward_params() of the same step then start from g256 / g512."""
        ops, P = self.ops, self.net.P
        assert self.gp_layout and len(scales) == 2
        key = (b, tuple(scales), "all")
        if key not in self._dval_const:
            seed = torch.tensor(list(scales) + [1.0], dtype=torch.float32).repeat_interleave(b)[:, None]
            self._dval_const[key] = ops.to_device(seed)
        R = 3 * b
        ops.gemm(self.g256[:R], self._dval_const[key], P["model.9.weight"], False, False, epi=EPI_MUL, E=self.pm3[:R])
        ops.gemm(self.g512[:R], self.g256[:R], P["model.7.weight"], False, False, epi=EPI_MUL, E=self.pm2[:R])
        self._stages_done = True

    def backward_params(self, b, scales):
        """Critic step: parameter gradients of  sum_s scales[s] * sum(validity of segment s)  over the first
        len(scales) segments (needs nx == nseg), accumulated into the gradient arena - except the first layer's
        weight, which w1_grad() computes together with the penalty's terms from the dh0 this leaves behind."""
        ops, P, G, Gd = self.ops, self.net.P, self.net.G, self.Gd
        assert self.nx == self.S
        ns = len(scales)
        R = ns * b
        # The seed gradients: a buffer of their own per (batch size, scales), written once when first needed (an eager
        # call: a step is captured at its second call) and never again, so a captured step reads constants and a
        # batch of another size in between cannot disturb them.
        key = (b, tuple(scales))
        if key not in self._dval_const:
            self._dval_const[key] = ops.to_device(torch.tensor(scales, dtype=torch.float32).repeat_interleave(b)[:, None])
        W1 = P["model.0.weight"]
        dval, dh3 = self._dval_const[key], self.g256[:R]
        ops.gemm(G["model.9.weight"], dval, self.a3[:R], True, False, accumulate=True)
        # the head's bias gradient is the sum of the seed gradients: exactly zero for the critic step's (-1/b, 1/b)
        # (the two halves cancel term by term), and adding zero is left out
        if sum(scales) != 0.0:
            ops.colsum(G["model.9.bias"], dval, 1.0, True)
        if not self._stages_done:
            ops.gemm(dh3, dval, P["model.9.weight"], False, False, epi=EPI_MUL, E=self.pm3[:R])
        ops.gemm(G["model.7.weight"], dh3, self.a2[:R], True, False, accumulate=True)
        dh2 = self.g512[:R]
        if not self._stages_done:
            ops.gemm(dh2, dh3, P["model.7.weight"], False, False, epi=EPI_MUL, E=self.pm2[:R])
        self._stages_done = False
        ops.gemm(G["model.4.weight"], dh2, self.a1[:R], True, False, accumulate=True)
        # the bias gradients of this chain and the column sum the penalty chain left for here, in one launch
        jobs = [(G["model.7.bias"], dh3, 1.0, True), (G["model.4.bias"], dh2, 1.0, True)] + self._late_colsums
        self._late_colsums = []
        if hasattr(ops, "colsum_multi"):
            ops.colsum_multi(jobs)
        else:
            for out, X, scale, acc in jobs:
                ops.colsum(out, X, scale, acc)
        da1 = self.g1024a[:R]
        ops.gemm(da1, dh2, P["model.4.weight"], False, False)
        dh0 = self.g1024b[:R]
        ops.bn_bwd_seg(dh0, da1, self.h0[:R], None, self.pm1[:R], ns, self.stats[0], self.stats[1], P["model.1.weight"],
                       G["model.1.weight"], G["model.1.bias"], True, 0.2)
        assert self.gp_layout and ns == 2       # the embedding tables' gradient: w1_grad()

    def w1_grad(self, b):
        """The critic step's whole gradient of the first layer's weight (a third of the step's products) as one
        product over three row blocks.  Its terms are dh0_r^T x_r + dh0_f^T x_f (Wasserstein chain), vh0^T x_i and
        u0^T V (penalty chain).  The interpolates are x_i = a x_r + (1 - a) x_f row by row (GAN_functions.py:196),
        so vh0^T x_i = (a vh0)^T x_r + ((1 - a) vh0)^T x_f: cg_gp_combine adds those multiples of vh0 to dh0 in place
        and  dW1 = [dh0_r + a vh0; dh0_f + (1 - a) vh0; u0]^T [x_r; x_f; V]  - the adjacent row blocks of the two
        workspaces.  The embedding columns of the interpolates are the generated batch's, not interpolated: there
        the product misses (a vh0)^T (emb_f - emb_r), added by a second, 12-column product.  Call after
        backward_params() and gp_backward() of the same step; written (first touch) to the gradient arena."""
        ops, G, P, Gd = self.ops, self.net.G, self.net.P, self.Gd
        assert self.gp_layout and self.alpha is not None
        av, de = self.h1024a[:b], self.demb_diff[:b]
        labels = self.have_labels and self.E > 0
        tables = _emb_tables(G) if labels else []
        # The embedding columns of the first layer, by label sums.  The tables' gradient is
        # sum over rows of label t of ([dh0_r; dh0_f; u0; vh0] W1[:, G:])[m] = (sum of those rows) W1[:, G:] - the rows of
        # u0 carry the label -1 (no cell) and drop out - and the correction product (a vh0)^T (emb_f - emb_r) is
        # (sum of a vh0 by generated label - sum by real label)^T table: two passes over hidden gradients and two
        # products with a dozen rows instead of two products with thousands of rows and a scatter.
        by_label = labels and hasattr(ops, "label_colsum") and not os.environ.get("CGRNA_NO_LABEL_SUMS") and \
            all(G[k].shape[0] <= ops.LABEL_COLSUM_MAX_TYPES for k, _ in tables)
        if by_label:
            off = Gd
            for i, (k, w) in enumerate(tables):
                S = self._label_sums(k)
                ops.label_colsum(S, self.g1024b[:4 * b], self.lab[:4 * b, i:i + 1])
                ops.gemm(G[k], S, P["model.0.weight"][:, off:off + w], False, False, accumulate=True)
                off += w
        elif labels:
            # the embedding tables' gradient of both chains in one product and one scatter over all four row blocks:
            # [dh0_r; dh0_f; u0; vh0] W1[:, G:] summed per cell type
            ops.gemm(self.demb[:4 * b], self.g1024b[:4 * b], P["model.0.weight"][:, Gd:], False, False)
            _embed_scatter(ops, G, self.demb[:4 * b], self.lab[:4 * b], True)
        plain = labels and not by_label
        ops.gp_combine(self.g1024b[:2 * b], self.g1024b[3 * b:4 * b], self.alpha, av,
                       self.emb[:2 * b] if plain else None, de if plain else None)
        ops.gemm(G["model.0.weight"], self.g1024b[:3 * b], self.xe[:3 * b, :self.W], True, False)
        if by_label:
            off = Gd
            for i, (k, w) in enumerate(tables):
                T = self._label_sums(k)
                ops.label_colsum(T, av, self.lab[b:2 * b, i:i + 1], minus=self.lab[:b, i:i + 1])
                ops.gemm(G["model.0.weight"][:, off:off + w], T, P[k], True, False, accumulate=True)
                off += w
        elif labels:
            ops.gemm(G["model.0.weight"][:, Gd:self.W], av, de, True, False, accumulate=True)

    def _label_sums(self, key):
        """(types of the table `key`) x 1024 workspace for sums of hidden gradients by label."""
        if key not in self._lsum:
            self._lsum[key] = self.ops.zeros(self.net.G[key].shape[0], 1024)
        return self._lsum[key]

    def backward_input(self, b, coef, idx, scale, dfeat, dx_out):
        """Generator step (nx = nseg - 1): gradient w.r.t. the first input batch of
        coef[idx]*scale*sum(validity of segment 0) + <dfeat, features of segment 1>, accumulated into dx_out.
        Both segments look at the same input, so their first-layer gradients are added before the one
        product with W1."""
        ops, P, Gd = self.ops, self.net.P, self.Gd
        assert self.S - self.nx == 1
        R = 2 * b
        dval, dh3 = self.dval[:b], self.g256[:R]
        ops.fill_coef(dval, coef, idx, scale)
        ops.gemm(dh3[:b], dval, P["model.9.weight"], False, False, epi=EPI_MUL, E=self.pm3[:b])
        ops.mul(dh3[b:], dfeat, self.pm3[b:R])
        dh2 = self.g512[:R]
        ops.gemm(dh2, dh3, P["model.7.weight"], False, False, epi=EPI_MUL, E=self.pm2[:R])
        da1 = self.g1024a[:R]
        ops.gemm(da1, dh2, P["model.4.weight"], False, False)
        dh0 = self.g1024b[:R]
        ops.bn_bwd_seg(dh0, da1, self.h0[:R], None, self.pm1[:R], 2, self.stats[0], self.stats[1], P["model.1.weight"],
                       None, None, False, 0.2)
        dsum = self.dsum[:b]
        ops.axpby(dsum, 1.0, dh0[:b], 1.0, dh0[b:])
        ops.gemm(dx_out, dsum, P["model.0.weight"][:, :Gd], False, False, accumulate=True)

    def gp_backward(self, b, seg, lambda_gp, red, slot):
        """Segment `seg` (the last one) holds the forward on the interpolates: first-order chain to the input
        gradient, the penalty (row terms reduced into `slot`), and the second-order chain accumulating
        d(lambda*penalty)/d(params) into the gradient arena (GAN_functions.py:194-204, :336-339) - except the first
        layer's weight: its two terms here, vh0^T x_interp and u0^T V (V = d(penalty)/d(input gradient), zeros under
        the embedding columns), are left to w1_grad(), for which u0, vh0 and V are written into the row blocks
        named in the class docstring."""
        ops, P, G, Gd = self.ops, self.net.P, self.net.G, self.Gd
        assert self.gp_layout and seg == 2
        sl = slice(2 * b, 3 * b)
        mean, invstd = self.stats[0][seg], self.stats[1][seg]
        h0, pm1, pm2, pm3 = self.h0[sl], self.pm1[sl], self.pm2[sl], self.pm3[sl]
        W1, W2, W3, w4 = P["model.0.weight"], P["model.4.weight"], P["model.7.weight"], P["model.9.weight"]
        u3, u2, u1 = self.g256[sl], self.g512[sl], self.g1024a[sl]
        u0, vh0 = self.g1024b[2 * b:3 * b], self.g1024b[3 * b:4 * b]
        ux = self.xe[2 * b:3 * b, :Gd]
        if self.E and not self.have_labels:      # V is zero under the embedding columns (with labels: forward() did it)
            if hasattr(ops, "copy"):
                ops.copy(self.xe[2 * b:3 * b, Gd:self.W], self.zero_emb[:b])
            else:
                self.xe[2 * b:3 * b, Gd:self.W].zero_()
        if not self._stages_done:
            ops.gemm(u3, self.ones[:b], w4, False, False, epi=EPI_MUL, E=pm3)
            ops.gemm(u2, u3, W3, False, False, epi=EPI_MUL, E=pm2)
        ops.gemm(u1, u2, W2, False, False, epi=EPI_MUL, E=pm1)
        ops.bn_bwd(u0, u1, h0, None, None, mean, invstd, P["model.1.weight"], None, None, False, 1.0, self.ubar,
                   self.mbar)
        ops.gemm(ux, u0, W1[:, :Gd], False, False)
        if hasattr(ops, "gp_penalty_parts") and b <= 1024:          # ux then holds V; the row terms are the partial sums
            ops.gp_penalty_parts(ux, red, slot, ux, lambda_gp)
        else:
            ops.gp_penalty(ux, self.rowpen[:b], ux, lambda_gp)
            ops.reduce_sum(red, slot, self.rowpen[:b])
        w = self.h1024a[:b]
        ops.gemm(w, ux, W1[:, :Gd], False, True)
        t1 = self.h1024b[:b]        # V_u1 times the first layer's activation/dropout factor, in one pass
        ops.bn_bwd_bwd(t1, vh0, G["model.1.weight"], True, w, u1, h0, mean, invstd, P["model.1.weight"], self.ubar,
                       self.mbar, PM=pm1)
        ops.gemm(G["model.4.weight"], u2, t1, True, False, accumulate=True)
        t2 = self.h512[:b]
        ops.gemm(t2, t1, W2, False, True, epi=EPI_MUL, E=pm2)
        ops.gemm(G["model.7.weight"], u3, t2, True, False, accumulate=True)
        vu3 = self.h256[:b]
        ops.gemm(vu3, t2, W3, False, True, epi=EPI_MUL, E=pm3)
        # d(penalty)/d(head weight) = column sums of vu3: a leaf, summed together with the bias gradients at the end of
        # backward_params(), which the critic step calls next (h256 is not written in between)
        self._late_colsums = [(G["model.9.weight"], vu3, 1.0, True)]


class PendingMetrics:
    """The metrics of a generator step (GAN_functions.py:411-419) on their way to the host: iteration(...,
    metrics="deferred") queues the copies of the loss records into pinned memory behind the step and returns this
    instead of waiting for them, so the host can go on issuing the next iterations; get() waits for the copies (not
    for anything issued later) and returns the dict that metrics=True would have returned."""

    _free = {}      # device -> pinned staging buffers not in use

    def __init__(self, eng, critic):
        self.eng, self.critic = eng, critic
        dev = eng.ops.device
        pool = PendingMetrics._free.setdefault(dev, [])
        self.buf = pool.pop() if pool else (torch.empty(eng.d_tot.numel(), dtype=eng.d_tot.dtype).pin_memory(),
                                            torch.empty(eng.gloss.numel(), dtype=eng.gloss.dtype).pin_memory(),
                                            torch.empty(eng.var_out.numel(), dtype=eng.var_out.dtype).pin_memory())
        for dst, src in zip(self.buf, (eng.d_tot, eng.gloss, eng.var_out)):
            assert dst.numel() == src.numel() and dst.dtype == src.dtype
            dst.copy_(src.view(-1), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(dev))
        self.value = None

    def get(self):
        if self.value is None:
            self.event.synchronize()
            d, g, v = self.buf
            out = {}
            if self.critic:
                out.update(self.eng.d_losses(d))
            out.update(self.eng.g_losses(g, v))
            self.value = out
            PendingMetrics._free[self.eng.ops.device].append(self.buf)
            self.buf = self.eng = None
        return self.value


# graphs and events of engines that were dropped while a capture was running in the same thread: EngineBase.close()
# releases them once no capture is running
_DEFERRED_RELEASES = []

# This package never destroys a HIP stream, and therefore never destroys a HIP graph that has parallel branches.
# Two defects of the HIP runtime (the libamdhip64 that ships with torch 2.10 + ROCm 7.0) are behind the rule:
#  * Destroying a stream queues a last marker on it and waits for it on the host; the marker's completion handler,
#    run later by the runtime's signal thread, still decrements a counter and clears a flag in the stream's device
#    object (8 bytes at +0x98, 4 bytes at +0x378), which the destroying thread may have freed by then.  Whatever
#    the host allocated there since - a weight read back from the device, the next model's initial weights, a
#    recorded loss - has one word lowered by one and one word 736 bytes further set to zero; if the pages are gone
#    the process dies of a segmentation fault in that thread.  A graph with parallel branches owns up to four
#    internal streams, which hipGraphExecDestroy destroys.
#  * At every launch such a graph uses all but one of its internal streams, skipping those that share the launch
#    stream's hardware queue, without checking that enough are left.  A new stream joins the hardware queue that
#    carries the fewest streams, so the internal streams land on different queues as long as the process only ever
#    creates streams; after uneven destruction two of them can share the launch stream's queue, and the launch reads
#    past the end of its stream list (a segmentation fault inside hipGraphLaunch).
# Forked graphs of engines that are done are therefore parked here for the life of the process (with the fork/join
# events recorded in their captures); a process may make MAX_FORKED_GRAPHS of them, later steps are captured on one
# stream (a graph without internal streams, which is destroyed like any other object).
_PARKED_FORKED_GRAPHS = []
_forked_graphs_made = 0
MAX_FORKED_GRAPHS = int(os.environ.get("CGRNA_MAX_FORKED_GRAPHS", "256"))


class EngineBase:
    """Execution plumbing shared by the training engines: optimizer step on a flat arena, HIP-graph capture and
    replay of a step, independent branches on side streams.  A subclass provides self.ops, self.nets (name -> Net),
    self.max_b and self.sync."""

    use_graphs = False
    concurrent = False
    _capture_single_stream = False
    _pin = None
    _lane_sides = None      # side streams of the lane that is issuing work (None: the main lane's self.sides)

    # ---- two-stream concurrency ------------------------------------------------------
    def enable_concurrency(self):
        """Run the mouse->human and human->mouse halves of a step (and the two critics) on two HIP
        streams.  They are independent between a few join points, and most kernels of one half
        leave much of the chip idle at batch 512, so the halves overlap.  Only for noise that is
        drawn on the device: the order of draws then differs from the reference's call order."""
        self.sides = [self.ops.stream(f"side{i}") for i in range(3)]
        self.ops.prepare_stream(torch.cuda.current_stream(self.ops.device))
        self.concurrent = True

    def _check_stream_priority(self):
        """Branches on side streams joined to a caller's stream of raised priority stop making progress on this
        ROCm build (eagerly now and then, replayed from a graph always), so that combination is refused."""
        if self.concurrent and getattr(torch.cuda.current_stream(self.ops.device), "priority", 0) != 0:
            raise RuntimeError("the step's concurrent branches need a current stream of default priority "
                               "(call the engine from such a stream, or do not enable_concurrency())")

    def _many(self, *branches):
        """Run independent branches side by side: the first on the current stream, the others on side streams
        that start from what the current stream has queued so far; then join.  The host calls them in the order
        given, so noise is drawn in the same order as in a sequential run."""
        if not self.concurrent:
            for fn in branches:
                fn()
            return
        self._check_stream_priority()
        main = torch.cuda.current_stream(self.ops.device)
        used = (self.sides if self._lane_sides is None else self._lane_sides)[:len(branches) - 1]
        assert len(used) == len(branches) - 1, "this lane has too few side streams"
        for st in used:
            self._after(st, main)
        branches[0]()
        for st, fn in zip(used, branches[1:]):
            with torch.cuda.stream(st):
                fn()
        for st in used:
            self._after(main, st)

    _edge_events = None

    def _after(self, waiter, other):
        """Work queued on `waiter` from now on starts after what `other` has queued so far.  Like
        waiter.wait_stream(other), but with one long-lived event per (other, waiter) pair instead of an event made
        and destroyed per call while its record and its wait are still queued: a step forks and joins hundreds of
        times, and an event is only safe to reuse, never to outlive."""
        if self._edge_events is None:
            self._edge_events = {}
        key = (other.cuda_stream, waiter.cuda_stream)
        ev = self._edge_events.get(key)
        if ev is None:
            ev = self._edge_events[key] = torch.cuda.Event()
        ev.record(other)
        waiter.wait_event(ev)

    def _two(self, first, second):
        self._many(first, second)

    # ---- optimizer ---------------------------------------------------------------
    def set_lr(self, lr):
        """LRDecayScheduler writes one lr into all three optimizers (:244-246)."""
        self.lr_g = lr
        self.lr_d = lr

    def _begin_opt(self, arena, lr):
        """Host side of an optimizer step: advance the step count and upload {lr, step}
        (the Adam kernel reads them from device memory, so a captured step can be replayed).
        The upload is asynchronous and the host may run many steps ahead of the device, so each upload has
        its own pinned staging slot; a slot is reused only after the copy that read it has completed."""
        arena.step += 1
        if self.ops.device.type != "cuda":
            arena.lr_step.copy_(torch.tensor([lr, float(arena.step)], dtype=torch.float64))
            return
        if self._pin is None:
            self._pin = dict(buf=torch.zeros(PIN_SLOTS, 2, dtype=torch.float64, pin_memory=True),
                             done=[None] * PIN_SLOTS, next=0)
        ring = self._pin
        i = ring["next"]
        ring["next"] = (i + 1) % PIN_SLOTS
        if ring["done"][i] is not None:
            ring["done"][i].synchronize()
        else:
            ring["done"][i] = torch.cuda.Event()        # one event per slot, recorded again at every use of the slot
        host = ring["buf"][i]
        host[0], host[1] = lr, float(arena.step)
        arena.lr_step.copy_(host, non_blocking=True)
        ring["done"][i].record(torch.cuda.current_stream(self.ops.device))

    # betas (0.5, 0.9) with amsgrad: GAN_functions.py:223-225.  A subclass (or its config) may ask for torch.optim.Adam's
    # defaults instead, as GAN_optimize_optuna.ipynb cell-1 (train_with_curriculum) builds its optimizers.
    adam_betas = (0.5, 0.9)
    adam_amsgrad = True

    def _adam(self, arena, lo=0, hi=None):
        """The optimizer step on the element range [lo, hi) of the arena (default: all of it)."""
        b1, b2 = self.adam_betas
        sl = slice(lo, arena.n if hi is None else hi)
        kw = {} if arena.p16 is None else dict(p16=arena.p16[sl])
        self.ops.adam(arena.p[sl], arena.g[sl], arena.m[sl], arena.v[sl], arena.vmax[sl] if self.adam_amsgrad else None,
                      arena.lr_step, float(b1), float(b2), 1e-8, **kw)

    # ---- release -------------------------------------------------------------------
    def close(self):
        """Let go of this engine's HIP graphs and events, in a fixed order and with nothing in flight: wait for the
        device; park every graph that has parallel branches, with the fork/join events of its capture (such a graph is
        never destroyed, see _PARKED_FORKED_GRAPHS); destroy the others; then drop the eager fork/join events and
        the pinned upload ring.  Called by the training functions when they are done with an engine; an engine
        dropped without it is released the same way when it is collected (unless a graph capture is running in this
        thread: the release then waits for the next close() outside a capture instead of synchronising the device
        inside the capture)."""
        dev = getattr(getattr(self, "ops", None), "device", None)
        graphs, self._graphs = getattr(self, "_graphs", None) or {}, {}
        edge, self._edge_events = self._edge_events, None
        pin, self._pin = self._pin, None
        if dev is None or dev.type != "cuda" or not torch.cuda.is_initialized():
            return
        if torch.cuda.is_current_stream_capturing():
            _DEFERRED_RELEASES.append((graphs, edge, pin))        # released by the next close() outside a capture
            return
        torch.cuda.synchronize(dev)
        pending = _DEFERRED_RELEASES[:] + [(graphs, edge, pin)]
        del _DEFERRED_RELEASES[:]
        for graphs, _, _ in pending:
            for st in graphs.values():
                g, events = st.pop("graph", None), st.pop("events", None)
                if g is not None and st.get("forked"):
                    _PARKED_FORKED_GRAPHS.append((g, events))
                elif g is not None:
                    g.reset()               # hipGraphExecDestroy, hipGraphDestroy, the capture's memory pool
                del g, events
                st.clear()
        del pending[:], graphs, edge, pin   # the eager fork/join events and the upload ring
        torch.cuda.synchronize(dev)

    def __del__(self):
        if sys.is_finalizing():             # interpreter shutdown: torch and the runtime may be half gone
            return
        self.close()

    # ---- HIP graphs ------------------------------------------------------------------
    def enable_graphs(self):
        """Capture the device work of the critic step and of the generator step into HIP
        graphs (first full-size call runs eagerly, the second is captured, later ones are
        replayed).  Needs a noise source that draws into fixed device buffers
        (noise.DeviceNoise); ragged batches stay eager.  With several ranks the collectives stay outside the
        graphs (see _run_parts)."""
        self._graphs = {}
        self._cap_streams = {}
        self.use_graphs = True
        ok, why = (self.ops.forked_graphs_supported() if hasattr(self.ops, "forked_graphs_supported") else (True, ""))
        if self.concurrent and not ok:
            # the branches of a step still run side by side when it is launched eagerly; its captured form is one
            # chain of kernels on one stream
            warnings.warn(why + "; the step's HIP graphs are captured on one stream (slower)", RuntimeWarning)
            self._capture_single_stream = True

    def _graphed(self, kind, body, tensors, noise, static=None):
        """`static`: the fixed device buffers this graph reads its inputs from (default: buffers of its own).  Two
        graphs that feed the same consumer must share them, because the consumer's graph holds their addresses."""
        self._check_stream_priority()
        st = self._graphs.setdefault(kind, dict(calls=0, graph=None, static=None, nbt=None))
        if st["static"] is None:
            st["static"] = list(static) if static is not None else [torch.empty_like(t) for t in tensors]
        # the step's inputs into the fixed buffers the captured graph reads: one launch for all of them when the
        # backend has it (four small launches between two graphs are 1 % of a 5000-gene iteration)
        pairs = list(zip(st["static"], tensors))
        if hasattr(self.ops, "copy_many") and all(
                d.is_contiguous() and t.is_contiguous() and d.dtype == t.dtype and d.shape == t.shape
                and d.device == t.device == self.ops.device and d.element_size() % 4 == 0 for d, t in pairs):
            self.ops.copy_many(pairs)
        else:
            for dst, src in pairs:
                dst.copy_(src)
        st["calls"] += 1
        if st["calls"] == 1:
            body(*st["static"], noise)
            return
        if st["graph"] is None:
            before = {n: dict(net.nbt) for n, net in self.nets.items()}
            # keep_graph: the captured graph stays with the executable graph made from it (g.instantiate() below)
            g = torch.cuda.CUDAGraph(keep_graph=True)
            # one capture stream per lane, with its scratch allocated beforehand: kernels captured on it take their
            # scratch by stream, and graphs of different lanes are replayed side by side
            lane = "ahead" if self._lane_sides is not None else "main"
            if lane not in self._cap_streams:
                self._cap_streams[lane] = self.ops.stream("capture-" + lane)
            # thread-local capture mode: helper threads of the process (a collective backend's progress thread, for
            # one) may keep making HIP calls while this thread captures
            # no garbage collection while capturing: destroying a dead object that owns HIP resources (an old graph,
            # a stream, an event) inside a capture aborts the process
            gc_was_on = gc.isenabled()
            gc.disable()
            # the fork/join events of a capture are its own: an event recorded while capturing refers to that
            # capture's nodes - they are kept with the graph and never recorded again
            eager_events, self._edge_events = self._edge_events, {}
            global _forked_graphs_made
            if self.concurrent and not self._capture_single_stream and _forked_graphs_made >= MAX_FORKED_GRAPHS:
                warnings.warn("this process has made %d HIP graphs with parallel branches, which are never destroyed; "
                              "further steps are captured on one stream (slower)" % _forked_graphs_made, RuntimeWarning)
                self._capture_single_stream = True
            forked = self.concurrent and not self._capture_single_stream
            was_concurrent = self.concurrent
            if self._capture_single_stream:
                self.concurrent = False             # every branch on the capture stream: a graph without forks
            try:
                with torch.cuda.graph(g, stream=self._cap_streams[lane], capture_error_mode="thread_local"):
                    body(*st["static"], noise)
                g.instantiate()
            finally:
                self.concurrent = was_concurrent
                st["events"], self._edge_events = self._edge_events, eager_events
                if gc_was_on:
                    gc.enable()
            st["forked"] = forked
            _forked_graphs_made += int(forked)
            st["graph"] = g
            st["nbt"] = {n: {k: net.nbt[k] - before[n][k] for k in net.nbt} for n, net in self.nets.items()}
            g.replay()
            return
        st["graph"].replay()
        for n, d in st["nbt"].items():
            for k, v in d.items():
                if v:
                    self.nets[n].nbt[k] += v



# --------------------------------------------------------------------------- #
class StepEngine(EngineBase):
    """State of a training run on one device + the two halves of an iteration."""

    def __init__(self, nets: Dict[str, Dict[str, torch.Tensor]], cfg: dict, mouse_centroids, human_centroids, ops,
                 max_batch: int):
        self.ops, self.cfg = ops, dict(cfg)
        self.max_b = max_batch
        gm = nets["g_m2h"]["encoder.0.weight"].shape[1]
        gh = nets["g_h2m"]["encoder.0.weight"].shape[1]
        self.g_mouse, self.g_human = gm, gh
        self.emb_dim = sum(w for _, w in _emb_tables(nets["g_m2h"]))
        self.n_cov = len(_emb_tables(nets["g_m2h"]))
        split = {n: split_state(sd) for n, sd in nets.items()}
        order = {n: _arena_order(list(split[n][0].keys())) for n in nets}
        self.arena_g = Arena(ops, [((n, k), tuple(split[n][0][k].shape)) for n in ("g_m2h", "g_h2m") for k in order[n]])
        self.arena_d = Arena(ops, [((n, k), tuple(split[n][0][k].shape)) for n in ("d_m", "d_h") for k in order[n]])
        self.nets = {}
        for n in ("g_m2h", "g_h2m", "d_m", "d_h"):
            arena = self.arena_g if n.startswith("g_") else self.arena_d
            net = Net(n, arena, order[n], split[n][1], ops)
            for k in order[n]:
                net.P[k].copy_(ops.to_device(split[n][0][k].detach().float()))
            self.nets[n] = net
        self.arena_g.sync_shadow()
        self.arena_d.sync_shadow()
        # opt-in departures from GAN_functions.py for the curriculum training of GAN_optimize_optuna.ipynb cell-1
        # (absent keys = the reference's step): plain Adam with the config's betas, one lr for all three
        # optimizers, Poisson cycle loss, loss weights handed in per epoch
        if "adam_betas" in cfg:
            self.adam_betas = tuple(cfg["adam_betas"])
        self.adam_amsgrad = bool(cfg.get("adam_amsgrad", True))
        self.cycle_poisson = cfg.get("cycle_loss", "l1") == "poisson"
        assert cfg.get("cycle_loss", "l1") in ("l1", "poisson")
        self.ext_weights = None              # set_loss_weights()
        self.lr_g = cfg["lr"] * float(cfg.get("g_lr_mult", 1.5))          # GAN_functions.py:223
        self.lr_d = cfg["lr"]                # :224-225
        self.mouse_centroids = ops.to_device(mouse_centroids.float()).contiguous()
        self.human_centroids = ops.to_device(human_centroids.float()).contiguous()
        B = max_batch
        N = self.nets
        self.gen_a = {"g_m2h": GenPass(self, N["g_m2h"], gm, gh, B), "g_h2m": GenPass(self, N["g_h2m"], gh, gm, B)}
        self.gen_b = {"g_m2h": GenPass(self, N["g_m2h"], gm, gh, B), "g_h2m": GenPass(self, N["g_h2m"], gh, gm, B)}
        # critic step: real | generated | interpolated; generator step: generated (adv, fm) | real
        self.dstack = {"d_m": CriticStack(self, N["d_m"], gm, B, 3, 3), "d_h": CriticStack(self, N["d_h"], gh, B, 3, 3)}
        self.gstack = {"d_m": CriticStack(self, N["d_m"], gm, B, 3, 2), "d_h": CriticStack(self, N["d_h"], gh, B, 3, 2)}
        Z = ops.zeros
        self.red = (Z(N_SLOTS, MAX_PARTS), Z(N_SLOTS, dtype=torch.int32))
        self.d_fake = {"human": Z(B, gh), "mouse": Z(B, gm)}
        self.d_recov = {"mouse": Z(B, gm), "human": Z(B, gh)}
        self.d_cyc = {"human": Z(B, gh), "mouse": Z(B, gm)}       # d(cycle loss)/d(generated batch)
        self.fm_mean = {k: Z(256) for k in ("h_real", "h_fake", "m_real", "m_fake")}
        self.d_feat = Z(B, 256)
        self.one = ops.to_device(torch.ones(1))
        self.ema = Z(5, dtype=torch.float64)                 # :283-287
        self.gloss = Z(12, dtype=torch.float64)
        self.coef = Z(6)
        self.d_tot = Z(6, dtype=torch.float64)
        self.d_slots = ops.to_device(torch.arange(6, dtype=torch.int32))
        self.d_scales = Z(6, dtype=torch.float64)
        self.var_out = Z(2)
        self.var_ws = [Z(max((max(gm, gh) + 31) // 32, 1), dtype=torch.float64) for _ in range(2)]
        self._d_scales_b = None
        self.d_feat2 = Z(B, 256)
        from .ddp import GradSync
        self.sync = GradSync(ops.device, stream=(lambda: ops.stream("comm", scratch=False)) if hasattr(ops, "stream") else None)
        self.g_tot = Z(N_GSLOTS, dtype=torch.float64)
        self.g_slots = ops.to_device(torch.arange(S_ADV_H, S_ADV_H + N_GSLOTS, dtype=torch.int32))
        self.g_ones = ops.to_device(torch.ones(N_GSLOTS, dtype=torch.float64))
        # identity loss: L1(g_m2h(human), human) + L1(g_h2m(mouse), mouse), the identity-mapping term of CycleGAN.
        # GAN_functions.py has no such term; it is an extension that config key 'lambda_identity' (> 0) switches on,
        # with a fixed weight outside the dynamic re-scaling.  Needs the two species to share one gene space.
        self.lambda_identity = float(cfg.get("lambda_identity", 0.0))
        self.gen_i = None
        if self.lambda_identity > 0:
            assert gm == gh, "the identity loss needs the same genes on both sides"
            self.gen_i = {"g_m2h": GenPass(self, N["g_m2h"], gm, gh, B), "g_h2m": GenPass(self, N["g_h2m"], gh, gm, B)}
            self.d_idt = {"human": Z(B, gh), "mouse": Z(B, gm)}

    @staticmethod
    def _critic_noise(noise, b, nseg, draw_order=None, alpha_before=None, first=None):
        """Dropout keep masks of `nseg` stacked critic calls (+ the penalty's alpha).  A source that draws in HBM
        makes one draw per layer for all segments; a replaying source is asked call by call, in the reference's
        order `draw_order` of segments (two masks per call; alpha just before the call on the interpolates,
        GAN_functions.py:196-199).  `first`: masks of segment 0 drawn earlier (the generator step's adversarial
        call, :347)."""
        alpha = None
        if getattr(noise, "stacked_draws", False):
            if alpha_before is not None:
                alpha = noise.rand((b, 1))
            return noise.keep((nseg * b, 1024), 0.3), noise.keep((nseg * b, 512), 0.3), alpha
        pairs = {} if first is None else {0: first}
        for sg in (range(nseg) if draw_order is None else draw_order):
            if sg == alpha_before:
                alpha = noise.rand((b, 1))
            pairs[sg] = (noise.keep((b, 1024), 0.3), noise.keep((b, 512), 0.3))
        return (torch.cat([pairs[i][0] for i in range(nseg)]), torch.cat([pairs[i][1] for i in range(nseg)]), alpha)

    # ---- critic step (:322-340) -------------------------------------------------------
    def d_step(self, mouse, human, mouse_cat, human_cat, noise):
        b = mouse.shape[0]
        assert human.shape[0] == b and b <= self.max_b
        if self._d_scales_b != b:
            lam = float(self.cfg["lambda_gp"])
            self.d_scales.copy_(torch.tensor([-1.0 / b, 1.0 / b, lam / b] * 2, dtype=torch.float64))
            self._d_scales_b = b
        ins = (mouse, human, mouse_cat, human_cat)
        if not self.sync.enabled:
            self._begin_opt(self.arena_d, self.lr_d)
            self._run_parts("d", b, ins, noise, [("graph", self._d_gen), ("graph", self._d_critics),
                                                 ("graph", self._d_update)])
        else:
            # several ranks: the gradient all-reduce of this step is left in flight and the critics' Adam update is
            # applied just before the critics are used next - after the generators' forward of the next step, which
            # does not need them - so the transfer hides behind that forward
            pend = self._d_pending_lr is not None
            self._run_parts("d", b, ins, noise, [("graph", self._d_gen),
                                                 ("eager", self._d_wait_pending if pend else None),
                                                 ("graph", self._d_critics),
                                                 ("eager", self._d_launch_comm)])

    def _run_parts(self, kind, b, tensors, noise, parts):
        """A step is a list of parts, ("graph", fn) for device work and ("eager", fn) for collectives and host
        work; a part whose fn is None is skipped.  The first part receives (tensors..., noise).  On one rank the
        whole list is captured as one HIP graph; with several ranks every device part is a graph of its own and
        the eager parts run between the replays."""
        graphs = self.use_graphs and b == self.max_b
        parts = [(k, fn) for k, fn in parts if fn is not None]

        def whole(*a):
            parts[0][1](*a)
            for _, fn in parts[1:]:
                fn()
        if not self.sync.enabled:
            if graphs:
                self._graphed(kind, whole, tensors, noise)
            else:
                whole(*tensors, noise)
            return
        for i, (k, fn) in enumerate(parts):
            if i == 0:
                if graphs:
                    self._graphed(f"{kind}:{fn.__name__}", fn, tensors, noise)
                else:
                    fn(*tensors, noise)
            elif k == "eager" or not graphs:
                fn()
            else:
                self._graphed(f"{kind}:{fn.__name__}", lambda _noise, fn=fn: fn(), (), noise)

    # Deferred critic update (several ranks only).
    _d_pending_lr = None

    def _d_wait_pending(self):
        self.sync.wait()

    def _d_launch_comm(self):
        # the critics' optimizer step rides on the communication stream behind the buckets: it needs nothing but the
        # averaged gradients, and runs beside the generators' forward passes that the calling stream issues next
        def update():
            self._begin_opt(self.arena_d, self.lr_d)
            self._adam(self.arena_d)
        self.sync.launch(self.arena_d.g, then=update)
        self._d_pending_lr = self.lr_d

    def flush(self):
        """Wait for a critic update that is still in flight (several ranks); call before reading critic parameters."""
        if self._d_pending_lr is not None:
            self._d_wait_pending()
            self._d_pending_lr = None

    def _d_gen(self, mouse, human, mouse_cat, human_cat, noise):
        out = {}

        def gen(which, x, lab):
            out[which] = self.gen_a[which].forward(x, lab, noise)
        self._two(lambda: gen("g_m2h", mouse, mouse_cat), lambda: gen("g_h2m", human, human_cat))
        self._d_state = (mouse, human, mouse_cat, human_cat, noise, out["g_m2h"], out["g_h2m"])

    # the critic's first layer on the interpolates from its outputs on the real and the generated batch (see
    # CriticStack.forward); False: multiplied out, as the reference does
    interp_by_linearity = not os.environ.get("CGRNA_INTERP_MULTIPLIED_OUT")

    def _d_critics(self):
        ops, cfg = self.ops, self.cfg
        mouse, human, mouse_cat, human_cat, noise, fake_human, fake_mouse = self._d_state
        b = mouse.shape[0]

        drawn, self._d_noise = self._d_noise, None
        if not drawn and getattr(noise, "stacked_draws", False):
            drawn = self._draw_d_noise(noise, b)

        def critic_fwd(name, real, real_lab, fake, fake_lab, s0):
            ds = self.dstack[name]
            m1, m2, alpha = drawn[name] if drawn else self._critic_noise(noise, b, 3, alpha_before=2)
            # real | generated | interpolated batch and their labels into the stacked input, one launch
            ds.have_labels = real_lab is not None
            ops.critic_inputs(ds.x[:4 * b], ds.lab[:4 * b] if ds.have_labels else None, real, fake, alpha,
                              real_lab.contiguous() if ds.have_labels else None,
                              fake_lab.contiguous() if ds.have_labels else None, interp_block=3)
            ds.alpha = alpha
            val, _ = ds.forward(b, m1, m2, (0, 1, 2), interp=alpha if self.interp_by_linearity else None)
            if hasattr(ops, "reduce_sum_seg"):
                ops.reduce_sum_seg(self.red, s0, val[:2 * b], 2)
            else:
                ops.reduce_sum(self.red, s0, val[:b])
                ops.reduce_sum(self.red, s0 + 1, val[b:2 * b])
        # One branch per critic, forward and backward (nothing of one critic waits for the other): its gradients cleared, the penalty's double-backward chain on the
        # interpolates, the Wasserstein chain on the real and the generated rows, then the first layer's weight
        # gradient from what both left behind (CriticStack.w1_grad).  Every tensor is accumulated by one stream, so
        # there is no second gradient buffer and nothing to add afterwards.  (Four branches - the two chains of a
        # critic side by side - measured the same with the chains' own first-layer products, 2.887 against 2.904 ms;
        # with one product for both chains they must meet anyway.)
        def backward(name, slot):
            ds = self.dstack[name]
            self.arena_d.net_grad(name).zero_()
            if ds.merged_stages:
                ds.first_order_stages(b, (-1.0 / b, 1.0 / b))
            ds.gp_backward(b, 2, cfg["lambda_gp"], self.red, slot)
            ds.backward_params(b, (-1.0 / b, 1.0 / b))
            ds.w1_grad(b)
            if not self.sync.enabled:       # one rank: the critic's optimizer step at the end of its own branch,
                self._adam(self.arena_d, *self.arena_d.net_range[name])     # beside the other critic's products

        def critic(name, real, real_lab, fake, fake_lab, s0):
            critic_fwd(name, real, real_lab, fake, fake_lab, s0)
            backward(name, s0 + 2)
        self._many(lambda: critic("d_m", mouse, mouse_cat, fake_mouse, human_cat, S_DM_REAL),
                   lambda: critic("d_h", human, human_cat, fake_human, mouse_cat, S_DH_REAL))
        ops.sum_parts(self.d_tot, self.red, self.d_slots, self.d_scales)

    _d_noise = None      # masks and alpha of the critic step drawn before its critics' part is issued

    def _draw_d_noise(self, noise, b):
        """The critic step's draws in the order of _d_critics (mouse critic, then human critic).  A source that draws
        in HBM makes them for both critics at once - one draw for all dropout masks, one for the two alphas: three
        launches on the stream the critics' branches fork from instead of ten."""
        if getattr(noise, "stacked_draws", False):
            n1, n2 = 3 * b * 1024, 3 * b * 512
            keep = noise.keep((2, n1 + n2), 0.3)
            alpha = noise.rand((2, b, 1))
            return {name: (keep[k, :n1].view(3 * b, 1024), keep[k, n1:].view(3 * b, 512), alpha[k])
                    for k, name in enumerate(("d_m", "d_h"))}
        return {name: self._critic_noise(noise, b, 3, alpha_before=2) for name in ("d_m", "d_h")}

    def _d_update(self):
        if self.sync.enabled:       # one rank: done in the critics' backward branches (_d_critics)
            self._adam(self.arena_d)

    def d_losses(self, t=None):
        """t: a host copy of the critic step's six loss sums (default: read now, which waits for the device)."""
        t = self.d_tot.cpu() if t is None else t
        f32 = lambda v: float(torch.tensor(v, dtype=torch.float32))  # noqa: E731
        return dict(d_m_loss=f32(float(t[0]) + float(t[1]) + float(t[2])),
                    d_h_loss=f32(float(t[3]) + float(t[4]) + float(t[5])),
                    gp_m=float(t[2]) / float(self.cfg["lambda_gp"]), gp_h=float(t[5]) / float(self.cfg["lambda_gp"]))

    # ---- generator step (:343-419) -------------------------------------------------------
    # the passes whose encoder the generator step's first forward takes over (GenPass.forward, encoder_of): set by
    # iteration() around a generator step that follows the critic step of the same batch
    _g_encoder_of = None
    reuse_encoder = not os.environ.get("CGRNA_NO_ENCODER_REUSE")

    def g_step(self, mouse, human, mouse_cat, human_cat, noise):
        b = mouse.shape[0]
        self._begin_opt(self.arena_g, self.lr_g)
        multi = self.sync.enabled and self._d_pending_lr is not None
        # a captured generator step with the encoder taken over is a different graph from one that computes it
        self._run_parts("g" if self._g_encoder_of is None else "ge", b, (mouse, human, mouse_cat, human_cat), noise,
                        [("graph", self._g_first),
                         ("eager", self._d_wait_pending_clear if multi else None),   # the critics are needed from here on
                         ("graph", self._g_rest),
                         ("eager", self._g_loss_comm if self.sync.enabled else None),
                         ("graph", self._g_backward),
                         # several ranks: the decoders' gradients (complete first) are averaged while the rest of the
                         # backward pass runs
                         ("eager", self._g_grad_comm_decoders if self.sync.enabled else None),
                         ("graph", self._g_backward_rest if self.sync.enabled else None),
                         ("eager", self._g_grad_comm_rest if self.sync.enabled else None),
                         # (several ranks: nothing left for the device here, the optimizer step went out behind the
                         # all-reduces)
                         ("eager" if self.sync.enabled else "graph", self._g_update)])

    def _d_wait_pending_clear(self):
        self._d_wait_pending()
        self._d_pending_lr = None

    def _g_first(self, mouse, human, mouse_cat, human_cat, noise):
        A, Bp = self.gen_a["g_m2h"], self.gen_a["g_h2m"]
        o = {}
        # Forward, in the reference's call order (:344-372).  Each pair below is independent and may run on two
        # streams; the human-side member comes first, as in the reference, so a sequential run draws its noise in
        # the reference's order.

        src = self._g_encoder_of or {}

        def fwd(key, p, x, lab, name):
            o[key] = p.forward(x, lab, noise, encoder_of=src.get(name))
        self._two(lambda: fwd("fh", A, mouse, mouse_cat, "g_m2h"), lambda: fwd("fm", Bp, human, human_cat, "g_h2m"))
        self._g_state = (mouse, human, mouse_cat, human_cat, o, noise)

    def _g_rest(self):
        ops = self.ops
        mouse, human, mouse_cat, human_cat, o, noise = self._g_state
        b = mouse.shape[0]
        A, Bp = self.gen_a["g_m2h"], self.gen_a["g_h2m"]
        Rm, Rh = self.gen_b["g_h2m"], self.gen_b["g_m2h"]
        red = self.red
        fake_human, fake_mouse = o["fh"], o["fm"]

        adv_masks = {}
        if not getattr(noise, "stacked_draws", False):      # the adversarial calls' masks, at the reference's place
            for name in ("d_h", "d_m"):                                     # in the draw order (:347)
                adv_masks[name] = (noise.keep((b, 1024), 0.3), noise.keep((b, 512), 0.3))

        def cyc(key, p, x, lab, target, slot):
            o[key] = p.forward(x, lab, noise)                               # :349-357
            if self.cycle_poisson:                                          # poisson_loss, :216-217
                ops.reduce_poisson(red, slot, o[key], target)
            else:
                ops.reduce_l1(red, slot, o[key], target)
            if self.gen_i:      # identity pass of the same network, in the same branch (its running statistics see
                if key == "rm":                                     # one order): g_h2m(mouse) / g_m2h(human)
                    o["im"] = self.gen_i["g_h2m"].forward(mouse, mouse_cat, noise)
                    ops.reduce_l1(red, S_IDT_M, o["im"], mouse)
                else:
                    o["ih"] = self.gen_i["g_m2h"].forward(human, human_cat, noise)
                    ops.reduce_l1(red, S_IDT_H, o["ih"], human)

        def critics(name, fake, real, lab, cents, src_lab, s_adv, s_ct, m_real, m_fake, s_fm, gen_first, s_kl, k_var):
            # segments: adversarial call on the generated batch (:347) | feature call on it (:368,:371) | feature
            # call on the real batch (:367,:370); the reference makes them in the order 0, 2, 1
            if k_var == 0:      # the generators' gradients are cleared here, beside the forward passes: nothing adds to
                self.arena_g.g.zero_()                                  # them before the backward pass
            gs = self.gstack[name]
            gs.load(0, b, fake, lab)
            gs.load(1, b, real, lab)
            m1, m2, _ = self._critic_noise(noise, b, 3, draw_order=(2, 1), first=adv_masks.get(name))
            val, feat = gs.forward(b, m1, m2, (0, 2, 1))
            ops.reduce_sum(red, s_adv, val[:b])
            ops.reduce_centroid_sq(red, s_ct, fake, cents, src_lab)         # :363-364
            ops.colsum(self.fm_mean[m_real], feat[2 * b:], 1.0 / b)         # :366-372
            ops.colsum(self.fm_mean[m_fake], feat[b:2 * b], 1.0 / b)
            ops.reduce_sqdiff(red, s_fm, self.fm_mean[m_real], self.fm_mean[m_fake])
            ops.reduce_kl(red, s_kl, gen_first.ml[:b])                      # :375-376
            # the recorded variance of the generated batch (:418-419) needs nothing but that batch: computed in this
            # branch instead of after the optimizer step, where it would run alone
            ops.colvar_mean(self.var_out[k_var:k_var + 1], self.var_ws[k_var], fake)
        # the cycle passes and the critic calls only need the generated batches: four independent branches
        self._many(lambda: cyc("rm", Rm, fake_human, human_cat, mouse, S_CYC_M),
                   lambda: cyc("rh", Rh, fake_mouse, mouse_cat, human, S_CYC_H),
                   lambda: critics("d_h", fake_human, human, human_cat, self.human_centroids, mouse_cat, S_ADV_H, S_CT_H,
                                   "h_real", "h_fake", S_FM_H, A, S_KL_H, 0),
                   lambda: critics("d_m", fake_mouse, mouse, mouse_cat, self.mouse_centroids, human_cat, S_ADV_M, S_CT_M,
                                   "m_real", "m_fake", S_FM_M, Bp, S_KL_M, 1))
        if self.sync.enabled:
            ops.sum_parts(self.g_tot, red, self.g_slots, self.g_ones)

    def _g_loss_comm(self):
        # every rank must derive the same loss scales: the local loss sums are replaced by their sum over ranks
        self.sync.sum_scalars(self.g_tot)

    def _g_backward(self):
        ops, cfg, red = self.ops, self.cfg, self.red
        mouse, human, mouse_cat, human_cat, o, _ = self._g_state
        fake_human, fake_mouse, recov_mouse, recov_human = o["fh"], o["fm"], o["rm"], o["rh"]
        b = mouse.shape[0]
        gm, gh = self.g_mouse, self.g_human
        A, Bp = self.gen_a["g_m2h"], self.gen_a["g_h2m"]
        Rm, Rh = self.gen_b["g_h2m"], self.gen_b["g_m2h"]
        L = cfg["latent_dim"]
        if self.sync.enabled:
            red[0][S_ADV_H:S_ADV_H + N_GSLOTS, 0] = self.g_tot.to(red[0].dtype)
            red[1][S_ADV_H:S_ADV_H + N_GSLOTS] = 1
        bl = b
        b = b * self.sync.world          # normalisation of the global-batch losses
        desc = dict(
            terms=[[(S_ADV_H, -1.0 / b), (S_ADV_M, -1.0 / b)],
                   [(S_CYC_M, 1.0 / (b * gm)), (S_CYC_H, 1.0 / (b * gh))],
                   [(S_CT_H, 1.0 / (b * gh)), (S_CT_M, 1.0 / (b * gm))],
                   [(S_FM_H, 0.5 / 256 / self.sync.world), (S_FM_M, 0.5 / 256 / self.sync.world)],
                   [(S_KL_H, -0.5 / (b * L)), (S_KL_M, -0.5 / (b * L))],
                   [(S_IDT_H, 1.0 / (b * gh)), (S_IDT_M, 1.0 / (b * gm))] if self.gen_i else []],
            ema_alpha=0.3)
        b = bl
        desc["lambda"] = [float(cfg["lambda_adv"]), float(cfg["lambda_cycle"]), float(cfg["lambda_celltype"]),
                          float(cfg["lambda_fm"]), float(cfg["lambda_kl"]), self.lambda_identity]
        desc["weights"] = self.ext_weights
        ops.gloss_finalize(self.ema, self.gloss, self.coef, red, desc)     # :380-406
        coef = self.coef
        # ---- backward.  Human side: everything that produces d(loss)/d(fake_human) (and g_h2m's gradients through
        # the cycle pass); mouse side likewise; then each first-pass generator.  The two sides touch disjoint buffers.
        dfh, dfm = self.d_fake["human"][:b], self.d_fake["mouse"][:b]

        def crit(dfake, fake, cents, src_lab, gdim, gs, m_fake, m_real, dfe):
            ops.centroid_bwd(dfake, fake, cents, src_lab, coef, 2, 2.0 / (b * gdim))
            ops.fm_bwd(dfe, self.fm_mean[m_fake], self.fm_mean[m_real], coef, 3, 1.0 / (256 * b))
            gs.backward_input(b, coef, 0, -1.0 / b, dfe, dfake)

        def cycb(drec, recov, target, tdim, R, dx):
            if self.cycle_poisson:
                ops.poisson_bwd(drec, recov, target, coef, 1, 1.0 / (b * tdim))
            else:
                ops.l1_bwd(drec, recov, target, coef, 1, 1.0 / (b * tdim))
            R.backward(drec, need_dx=True, dx_out=dx)
            if self.gen_i:      # the identity pass of the same network (its gradients go to the same tensors)
                side = "mouse" if R is Rm else "human"
                I = self.gen_i["g_h2m" if R is Rm else "g_m2h"]
                di = self.d_idt[side][:b]
                ops.l1_bwd(di, o["im" if R is Rm else "ih"], target, coef, 5, 1.0 / (b * tdim))
                I.backward(di)
        dch, dcm = self.d_cyc["human"][:b], self.d_cyc["mouse"][:b]
        self._many(lambda: crit(dfh, fake_human, self.human_centroids, mouse_cat, gh, self.gstack["d_h"], "h_fake",
                                "h_real", self.d_feat[:b]),
                   lambda: cycb(self.d_recov["mouse"][:b], recov_mouse, mouse, gm, Rm, dch),
                   lambda: crit(dfm, fake_mouse, self.mouse_centroids, human_cat, gm, self.gstack["d_m"], "m_fake",
                                "m_real", self.d_feat2[:b]),
                   lambda: cycb(self.d_recov["human"][:b], recov_human, human, gh, Rh, dcm))
        def first(p, dfake, dcyc, whole):      # each side adds its cycle term inside its own branch
            ops.axpby(dfake, 1.0, dfake, 1.0, dcyc)
            if whole:
                # one rank: a network's optimizer step is issued range by range inside its backward branch, as soon
                # as the range's gradients are complete (the cycle passes' contributions are in already) - bound by
                # HBM, it runs beside the other network's products instead of alone after both branches
                ranges = self._adam_stages()[p.net.name]
                p.backward(dfake, kl=(coef, 4, 1.0),
                           hook=lambda stage: [self._adam(self.arena_g, lo, hi) for lo, hi in ranges[stage]])
            else:
                p.backward_decoder(dfake)
        # several ranks: the first passes' backward in two halves with a collective launched between
        whole = not self.sync.enabled
        self._two(lambda: first(A, dfh, dch, whole), lambda: first(Bp, dfm, dcm, whole))

    def _g_backward_rest(self):
        A, Bp = self.gen_a["g_m2h"], self.gen_a["g_h2m"]
        self._two(lambda: A.backward_rest(kl=(self.coef, 4, 1.0)), lambda: Bp.backward_rest(kl=(self.coef, 4, 1.0)))

    def _g_decoder_ranges(self):
        """Element ranges of the generator gradient arena holding the decoder.* tensors of the two generators."""
        out = []
        for net in ("g_m2h", "g_h2m"):
            offs = [(o, o + _pad4(n)) for (nn_, k), (o, n, _) in self.arena_g.offsets.items()
                    if nn_ == net and k.startswith("decoder.")]
            out.append((min(o for o, _ in offs), max(h for _, h in offs)))
        return out

    def _g_grad_comm_decoders(self):
        # each generator's decoder gradients are complete once its first pass has gone back through the decoder
        # (the cycle passes, which add to the same tensors, ran before): 2 x 23 MB of the 151 MB at 5000 genes
        # travel beside the latent blocks' and the encoder's backward; the encoder gradients, the last to be
        # complete and 2 x 52 MB, are exposed
        # ... and their optimizer step follows them on the communication stream, beside the same backward work
        ranges = self._g_decoder_ranges()
        self.sync.launch(self.arena_g.g, ranges, then=lambda: [self._adam(self.arena_g, lo, hi) for lo, hi in ranges])

    def _g_grad_comm_rest(self):
        (a0, a1), (b0, b1) = self._g_decoder_ranges()
        # the rest range by range: a range's optimizer step is issued on the calling stream as soon as the range has
        # been averaged, and runs while the next range travels
        rest = [(lo, hi) for lo, hi in [(0, a0), (a1, b0), (b1, self.arena_g.n)] if hi > lo]
        done = [self.sync.launch(self.arena_g.g, [r]) for r in rest]
        cur = torch.cuda.current_stream(self.ops.device) if self.ops.device.type == "cuda" else None
        for (lo, hi), ev in zip(rest, done):
            if ev is not None:
                cur.wait_event(ev)
            self._adam(self.arena_g, lo, hi)
        self.sync.wait()        # the decoders' ranges and their optimizer step

    _adam_stage_ranges = None

    def _adam_stages(self):
        """Per generator: the element ranges of the gradient arena whose optimizer step may start at each stage of
        the first pass's backward (GenPass.backward, hook): "decoder" - decoder.* and whatever follows it in the
        arena; "first" - the encoder's first Linear and its BatchNorm, and whatever precedes them; "middle" - the
        rest."""
        if self._adam_stage_ranges is None:
            out = {}
            for net in ("g_m2h", "g_h2m"):
                lo_net, hi_net = self.arena_g.net_range[net]
                keys = [(k, o, o + _pad4(n)) for (nn_, k), (o, n, _) in self.arena_g.offsets.items() if nn_ == net]
                dec = [(o, h) for k, o, h in keys if k.startswith("decoder.")]
                fst = [(o, h) for k, o, h in keys if k.startswith(("encoder.0.", "encoder.1."))]
                d_lo, f_hi = min(o for o, _ in dec), max(h for _, h in fst)
                assert f_hi <= d_lo and all(k.startswith("decoder.") or ".gate." in k for k, o, _ in keys if o >= d_lo)
                assert all(k.startswith(("encoder.0.", "encoder.1.", "embeddings.")) for k, _, h in keys if h <= f_hi)
                out[net] = {"decoder": [(d_lo, hi_net)], "middle": [(f_hi, d_lo)], "first": [(lo_net, f_hi)]}
            self._adam_stage_ranges = out
        return self._adam_stage_ranges

    def _g_update(self):
        ops = self.ops
        o = self._g_state[4]
        fake_human, fake_mouse = o["fh"], o["fm"]
        # the optimizer step: one rank - range by range inside the backward branches (_g_backward); several ranks -
        # range by range behind each range's all-reduce (_g_grad_comm_decoders, _g_grad_comm_rest)
        self.fake_human, self.fake_mouse = fake_human, fake_mouse

    def set_loss_weights(self, weights):
        """Five loss weights (adv, cycle, celltype, fm, kl) used as they are from now on, in place of
        lambda_k x the moving-average scales of GAN_functions.py:380-406: the per-epoch curriculum weights
        x epoch-level scales of GAN_optimize_optuna.ipynb cell-1 (get_curriculum_weights, train_epoch).  They
        live in device memory, so a captured generator step picks up new values without a new capture.  Call
        before the first generator step (a step captured without them keeps the moving-average rule)."""
        w = torch.tensor([float(x) for x in weights], dtype=torch.float32)
        assert w.numel() == 5
        if self.ext_weights is None:
            assert not any(str(k).startswith("g") for k in getattr(self, "_graphs", {})), \
                "set_loss_weights() after a generator step was captured"
            self.ext_weights = self.ops.zeros(5)
        self.ext_weights.copy_(self.ops.to_device(w))

    def g_losses(self, t=None, v=None):
        """t, v: host copies of the generator step's loss record and variance metrics (default: read now)."""
        t = self.gloss.cpu() if t is None else t
        v = self.var_out.cpu() if v is None else v
        f32 = lambda x: float(torch.tensor(float(x), dtype=torch.float32))  # noqa: E731
        return dict(g_adv=float(t[0]), cycle_loss=float(t[1]), celltype_loss=float(t[2]), fm_loss=float(t[3]),
                    kl_loss=float(t[4]), g_loss=f32(t[5]), identity_loss=float(t[11]),
                    scales=dict(adv=float(t[6]), cycle=float(t[7]), celltype=float(t[8]), fm=float(t[9]), kl=float(t[10])),
                    fake_human_var=float(v[0]), fake_mouse_var=float(v[1]))

    def iteration(self, i, mouse, human, mouse_cat, human_cat, noise, nxt=None, metrics=True, critic=True):
        """One pass of the loop body :298-419; returns the metrics the reference
        records (only on generator steps, like :411-419).  `nxt`: the next iteration's
        (mouse, human, mouse_cat, human_cat), used only after enable_lookahead().  metrics=False skips the
        device-to-host read of the losses (the host then never waits for the device).  critic=False skips the
        critic step (GAN_optimize_optuna.ipynb cell-1, train_epoch: critics train only while lambda_adv > 0)."""
        if not critic:          # a curriculum phase without adversarial term trains the generators only
            assert not self.lookahead
            if i % self.cfg["nb_critic"] == 0:
                self.g_step(mouse, human, mouse_cat, human_cat, noise)
        elif self.lookahead and not self.sync.enabled:
            self._iteration_lookahead(i, mouse, human, mouse_cat, human_cat, noise, nxt)
        else:
            self.d_step(mouse, human, mouse_cat, human_cat, noise)
            if i % self.cfg["nb_critic"] == 0:
                # the critic step has just called both generators on this batch, in these workspaces
                self._g_encoder_of = self.gen_a if self.reuse_encoder else None
                try:
                    self.g_step(mouse, human, mouse_cat, human_cat, noise)
                finally:
                    self._g_encoder_of = None
        out = {}
        if metrics and i % self.cfg["nb_critic"] == 0:
            if metrics == "deferred" and self.ops.device.type == "cuda":
                return PendingMetrics(self, critic)
            if critic:
                out.update(self.d_losses())
            out.update(self.g_losses())
        return out

    # ---- generator forwards issued ahead of the critics' work (one rank) ---------------------------------
    lookahead = False

    def enable_lookahead(self):
        """The generators' forward passes of a critic step need nothing from the critics, and the generators only
        change in a generator step.  With lookahead they run on a second lane (a stream with side streams and
        scratch of its own) beside the critics' part of the previous critic step: the forward for iteration i+1
        while the critics work on iteration i, and the first forward of a generator step beside the critics' part
        of its own iteration.  Calls of a generator stay in the reference's order (the lane is one queue), so the
        running statistics see the same sequence; the values computed are those of the plain schedule given the
        same noise.  The critic step's forwards alternate between two workspaces, because one is still being read
        by the critics when the next is written."""
        assert self.concurrent, "enable_concurrency() first"
        self.ahead = self.ops.stream("ahead")
        self.ahead_sides = [self.ops.stream("ahead-side0")]
        N, B = self.nets, self.max_b
        gm, gh = self.g_mouse, self.g_human
        self.gen_d = [{"g_m2h": GenPass(self, N["g_m2h"], gm, gh, B), "g_h2m": GenPass(self, N["g_h2m"], gh, gm, B)}
                      for _ in range(2)]
        self._d_states = [None, None]
        # the last full-size state of each set: a replayed graph does not run the host code that records the state,
        # so after a ragged batch (run eagerly, on the same set) the full-size record is put back from here
        self._d_states_full = [None, None]
        # the batch of workspace set s has one fixed home: the set is filled either by "dgen{s}" or by
        # "pair{1-s}n", and the critics' graphs that read it later are captured once, with these addresses
        self._real = [None, None]
        self._dset = 0
        self._pre = None        # (workspace set, batch) of a critic-step forward already issued
        self.lookahead = True

    class _Lane:
        """Issue work on one stream of the lookahead lane (no further forks from it)."""

        def __init__(self, eng, noise, stream):
            self.eng, self.noise, self.stream = eng, noise, stream

        def __enter__(self):
            self.ctx = torch.cuda.stream(self.stream)
            self.ctx.__enter__()
            self.eng._lane_sides = ()
            if hasattr(self.noise, "space"):
                self.noise.space = "ahead"

        def __exit__(self, *exc):
            self.eng._lane_sides = None
            if hasattr(self.noise, "space"):
                self.noise.space = ""
            return self.ctx.__exit__(*exc)

    def _run_graph_or_eager(self, kind, body, tensors, noise, b, static=None):
        if self.use_graphs and b == self.max_b:
            self._graphed(kind, body, tensors, noise, static)
        else:
            body(*tensors, noise)

    def _real_set(self, s, like):
        if like[0].shape[0] != self.max_b:       # a ragged batch runs eagerly, on the caller's tensors
            return None
        if self._real[s] is None:
            self._real[s] = [torch.empty_like(t) for t in like]
        return self._real[s]

    def _d_gen_set(self, s):
        def body(mouse, human, mouse_cat, human_cat, noise):
            out = {}

            def gen(which, x, lab):
                out[which] = self.gen_d[s][which].forward(x, lab, noise)
            self._two(lambda: gen("g_m2h", mouse, mouse_cat), lambda: gen("g_h2m", human, human_cat))
            self._set_d_state(s, (mouse, human, mouse_cat, human_cat, noise, out["g_m2h"], out["g_h2m"]))
        return body

    def _set_d_state(self, s, state):
        self._d_states[s] = state
        if state[0].shape[0] == self.max_b:
            self._d_states_full[s] = state

    def _d_crit_set(self, s):
        def body(noise):
            self._d_state = self._d_states[s]
            self._d_critics()
            self._d_update()
        return body

    def _g_tail(self, noise):
        self._g_rest()
        self._g_backward()
        self._g_update()

    def _pair_body(self, s, mode):
        """The critics' part of a critic step on workspace set s with a generator forward beside it on the lane:
        mode "n" - the forward of the next critic step (into the other set), mode "g" - the first forward of this
        iteration's generator step, followed by the rest of that step.  One body, so that it is captured as one HIP
        graph: two graphs launched on two streams do not run side by side, the branches of one graph do."""
        def body(*a):
            mouse, human, mouse_cat, human_cat, noise = a
            main = torch.cuda.current_stream(self.ops.device)
            # the critics' noise first: draws then come in the order of the plain schedule (critic step, then the
            # forward that follows it), so both schedules see the same numbers from one generator
            self._d_noise = self._draw_d_noise(noise, mouse.shape[0])
            # one generator each, both forked from the main stream (or, when the step is being captured without
            # forks, issued on the main stream like everything else)
            single = self._capture_single_stream and torch.cuda.is_current_stream_capturing()
            lanes = (main, main) if single else (self.ahead, self.ahead_sides[0])
            for st in () if single else lanes:
                self._after(st, main)
            passes = self.gen_a if mode == "g" else self.gen_d[1 - s]
            # the generator step's first forward follows the critic step's forward of the same batch (set s)
            src = self.gen_d[s] if (mode == "g" and self.reuse_encoder) else {}
            # The critics' part is issued first and the lanes' forwards after it: the streams make them run side by
            # side either way, but the order of issue is the order of the captured graph's nodes, and with the critics
            # (the longer chain) first the replay takes 2.568 against 2.578 ms.  The draws keep their order (the
            # critics' noise was drawn above).  Captured on one stream the forward stays in front, as in a plain step.
            critics_first = not single
            if critics_first:
                self._d_crit_set(s)(noise)
            with self._Lane(self, noise, lanes[0]):
                out_h = passes["g_m2h"].forward(mouse, mouse_cat, noise, encoder_of=src.get("g_m2h"))
            with self._Lane(self, noise, lanes[1]):
                out_m = passes["g_h2m"].forward(human, human_cat, noise, encoder_of=src.get("g_h2m"))
            if mode == "g":
                self._g_state = (mouse, human, mouse_cat, human_cat, dict(fh=out_h, fm=out_m), noise)
            else:
                self._set_d_state(1 - s, (mouse, human, mouse_cat, human_cat, noise, out_h, out_m))
            if not critics_first:
                self._d_crit_set(s)(noise)
            for st in () if single else lanes:
                self._after(main, st)
            if mode == "g":
                self._g_tail(noise)
        return body

    def _iteration_lookahead(self, i, mouse, human, mouse_cat, human_cat, noise, nxt):
        b = mouse.shape[0]
        assert human.shape[0] == b and b <= self.max_b
        if self._d_scales_b != b:
            lam = float(self.cfg["lambda_gp"])
            self.d_scales.copy_(torch.tensor([-1.0 / b, 1.0 / b, lam / b] * 2, dtype=torch.float64))
            self._d_scales_b = b
        ins = (mouse, human, mouse_cat, human_cat)
        key = lambda ts: tuple((t.data_ptr(), tuple(t.shape)) for t in ts)  # noqa: E731
        do_g = i % self.cfg["nb_critic"] == 0
        if self._pre is not None and self._pre[1] == key(ins):
            s = self._pre[0]
        else:                   # nothing (or some other batch) was issued ahead: forward now
            s = self._dset = 1 - self._dset
            self._run_graph_or_eager(f"dgen{s}", self._d_gen_set(s), ins, noise, b, self._real_set(s, ins))
        self._pre = None
        if self.use_graphs and b == self.max_b and self._d_states_full[s] is not None:
            self._d_states[s] = self._d_states_full[s]
        self._begin_opt(self.arena_d, self.lr_d)
        if do_g:
            self._begin_opt(self.arena_g, self.lr_g)
            self._run_graph_or_eager(f"pair{s}g", self._pair_body(s, "g"), ins, noise, b)
        elif nxt is not None and nxt[0].shape[0] == b:
            self._dset = 1 - s
            self._run_graph_or_eager(f"pair{s}n", self._pair_body(s, "n"), tuple(nxt), noise, b,
                                     self._real_set(1 - s, nxt))
            self._pre = (1 - s, key(nxt))
        else:
            self._run_graph_or_eager(f"dcrit{s}", self._d_crit_set(s), (), noise, b)

    def state_dict(self, name):
        self.flush()
        return self.nets[name].state_dict()
