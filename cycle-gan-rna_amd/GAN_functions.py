"""Drop-in for the reference's GAN_functions.py on MI355X.

Same public names, constructor/forward signatures, state_dict keys and config
dict as /root/reference/GAN_functions.py; the arithmetic runs in the HIP kernels
of libcgrna.so:

  GeneExpressionGenerator / GeneExpressionDiscriminator   (:74-155, :158-191)
      nn.Modules whose layers are ordinary torch.nn containers (so the same
      torch.manual_seed gives the same initial weights and the same
      state_dict), with forward/backward routed through the fused kernels via
      one autograd.Function per module call
  compute_gradient_penalty                                  (:194-204)
      the penalty with its double backward, as one differentiable scalar
  train_enhanced_cycle_gan                                  (:220-550)
      the training loop on the fused step engine (engine.py)
  compute_cell_type_centroids, poisson_loss, plot_metrics, preprocess_data,
  GeneSelfAttention, LearnedSparseSelfAttention, ResidualBlock

Everything needs a HIP device; there is no CPU or eager-PyTorch fallback.
Extra config keys (all optional, defaults keep the reference's behaviour):
  'noise': 'device' (default; noise drawn in HBM) or 'host_replay' (drawn from
           torch's CPU generator with the reference's calls - parity mode)
  'progress': False silences tqdm/prints
  'precision': 'bf16' runs the large dense-layer products of the training step on the bf16 MFMA (float32
           accumulation); default 'fp32'
"""
from __future__ import annotations

import json
import os
import weakref
from collections import OrderedDict
from datetime import datetime

import numpy as np
import torch
import torch.nn as nn

from .engine import Arena, DiscPass, GenPass, Net, StepEngine, split_state, _arena_order
from .autograd_guard import first_order_only
from .noise import DeviceNoise, HostReplayNoise

try:                                   # optional, as in the reference (:21-25)
    import wandb
    has_wandb = True
except ImportError:
    has_wandb = False

_OPS = {}


def _ops(device, precision="fp32"):
    from .ops import HipOps
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("cycle-gan-rna_amd runs on a HIP device only (got device %r)" % (device,))
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, precision)
    if key not in _OPS:
        _OPS[key] = HipOps(torch.device("cuda", index), precision=precision)
    return _OPS[key]


# --------------------------------------------------------------------------- #
# parameter containers (GAN_functions.py:28-71)                                #
# --------------------------------------------------------------------------- #
class GeneSelfAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout_rate=0.1):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout_rate, batch_first=True)
        self.norm = nn.LayerNorm(embed_dim)
        self.num_heads = num_heads

    def forward(self, x):
        return _AttnFn.apply(self, x, *_attn_param_list(self))


class LearnedSparseSelfAttention(nn.Module):
    """The reference computes a gate and then drops it (:48-52); numerically this is
    GeneSelfAttention plus unused `gate` parameters, which are kept for the state_dict."""

    def __init__(self, embed_dim, num_heads, sparsity_target=0.9, temperature=0.5, dropout_rate=0.1):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout_rate, batch_first=True)
        self.gate = nn.Linear(embed_dim, 1)
        self.norm = nn.LayerNorm(embed_dim)
        self.sparsity_target = sparsity_target
        self.temperature = temperature
        self.num_heads = num_heads

    def forward(self, x):
        return _AttnFn.apply(self, x, *_attn_param_list(self))


def _attn_param_list(m):
    a = m.attention
    return [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, m.norm.weight, m.norm.bias]


class _AttnFn(torch.autograd.Function):
    """LayerNorm(x + MHA(x, x, x)) for (B, 16, E) inputs through cg_attn_ln_fwd / cg_attn_ln_bwd."""

    @staticmethod
    def forward(ctx, module, x, *params):
        ops = _ops(x.device)
        e, heads = x.shape[-1], module.num_heads
        if x.dim() != 3 or x.shape[1] != 16:
            raise RuntimeError("the attention kernels handle (B, 16, E) token tensors (GAN_functions.py:81)")
        flat = torch.cat([p.detach().reshape(-1) for p in params]).float()
        xc = x.detach().contiguous().float().view(-1, e)
        ops.attn_bwd_ws(x.shape[0], e, heads)           # raises for a width / head count that is not built
        keep = None
        if module.training:
            keep = torch.empty(x.shape[0] * heads, 16, 16, device=x.device).bernoulli_(0.9).div_(0.9)
        y = torch.empty_like(xc)
        ops.attn_fwd(y, xc, keep, flat, heads)
        ctx.save_for_backward(xc, flat, keep if keep is not None else torch.empty(0, device=x.device))
        ctx.shapes = [p.shape for p in params]
        ctx.xshape = x.shape
        ctx.heads = heads
        return y.view(x.shape)

    @staticmethod
    @first_order_only
    def backward(ctx, dy):
        xc, flat, keep = ctx.saved_tensors
        ops = _ops(dy.device)
        keep = keep if keep.numel() else None
        e = xc.shape[1]
        dx = torch.empty_like(xc)
        dp = torch.zeros_like(flat)
        ws = torch.empty(max(ops.attn_bwd_ws(xc.shape[0] // 16, e, ctx.heads), 1), device=dy.device)
        ops.attn_bwd(dx, dp, ws, dy.contiguous().float().view(-1, e), xc, keep, flat, ctx.heads, False)
        grads, o = [], 0
        for s in ctx.shapes:
            n = int(np.prod(s))
            grads.append(dp[o:o + n].view(s))
            o += n
        return (None, dx.view(ctx.xshape), *grads)


class ResidualBlock(nn.Module):
    def __init__(self, embed_dim):
        super().__init__()
        self.block = nn.Sequential(
            nn.Linear(embed_dim, embed_dim),
            nn.BatchNorm1d(embed_dim),
            nn.ReLU(inplace=True),
            nn.Linear(embed_dim, embed_dim),
            nn.BatchNorm1d(embed_dim),
        )

    def forward(self, x):
        b = self.block
        return _ResBlockFn.apply(self, x, b[0].weight, b[0].bias, b[1].weight, b[1].bias, b[3].weight, b[3].bias,
                                 b[4].weight, b[4].bias)


class _ResBlockFn(torch.autograd.Function):
    """x + BN(Lin(ReLU(BN(Lin x)))) over the B*seq rows (:66-71) through cg_gemm / cg_bn_*."""

    @staticmethod
    def forward(ctx, module, x, w0, b0, g1, be1, w1, b1, g2, be2):
        ops = _ops(x.device)
        bsz, seq, e = x.shape
        z = x.detach().contiguous().float().view(bsz * seq, e)
        t1, r, t2, u = (torch.empty_like(z) for _ in range(4))
        st = [torch.empty(e, device=x.device) for _ in range(4)]
        bn1, bn2 = module.block[1], module.block[4]
        ops.gemm(t1, z, w0.detach(), False, True, bias=b0.detach())
        if module.training:
            ops.bn_fwd(r, t1, g1.detach(), be1.detach(), st[0], st[1], bn1.running_mean, bn1.running_var, 0.0)
            bn1.num_batches_tracked += 1
        else:
            ops.bn_eval(r, t1, g1.detach(), be1.detach(), bn1.running_mean, bn1.running_var, 0.0)
        ops.gemm(t2, r, w1.detach(), False, True, bias=b1.detach())
        if module.training:
            ops.bn_fwd(u, t2, g2.detach(), be2.detach(), st[2], st[3], bn2.running_mean, bn2.running_var, 1.0)
            bn2.num_batches_tracked += 1
        else:
            ops.bn_eval(u, t2, g2.detach(), be2.detach(), bn2.running_mean, bn2.running_var, 1.0)
        y = torch.empty_like(z)
        ops.axpby(y, 1.0, z, 1.0, u)
        if not module.training:
            ctx.eval_mode = True
            return y.view(x.shape)
        ctx.eval_mode = False
        ctx.save_for_backward(z, t1, r, t2, *st, w0, g1, w1, g2)
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    @first_order_only
    def backward(ctx, dy):
        if ctx.eval_mode:
            raise RuntimeError("ResidualBlock backward is implemented for train mode")
        z, t1, r, t2, m1, i1, m2, i2, w0, g1, w1, g2 = ctx.saved_tensors
        ops = _ops(dy.device)
        dyc = dy.contiguous().float().view(z.shape)
        e = z.shape[1]
        dg1, db1, dg2, db2 = (torch.zeros(e, device=dy.device) for _ in range(4))
        dw0, dw1 = torch.zeros_like(w0), torch.zeros_like(w1)
        dt2, dr, dx = torch.empty_like(z), torch.empty_like(z), dyc.clone()
        ops.bn_bwd(dt2, dyc, t2, None, None, m2, i2, g2.detach(), dg2, db2, False, 1.0)
        ops.gemm(dw1, dt2, r, True, False)
        ops.gemm(dr, dt2, w1.detach(), False, False)
        dt1 = dt2
        ops.bn_bwd(dt1, dr, t1, r, None, m1, i1, g1.detach(), dg1, db1, False, 0.0)
        ops.gemm(dw0, dt1, z, True, False)
        ops.gemm(dx, dt1, w0.detach(), False, False, accumulate=True)
        zero = torch.zeros(e, device=dy.device)       # biases feeding a BatchNorm: gradient identically zero
        return (None, dx.view(ctx.xshape), dw0, zero, dg1, db1, dw1, zero.clone(), dg2, db2)


# --------------------------------------------------------------------------- #
# generator / critic (GAN_functions.py:74-191)                                 #
# --------------------------------------------------------------------------- #
class _Backend:
    """Per-module device state: an arena holding the module's parameters (the
    nn.Parameters are re-pointed at views of it), buffers shared with the
    module, and a forward/backward workspace."""

    def __init__(self, module, kind, cfg, in_dim, out_dim):
        self._module = weakref.ref(module)      # the module owns this object: no cycle, no garbage-collector delay
        self.kind, self.cfg, self.in_dim, self.out_dim = kind, cfg, in_dim, out_dim
        self.ops = None
        self.passes = {}
        self.keys = []

    @property
    def module(self):
        return self._module()

    def noise(self, device):
        """Noise for one module call: a source attached by the caller as
        `module._noise_source` (tests replay recorded draws), else fresh draws in HBM."""
        src = getattr(self.module, "_noise_source", None)
        if src is not None:
            return src
        return DeviceNoise(device, seed=int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))

    def ensure(self, device):
        ops = _ops(device)
        m = self.module
        params = OrderedDict(m.named_parameters())
        if self.ops is ops and all(params[k].data_ptr() == self.net.P[k].data_ptr() for k in self.keys[:2]):
            return
        self.ops = ops
        m.to(ops.device)                                        # parameters and BatchNorm buffers into HBM
        params = OrderedDict(m.named_parameters())
        keys = _arena_order(list(params.keys()))
        self.keys = keys
        self.emb_dim = m.embedding_dim
        self.arena = Arena(ops, [(("net", k), tuple(params[k].shape)) for k in keys])
        _, buffers = split_state(m.state_dict())
        self.net = Net("net", self.arena, keys, OrderedDict(), ops)
        for k in keys:
            self.net.P[k].copy_(params[k].detach().to(ops.device).float())
            params[k].data = self.net.P[k]                      # the module now reads/writes the arena
        named_buffers = dict(m.named_buffers())
        for k in buffers:
            if k.endswith("num_batches_tracked"):
                self.net.nbt[k] = int(named_buffers[k])
            else:
                self.net.B[k] = named_buffers[k].data           # BatchNorm buffers are shared, not copied
        self.passes = {}

    def get_pass(self, b):
        """Workspace for one module call.  A call that may be differentiated keeps
        its workspace (the saved activations) to itself until its backward has
        run; calls under no_grad share one per batch size."""
        def make():
            if self.kind == "gen":
                return GenPass(self, self.net, self.in_dim, self.out_dim, b)
            return DiscPass(self, self.net, self.in_dim, b)
        if torch.is_grad_enabled():
            return make()
        if b not in self.passes:
            self.passes[b] = make()
        return self.passes[b]

    def sync_counters(self):
        nb = dict(self.module.named_buffers())
        for k, v in self.net.nbt.items():
            nb[k].fill_(v)


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, be, p, x, cat, *params):
        noise = be.noise(x.device)
        xc = x.detach().contiguous().float()
        out = p.forward(xc, cat, noise, train=be.module.training)
        be.sync_counters()
        ctx.be, ctx.p, ctx.need_dx = be, p, x.requires_grad
        return out.clone(), p.mu().clone(), p.log_var().clone()

    @staticmethod
    @first_order_only
    def backward(ctx, d_out, d_mu, d_lv):
        be, p = ctx.be, ctx.p
        be.arena.g.zero_()
        f = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        if d_out is None:
            d_out = torch.zeros(p.b, p.out_dim, device=be.ops.device)
        dx = p.backward(f(d_out), need_dx=ctx.need_dx, d_mu=f(d_mu), d_lv=f(d_lv))
        return (None, None, dx, None, *[be.net.G[k].clone() for k in be.keys])


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, be, p, x, cat, *params):
        noise = be.noise(x.device)
        xc = x.detach().contiguous().float()
        val, feat = p.forward(xc, cat, noise, train=be.module.training)
        be.sync_counters()
        ctx.be, ctx.p, ctx.need_dx = be, p, x.requires_grad
        return val.clone(), feat.clone()

    @staticmethod
    @first_order_only
    def backward(ctx, d_val, d_feat):
        be, p = ctx.be, ctx.p
        ops = be.ops
        be.arena.g.zero_()
        b = p.b
        dx = torch.zeros(b, p.Gd, device=ops.device) if ctx.need_dx else None
        # route an arbitrary d_val through the head by hand, then reuse the feature-gradient entry
        dfeat = torch.zeros(b, 256, device=ops.device)
        if d_val is not None:
            dv = d_val.contiguous().float().view(b, 1)
            ops.gemm(be.net.G["model.9.weight"], dv, p.a3[:b], True, False, accumulate=True)
            ops.colsum(be.net.G["model.9.bias"], dv, 1.0, True)
            ops.gemm(dfeat, dv, be.net.P["model.9.weight"], False, False)
        if d_feat is not None:
            ops.axpby(dfeat, 1.0, dfeat, 1.0, d_feat.contiguous().float())
        p.backward(None, dfeat=dfeat, param_grads=True, dx_out=dx)
        return (None, None, dx, None, *[be.net.G[k].clone() for k in be.keys])


class GeneExpressionGenerator(nn.Module):
    def __init__(self, input_dim, output_dim, vocab_sizes, config):
        super().__init__()
        self.library_size = config['library_size']
        self.use_attention = config['use_self_attention']
        self.preprocess_type = config['preprocess_type']
        self.latent_dim = config['latent_dim']
        self.seq_len = 16
        self.embed_dim = self.latent_dim // self.seq_len
        assert self.embed_dim * self.seq_len == self.latent_dim, "latent_dim must be divisible by seq_len"
        self.embeddings = nn.ModuleList([nn.Embedding(vs, min(100, vs)) for vs in vocab_sizes])
        self.embedding_dim = sum(min(100, vs) for vs in vocab_sizes)
        hidden = config['hidden_dims']
        enc = [nn.Linear(input_dim, hidden[0]), nn.BatchNorm1d(hidden[0]), nn.ReLU(inplace=True)]
        in_dim = hidden[0]
        for h in hidden[1:]:
            enc += [nn.Linear(in_dim, h), nn.BatchNorm1d(h), nn.ReLU(inplace=True)]
            in_dim = h
        self.encoder = nn.Sequential(*enc)
        self.mu = nn.Linear(hidden[-1], self.latent_dim)
        self.log_var = nn.Linear(hidden[-1], self.latent_dim)
        blocks = []
        for _ in range(config['residual_blocks']):
            rb = ResidualBlock(self.embed_dim)
            if config['use_sparse_attention']:
                at = LearnedSparseSelfAttention(self.embed_dim, config['num_attention_heads'],
                                                sparsity_target=config['sparsity_target'],
                                                temperature=config['temperature'])
            else:
                at = GeneSelfAttention(self.embed_dim, config['num_attention_heads'])
            blocks.append(nn.Sequential(rb, at))
        self.residual_attention_blocks = nn.ModuleList(blocks)
        dec = [nn.Linear(self.latent_dim + self.embedding_dim, hidden[0]), nn.BatchNorm1d(hidden[0]),
               nn.ReLU(inplace=True)]
        in_dim = hidden[0]
        for h in hidden[1:]:
            dec += [nn.Linear(in_dim, h), nn.BatchNorm1d(h), nn.ReLU(inplace=True)]
            in_dim = h
        dec.append(nn.Linear(in_dim, output_dim))
        self.decoder = nn.Sequential(*dec)
        self._cfg = dict(config)
        self._dims = (input_dim, output_dim)
        object.__setattr__(self, "_backend", _Backend(self, "gen", self._cfg, input_dim, output_dim))

    def reparameterize(self, mu, log_var):
        std = torch.exp(0.5 * log_var)
        return mu + torch.randn_like(std) * std

    def forward(self, x, cat_covs=None):
        be = self._backend
        be.ensure(x.device)
        if cat_covs is not None:
            cat_covs = cat_covs.to(x.device)
        params = [dict(self.named_parameters())[k] for k in be.keys]
        return _GenFn.apply(be, be.get_pass(x.shape[0]), x, cat_covs, *params)

    def normalize_to_library_size(self, x):
        ops = _ops(x.device)
        xc = x.detach().contiguous().float()
        y, rs = torch.empty_like(xc), torch.empty(xc.shape[0], device=x.device)
        ops.libsize_fwd(y, xc, rs, self.library_size)
        return y


class GeneExpressionDiscriminator(nn.Module):
    def __init__(self, input_dim, vocab_sizes, config):
        super().__init__()
        self.embeddings = nn.ModuleList([nn.Embedding(vs, min(100, vs)) for vs in vocab_sizes])
        self.embedding_dim = sum(min(100, vs) for vs in vocab_sizes)
        self.model = nn.Sequential(
            nn.Linear(input_dim + self.embedding_dim, 1024), nn.BatchNorm1d(1024), nn.LeakyReLU(0.2, inplace=True),
            nn.Dropout(0.3), nn.Linear(1024, 512), nn.LeakyReLU(0.2, inplace=True), nn.Dropout(0.3),
            nn.Linear(512, 256), nn.LeakyReLU(0.2, inplace=True), nn.Linear(256, 1))
        self._dims = (input_dim,)
        object.__setattr__(self, "_backend", _Backend(self, "disc", dict(config), input_dim, None))

    def forward(self, x, cat_covs=None, return_features=False):
        be = self._backend
        be.ensure(x.device)
        if cat_covs is not None:
            cat_covs = cat_covs.to(x.device)
        params = [dict(self.named_parameters())[k] for k in be.keys]
        out, feat = _DiscFn.apply(be, be.get_pass(x.shape[0]), x, cat_covs, *params)
        return (out, feat) if return_features else out


# --------------------------------------------------------------------------- #
# gradient penalty (GAN_functions.py:194-204)                                  #
# --------------------------------------------------------------------------- #
class _GPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, be, p, real, fake, cat, *params):
        ops = be.ops
        b = real.shape[0]
        noise = be.noise(real.device)
        alpha = noise.rand((b, 1))
        xhat = torch.empty(b, real.shape[1], device=real.device)
        rc, fc = real.detach().contiguous().float(), fake.detach().contiguous().float()
        ops.gp_interp(xhat, rc, fc, alpha)
        p.forward(xhat, cat, noise, train=be.module.training)
        be.sync_counters()
        # first-order chain + penalty with lambda = 1; the parameter gradients land in a scratch arena image
        be.arena.g.zero_()
        red = (torch.zeros(1, 1024, device=real.device), torch.zeros(1, dtype=torch.int32, device=real.device))
        p.gp_backward(1.0, red, 0)
        tot = torch.zeros(1, dtype=torch.float64, device=real.device)
        ops.sum_parts(tot, red, torch.zeros(1, dtype=torch.int32, device=real.device),
                      torch.full((1,), 1.0 / b, dtype=torch.float64, device=real.device))
        ctx.be, ctx.alpha = be, alpha
        ctx.grads = [be.net.G[k].clone() for k in be.keys]
        # gradient w.r.t. the interpolates: v_h0 W1[:, :G]
        dxh = torch.zeros_like(xhat)
        ops.gemm(dxh, p.h1024a[:b], be.net.P["model.0.weight"][:, :p.Gd], False, False)
        ctx.dxh = dxh
        ctx.need = (real.requires_grad, fake.requires_grad)
        return tot.float().reshape(())

    @staticmethod
    @first_order_only
    def backward(ctx, dgp):
        s = dgp.float()
        d_real = ctx.alpha * ctx.dxh * s if ctx.need[0] else None
        d_fake = (1 - ctx.alpha) * ctx.dxh * s if ctx.need[1] else None
        return (None, None, d_real, d_fake, None, *[g * s for g in ctx.grads])


def compute_gradient_penalty(discriminator, real_samples, fake_samples, device, cat_covs=None):
    """mean((||d D(x_hat)/d x_hat||_2 - 1)^2) at x_hat = a*real + (1-a)*fake; the
    returned scalar back-propagates into the critic's parameters (and into
    real/fake) through the hand-derived second-order chain."""
    be = discriminator._backend
    be.ensure(real_samples.device)
    if cat_covs is not None:
        cat_covs = cat_covs.to(real_samples.device)
    params = [dict(discriminator.named_parameters())[k] for k in be.keys]
    return _GPFn.apply(be, be.get_pass(real_samples.shape[0]), real_samples, fake_samples, cat_covs, *params)


def compute_cell_type_centroids(adata, tensor, device):
    """GAN_functions.py:207-213."""
    centroids = {}
    obs = adata.obs['myannotations']
    for ct in obs.unique():
        mask = torch.as_tensor(np.asarray(obs == ct))
        if mask.sum() > 0:
            centroids[ct] = tensor[mask].mean(dim=0).clone().detach().to(device)
    return centroids


def poisson_loss(pred, target):
    """GAN_functions.py:216-217 (defined by the reference, not used by its step)."""
    return torch.mean(pred - target * torch.log(pred + 1e-10))


# --------------------------------------------------------------------------- #
# training loop (GAN_functions.py:220-550)                                     #
# --------------------------------------------------------------------------- #
class _ShuffledIter:
    """One pass over a shuffled, device-resident dataset with the draws a CPU
    DataLoader(shuffle=True) makes from torch's global generator (one int64 when
    the iterator is created, one when the sampler is first advanced)."""

    def __init__(self, owner):
        self.o = owner
        torch.empty((), dtype=torch.int64).random_()
        self.perm = None
        self.pos = 0

    def __next__(self):
        o = self.o
        if self.perm is None:
            gen = torch.Generator()
            gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
            self.perm = torch.randperm(o.n, generator=gen).to(o.x.device)
        if self.pos >= o.n:
            raise StopIteration
        idx = self.perm[self.pos:self.pos + o.bs]
        self.pos += o.bs
        return o.x.index_select(0, idx), o.labels.index_select(0, idx)


class _ShuffledBatches:
    def __init__(self, x, labels, batch_size):
        self.x, self.labels, self.bs, self.n = x, labels, batch_size, x.shape[0]

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        return _ShuffledIter(self)


def _lr_decay_step(cfg, epoch):
    """LRDecayScheduler.step (:236-247): (returned value, lr to write or None)."""
    if epoch < cfg['decay_start_epoch']:
        return cfg['lr'], None
    progress = (epoch - cfg['decay_start_epoch']) / (cfg['epochs'] - cfg['decay_start_epoch'])
    if cfg['decay_type'] == 'cosine':
        new_lr = cfg['lr'] * (1 + np.cos(np.pi * progress)) / 2 * cfg['decay_factor']
    else:
        new_lr = cfg['lr'] * (1 - progress * cfg['decay_factor'])
    return new_lr, new_lr


def train_enhanced_cycle_gan(g_m2h, g_h2m, d_m, d_h, mouse_adata, human_adata,
                             mouse_tensor, human_tensor, cat_tensor, human_umap_coords, umap_model,
                             idx_to_cell_type, config, device, pca_model=None, scaler=None):
    device = torch.device(device)
    ops = _ops(device, config.get('precision', 'fp32'))
    modules = OrderedDict(g_m2h=g_m2h, g_h2m=g_h2m, d_m=d_m, d_h=d_h)
    n_mouse = mouse_tensor.size(0)
    cat = torch.as_tensor(np.asarray(cat_tensor), dtype=torch.long)
    mouse_cat_tensor, human_cat_tensor = cat[:n_mouse].to(device), cat[n_mouse:].to(device)   # :255-256
    mouse_dev, human_dev = mouse_tensor.float().to(device), human_tensor.float().to(device)
    mouse_loader = _ShuffledBatches(mouse_dev, mouse_cat_tensor, config['batch_size'])        # :258-262
    human_loader = _ShuffledBatches(human_dev, human_cat_tensor, config['batch_size'])

    mouse_centroids = compute_cell_type_centroids(mouse_adata, mouse_tensor, device)          # :264-265
    human_centroids = compute_cell_type_centroids(human_adata, human_tensor, device)
    n_types = max(idx_to_cell_type.keys()) + 1

    def table(cents, width):
        t = torch.zeros(n_types, width, device=device)
        for idx, ct in idx_to_cell_type.items():
            if ct in cents:
                t[idx] = cents[ct]
        return t

    eng = StepEngine({k: m.state_dict() for k, m in modules.items()}, config,
                     table(mouse_centroids, mouse_tensor.shape[1]), table(human_centroids, human_tensor.shape[1]),
                     ops, config['batch_size'])
    try:      # the engine's HIP graphs and events are released when training ends, however it ends
        if config.get('noise', 'device') == 'host_replay':
            noise = HostReplayNoise(device)          # the reference's CPU draws, call by call (parity runs)
        else:
            # noise drawn in HBM from the default HIP generator (seeded from torch's global CPU generator, so
            # torch.manual_seed still decides the run); the step is then replayed as HIP graphs and its independent
            # branches run on side streams
            noise = DeviceNoise(device, seed=int(torch.empty((), dtype=torch.int64).random_().item() % (2 ** 31)),
                                default_generator=True)
            if config.get('concurrent_streams', True):
                eng.enable_concurrency()
                if config.get('lookahead', True):
                    # the generators' forward of the next critic step is issued beside the critics' work of this one
                    eng.enable_lookahead()
            if config.get('hip_graphs', True):
                eng.enable_graphs()
        metrics = {k: [] for k in ('g_loss', 'd_m_loss', 'd_h_loss', 'fake_human_var', 'fake_mouse_var', 'cycle_loss',
                                   'celltype_loss', 'fm_loss', 'kl_loss', 'current_lr')}
        timestamp = datetime.now().strftime("%Y%m%d_%H%M")
        output_dir = f"results/enhanced_cyclegan_{timestamp}"
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, 'config.json'), 'w') as f:                              # :276-277
            json.dump(config, f, indent=4)
        use_wandb = config.get('use_wandb', False) and has_wandb
        if use_wandb:
            wandb.init(project="Grok_gene_cycleGAN_lr", config=config, name=f"run_{timestamp}")
        verbose = config.get('progress', True)
        epochs = range(config['epochs'])
        if verbose:
            try:
                from tqdm import tqdm
                epochs = tqdm(epochs, desc="Training")
            except ImportError:
                pass
        key_map = (('g_loss', 'g_loss'), ('d_m_loss', 'd_m_loss'), ('d_h_loss', 'd_h_loss'), ('cycle_loss', 'cycle_loss'),
                   ('celltype_loss', 'celltype_loss'), ('fm_loss', 'fm_loss'), ('kl_loss', 'kl_loss'),
                   ('fake_human_var', 'fake_human_var'), ('fake_mouse_var', 'fake_mouse_var'))
        if float(config.get('lambda_identity', 0.0)) > 0:          # extension (see StepEngine): reported with the others
            key_map = key_map + (('identity_loss', 'identity_loss'),)
            metrics['identity_loss'] = []
        last = {}
        # the metrics of a generator step are read back one generator step late (and at the end of the epoch), so the
        # host does not stop issuing work while they travel; 'deferred_metrics': False reads them at once
        how = "deferred" if config.get('deferred_metrics', True) else True
        for epoch in epochs:
            acc = {k: [] for k, _ in key_map}
            pending = []

            def record(out):
                for k, src in key_map:
                    acc[k].append(out[src])
                return out
            mouse_iter, human_iter = iter(mouse_loader), iter(human_loader)

            def epoch_batches(mouse_iter=mouse_iter, human_iter=human_iter):
                for i in range(max(len(mouse_loader), len(human_loader))):                         # :298-309
                    try:
                        md, mc = next(mouse_iter)
                    except StopIteration:
                        mouse_iter = iter(mouse_loader)
                        md, mc = next(mouse_iter)
                    try:
                        hd, hc = next(human_iter)
                    except StopIteration:
                        human_iter = iter(human_loader)
                        hd, hc = next(human_iter)
                    n = min(md.size(0), hd.size(0))                                                # :315-319
                    yield i, (md[:n], hd[:n], mc[:n], hc[:n])
            # With lookahead one batch is drawn ahead of the iteration that uses it (within the epoch only), so that the
            # engine can be told the next batch; the draws of the shuffling keep their order among themselves.  Without
            # it every batch is drawn when its iteration starts, as in the reference: a loader that restarts in the
            # middle of an epoch draws its new permutation from the global generator between the noise draws of two
            # iterations, and replayed host noise ('noise': 'host_replay') must see exactly that order.
            batches = epoch_batches()
            cur = next(batches, None)
            while cur is not None:
                nxt = next(batches, None) if eng.lookahead else None
                i, batch = cur
                if hasattr(noise, "begin_step"):
                    noise.begin_step()
                out = eng.iteration(i, *batch, noise, nxt=None if nxt is None else nxt[1], metrics=how)
                if out:                                                                            # :411-419
                    pending.append(out)
                    while len(pending) > (1 if how == "deferred" else 0):
                        out = pending.pop(0)
                        last = record(out if isinstance(out, dict) else out.get())
                cur = nxt if eng.lookahead else next(batches, None)
            for out in pending:
                last = record(out if isinstance(out, dict) else out.get())
            if config['use_lr_decay']:                                                             # :421
                current_lr, new_lr = _lr_decay_step(config, epoch)
                if new_lr is not None:
                    eng.set_lr(new_lr)
            else:
                current_lr = config['lr']
            for k, _ in key_map:
                metrics[k].append(np.mean(acc[k]))
            metrics['current_lr'].append(current_lr)
            if verbose:
                print(f"Epoch {epoch + 1}/{config['epochs']}: G Loss: {metrics['g_loss'][-1]:.4f}, "
                      f"D_M Loss: {metrics['d_m_loss'][-1]:.4f}, D_H Loss: {metrics['d_h_loss'][-1]:.4f}, "
                      f"KL Loss: {metrics['kl_loss'][-1]:.4f}, FM Loss: {metrics['fm_loss'][-1]:.4f}, "
                      f"Cycle Loss: {metrics['cycle_loss'][-1]:.4f}, celltype Loss: {metrics['celltype_loss'][-1]:.4f}")
            if use_wandb and last:
                sc = last['scales']
                wandb.log({'epoch': epoch + 1, 'G Loss': metrics['g_loss'][-1], 'D_M Loss': metrics['d_m_loss'][-1],
                           'D_H Loss': metrics['d_h_loss'][-1], 'KL Loss': metrics['kl_loss'][-1],
                           'FM Loss': metrics['fm_loss'][-1], 'Cycle Loss': metrics['cycle_loss'][-1],
                           'Celltype Loss': metrics['celltype_loss'][-1],
                           'Fake Human Variance': metrics['fake_human_var'][-1],
                           'Fake Mouse Variance': metrics['fake_mouse_var'][-1],
                           'Scaledadv': config['lambda_adv'] * sc['adv'] * last['g_adv'],
                           'Scaled cycle': config['lambda_cycle'] * sc['cycle'] * last['cycle_loss'],
                           'Scaled celltype': config['lambda_celltype'] * sc['celltype'] * last['celltype_loss'],
                           'Scaled fm': config['lambda_fm'] * sc['fm'] * last['fm_loss'],
                           'Scaled kl': config['lambda_kl'] * sc['kl'] * last['kl_loss'],
                           'Current LR': metrics['current_lr']})
            if (epoch + 1) % 10 == 0 or epoch == config['epochs'] - 1:                             # :460-540
                _progress_figures(eng, ops, noise, config, epoch, output_dir, mouse_dev, mouse_cat_tensor, human_adata,
                                  human_centroids, human_umap_coords, umap_model, idx_to_cell_type, pca_model, scaler)
            if (epoch + 1) % 20 == 0 or epoch == config['epochs'] - 1:                             # :542-546
                for name in modules:
                    torch.save(eng.state_dict(name), os.path.join(output_dir, f'{name}_epoch_{epoch + 1}.pt'))
        for name, m in modules.items():            # the reference trains its modules in place: hand the weights back
            sd = eng.state_dict(name)
            m.load_state_dict({k: v.to(next(m.parameters()).device) for k, v in sd.items()})
        if use_wandb:
            wandb.finish()
        return metrics, output_dir
    finally:
        eng.close()


def _progress_figures(eng, ops, noise, config, epoch, output_dir, mouse_dev, mouse_cat, human_adata, human_centroids,
                      human_umap_coords, umap_model, idx_to_cell_type, pca_model, scaler):
    """Eval-mode translation of up to 3000 mouse cells and the two progress
    figures (GAN_functions.py:460-540).  The translation runs on the HIP
    kernels; the figures are host-side matplotlib."""
    n = mouse_dev.size(0)
    sub = min(3000, n)
    idx = torch.as_tensor(np.random.choice(n, sub, replace=False), device=mouse_dev.device)
    gp = GenPass(eng, eng.nets["g_m2h"], eng.g_mouse, eng.g_human, sub)
    labels = mouse_cat.index_select(0, idx)
    fake = gp.forward(mouse_dev.index_select(0, idx), labels, noise, train=False)
    fake_np = fake.cpu().numpy()
    if human_umap_coords is None or pca_model is None or scaler is None:
        return
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    sub_types = [idx_to_cell_type[int(i)] for i in labels[:, 0].cpu()]
    coords = umap_model.transform(pca_model.transform(scaler.transform(fake_np)))
    obs = np.asarray(human_adata.obs['myannotations'])
    types = list(set(obs) | set(sub_types))
    color = {ct: plt.cm.tab20(i % 20) for i, ct in enumerate(types)}
    plt.figure(figsize=(10, 8))
    for ct in set(obs):
        m = obs == ct
        plt.scatter(human_umap_coords[m, 0], human_umap_coords[m, 1], color=color[ct], label=f"Real {ct}", s=10, alpha=0.5)
    st = np.asarray(sub_types)
    for ct in set(sub_types):
        m = st == ct
        plt.scatter(coords[m, 0], coords[m, 1], color=color[ct], label=f"Translated {ct}", marker='^', s=50, alpha=0.8)
    plt.title(f"UMAP: Real Human vs Translated Mouse (Epoch {epoch + 1})")
    plt.xlim(human_umap_coords[:, 0].min() - 1, human_umap_coords[:, 0].max() + 1)
    plt.ylim(human_umap_coords[:, 1].min() - 1, human_umap_coords[:, 1].max() + 1)
    plt.legend(loc='center left', bbox_to_anchor=(1, 0.5), fontsize='small')
    plt.savefig(os.path.join(output_dir, f"umap_epoch_{epoch + 1}.png"), bbox_inches='tight')
    plt.close()
    translated = {ct: scaler.transform(fake_np[st == ct]).mean(axis=0) for ct in set(sub_types)}
    real = {ct: scaler.transform(c.cpu().numpy().reshape(1, -1))[0] for ct, c in human_centroids.items()}
    common = sorted(set(real) & set(translated))
    correlations = []
    if len(common) > 1:
        r = np.stack([real[ct] for ct in common])
        t = np.stack([translated[ct] for ct in common])
        rc, tc = r - r.mean(0), t - t.mean(0)
        den = np.sqrt((rc ** 2).sum(0) * (tc ** 2).sum(0))
        with np.errstate(invalid='ignore', divide='ignore'):
            corr = (rc * tc).sum(0) / den
        correlations = corr[~np.isnan(corr)].tolist()
    plt.figure(figsize=(8, 6))
    plt.hist(correlations, bins=50)
    plt.title(f"Centroid Correlation Histogram (Epoch {epoch + 1}) - Scaled")
    plt.xlabel("Pearson Correlation Coefficient")
    plt.ylabel("Frequency")
    plt.savefig(os.path.join(output_dir, f"centroid_correlations_epoch_{epoch + 1}_scaled.png"))
    plt.close()


def plot_metrics(metrics, output_dir):
    """GAN_functions.py:553-578."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    fig, axs = plt.subplots(4, 2, figsize=(15, 20))
    for ax, series, title in (
            (axs[0, 0], (('g_loss', 'Generator Loss'), ('d_m_loss', 'Mouse Disc Loss'), ('d_h_loss', 'Human Disc Loss')), 'Losses'),
            (axs[0, 1], (('cycle_loss', 'Cycle Loss'), ('celltype_loss', 'Celltype Loss')), 'Component Losses'),
            (axs[1, 0], (('fake_human_var', 'Fake Human Variance'), ('fake_mouse_var', 'Fake Mouse Variance')), 'Output Variance'),
            (axs[1, 1], (('fm_loss', 'Feature Matching Loss'), ('kl_loss', 'KL Divergence Loss')), 'Additional Losses')):
        for k, label in series:
            ax.plot(metrics[k], label=label)
        ax.legend()
        ax.set_title(title)
    plt.tight_layout()
    plt.savefig(os.path.join(output_dir, 'training_metrics.png'))
    plt.close()


def preprocess_data(adata, layer_type='counts', library_size=None):
    """GAN_functions.py:581-610: AnnData layer -> dense float32 tensor (host-side I/O)."""
    if layer_type not in ['counts', 'log_normalized']:
        raise ValueError("layer_type must be 'counts' or 'log_normalized'")
    if layer_type == 'counts':
        data = adata.layers['counts']
        if library_size is not None:
            import scanpy as sc
            sc.pp.normalize_total(adata, target_sum=library_size, layer='counts')
            data = adata.layers['counts']
    else:
        data = adata.layers['log_normalized']
    import scipy.sparse
    if scipy.sparse.issparse(data):
        data = data.toarray()
    return torch.tensor(data, dtype=torch.float32)


def create_default_config():
    """GAN_main.py:47-77."""
    return {
        'epochs': 800, 'batch_size': 64, 'lr': 2e-4, 'hidden_dims': [2048, 1024, 512], 'latent_dim': 256,
        'num_attention_heads': 4, 'residual_blocks': 4, 'nb_critic': 4, 'lambda_adv': 1, 'lambda_cycle': 1,
        'lambda_celltype': 0.2, 'lambda_gp': 13, 'lambda_fm': 0.3, 'lambda_kl': 0.1, 'use_self_attention': False,
        'use_sparse_attention': False, 'sparsity_target': 0.9, 'temperature': 0.5, 'library_size': 10000,
        'use_lr_decay': True, 'decay_start_epoch': 100, 'decay_type': 'cosine', 'decay_factor': 0.1,
        'use_wandb': has_wandb, 'device': 'cuda' if torch.cuda.is_available() else 'cpu',
        'preprocess_type': 'counts', 'cycle_loss_type': 'poisson',
    }
