"""CPU oracle for the CycleGAN / cycle-cGAN training step.

TEST INFRASTRUCTURE ONLY.  This file restates, in plain functional torch-CPU
arithmetic, what the reference computes on its hot path; it is imported by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` as the checker / the timed CPU baseline, never by the product
package.

Parity status: the reference ships no tests or golden vectors for this path,
so the restatement is pinned against outputs of the reference itself, run in
the build container by ``oracle/make_golden.py`` (fixtures under
``tests/golden/``).

Every function cites the lines of ``/root/reference/GAN_functions.py`` it
follows.  No ``torch.nn`` module is used: parameters live in flat dicts keyed
by the reference's ``state_dict`` names, normalisation / attention / dropout /
Adam are spelled out, and differentiation is left to ``torch.autograd`` exactly
as in the reference (including the double backward of the gradient penalty).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5
SEQ_LEN = 16  # GAN_functions.py:81


# --------------------------------------------------------------------------- #
# noise: the reference draws from torch's global CPU generator in call order.  #
# --------------------------------------------------------------------------- #
class GlobalRngNoise:
    """Draws noise from the global torch CPU generator with the calls (shape,
    dtype, order) the reference's modules make, so that after the same
    ``torch.manual_seed`` both see the same numbers.

    randn  <- GAN_functions.py:129  (torch.randn_like(std))
    keep   <- nn.Dropout / MHA dropout on CPU: empty_like(x).bernoulli_(1-p).div_(1-p)
    rand   <- GAN_functions.py:195  (torch.rand(B, 1), always float32)
    """

    def randn(self, shape, dtype):
        return torch.randn(shape, dtype=dtype)

    def keep(self, shape, p_drop, dtype):
        return torch.empty(shape, dtype=dtype).bernoulli_(1.0 - p_drop).div_(1.0 - p_drop)

    def rand(self, shape):
        return torch.rand(shape)


class WideNoise(GlobalRngNoise):
    """The float32 draws of GlobalRngNoise, widened: a float64 run of the oracle sees the same noise values as a
    float32 run (and as the reference) after the same torch.manual_seed."""

    def randn(self, shape, dtype):
        return torch.randn(shape, dtype=torch.float32).to(dtype)

    def keep(self, shape, p_drop, dtype):
        return super().keep(shape, p_drop, torch.float32).to(dtype)


class GeneratorNoise:
    """GlobalRngNoise drawing from a generator of its own: after Generator().manual_seed(s) it sees the numbers the
    global generator gives after torch.manual_seed(s), so several replicas of the step (one per emulated rank) can
    be advanced side by side in one process."""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)

    def randn(self, shape, dtype):
        return torch.randn(shape, dtype=dtype, generator=self.gen)

    def keep(self, shape, p_drop, dtype):
        return torch.empty(shape, dtype=dtype).bernoulli_(1.0 - p_drop, generator=self.gen).div_(1.0 - p_drop)

    def rand(self, shape):
        return torch.rand(shape, generator=self.gen)


class RecordedNoise:
    """Replays a list of tensors (in draw order) and checks shapes."""

    def __init__(self, tensors):
        self.tensors = list(tensors)
        self.pos = 0

    def _next(self, shape):
        t = self.tensors[self.pos]
        self.pos += 1
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t

    def randn(self, shape, dtype):
        return self._next(shape).to(dtype)

    def keep(self, shape, p_drop, dtype):
        return self._next(shape).to(dtype)

    def rand(self, shape):
        return self._next(shape)


class RecordingNoise(GlobalRngNoise):
    """GlobalRngNoise that also keeps every draw (to hand the same noise to the
    device path)."""

    def __init__(self):
        self.tensors = []

    def randn(self, shape, dtype):
        t = super().randn(shape, dtype)
        self.tensors.append(t)
        return t

    def keep(self, shape, p_drop, dtype):
        t = super().keep(shape, p_drop, dtype)
        self.tensors.append(t)
        return t

    def rand(self, shape):
        t = super().rand(shape)
        self.tensors.append(t)
        return t


# --------------------------------------------------------------------------- #
# primitive layers                                                             #
# --------------------------------------------------------------------------- #
def linear(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def batchnorm_train(x: Tensor, P: Dict[str, Tensor], prefix: str, update_stats=True) -> Tensor:
    """nn.BatchNorm1d in train mode over the rows of a 2-D input
    (GAN_functions.py:60,63,89,93,117,121,166): biased variance for the
    normalisation, unbiased for running_var, momentum 0.1."""
    n = x.shape[0]
    mean = x.mean(dim=0)
    var = ((x - mean) ** 2).mean(dim=0)
    y = (x - mean) * torch.rsqrt(var + BN_EPS) * P[prefix + ".weight"] + P[prefix + ".bias"]
    if update_stats:
        with torch.no_grad():
            rm, rv = P[prefix + ".running_mean"], P[prefix + ".running_var"]
            rm.mul_(1 - BN_MOMENTUM).add_(mean.detach() * BN_MOMENTUM)
            unbiased = var.detach() * (n / max(n - 1, 1))
            rv.mul_(1 - BN_MOMENTUM).add_(unbiased * BN_MOMENTUM)
            P[prefix + ".num_batches_tracked"] += 1
    return y


def batchnorm_eval(x: Tensor, P: Dict[str, Tensor], prefix: str) -> Tensor:
    rm, rv = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    return (x - rm) * torch.rsqrt(rv + BN_EPS) * P[prefix + ".weight"] + P[prefix + ".bias"]


def batchnorm(x, P, prefix, train):
    return batchnorm_train(x, P, prefix) if train else batchnorm_eval(x, P, prefix)


def leaky_relu(x: Tensor, slope: float) -> Tensor:
    return torch.where(x > 0, x, x * slope)


def layernorm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + LN_EPS) * w + b


def embed(P: Dict[str, Tensor], cat_covs: Optional[Tensor], n_rows: int, emb_dim: int, dtype) -> Tensor:
    """GAN_functions.py:133-137 / :179-183 - one Embedding per covariate column,
    concatenated; zeros when no covariates are given."""
    if cat_covs is None:
        return torch.zeros(n_rows, emb_dim, dtype=dtype)
    cols = []
    i = 0
    while f"embeddings.{i}.weight" in P:
        cols.append(P[f"embeddings.{i}.weight"][cat_covs[:, i]])
        i += 1
    return torch.cat(cols, dim=1)


# --------------------------------------------------------------------------- #
# generator                                                                    #
# --------------------------------------------------------------------------- #
def residual_block(z: Tensor, P, prefix: str, train: bool) -> Tensor:
    """ResidualBlock.forward, GAN_functions.py:66-71."""
    b, s, e = z.shape
    x = z.reshape(b * s, e)
    h = linear(x, P[prefix + ".block.0.weight"], P[prefix + ".block.0.bias"])
    h = torch.relu(batchnorm(h, P, prefix + ".block.1", train))
    h = linear(h, P[prefix + ".block.3.weight"], P[prefix + ".block.3.bias"])
    h = batchnorm(h, P, prefix + ".block.4", train)
    return z + h.reshape(b, s, e)


def self_attention(z: Tensor, P, prefix: str, num_heads: int, noise, train: bool, p_drop=0.1) -> Tensor:
    """GeneSelfAttention.forward, GAN_functions.py:34-36: LayerNorm(x + MHA(x,x,x))
    with nn.MultiheadAttention(embed, heads, dropout=0.1, batch_first=True) on its
    need_weights=True path: softmax((q/sqrt(hd)) k^T) -> dropout -> @ v -> out_proj."""
    b, s, e = z.shape
    hd = e // num_heads
    w_in, b_in = P[prefix + ".attention.in_proj_weight"], P[prefix + ".attention.in_proj_bias"]
    qkv = z @ w_in.t() + b_in                                  # (b, s, 3e), order q,k,v
    q, k, v = qkv[..., :e], qkv[..., e:2 * e], qkv[..., 2 * e:]

    def heads(t):  # (b, s, e) -> (b*heads, s, hd) with index b*heads + h
        return t.reshape(b, s, num_heads, hd).permute(0, 2, 1, 3).reshape(b * num_heads, s, hd)

    q, k, v = heads(q), heads(k), heads(v)
    q = q * math.sqrt(1.0 / hd)
    att = torch.softmax(q @ k.transpose(1, 2), dim=-1)          # (b*heads, s, s)
    if train and p_drop > 0:
        att = att * noise.keep(att.shape, p_drop, att.dtype)
    o = att @ v                                                 # (b*heads, s, hd)
    o = o.reshape(b, num_heads, s, hd).permute(0, 2, 1, 3).reshape(b, s, e)
    o = o @ P[prefix + ".attention.out_proj.weight"].t() + P[prefix + ".attention.out_proj.bias"]
    return layernorm(z + o, P[prefix + ".norm.weight"], P[prefix + ".norm.bias"])


def generator_forward(P: Dict[str, Tensor], x: Tensor, cat_covs: Optional[Tensor], cfg: dict,
                      noise, train: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
    """GeneExpressionGenerator.forward, GAN_functions.py:132-150.  ``P`` holds
    parameters and BatchNorm buffers under the reference's state_dict keys."""
    hidden = cfg["hidden_dims"]
    latent = cfg["latent_dim"]
    e_tok = latent // SEQ_LEN
    emb_dim = sum(P[k].shape[1] for k in P if k.startswith("embeddings."))
    emb = embed(P, cat_covs, x.shape[0], emb_dim, x.dtype)
    h = x
    for li in range(len(hidden)):                               # encoder :89-96
        h = linear(h, P[f"encoder.{3 * li}.weight"], P[f"encoder.{3 * li}.bias"])
        h = torch.relu(batchnorm(h, P, f"encoder.{3 * li + 1}", train))
    mu = linear(h, P["mu.weight"], P["mu.bias"])                # :139
    log_var = linear(h, P["log_var.weight"], P["log_var.bias"])
    std = torch.exp(0.5 * log_var)                              # :127-130 (also in eval mode)
    z = mu + noise.randn(std.shape, std.dtype) * std
    b = z.shape[0]
    z = z.reshape(b, SEQ_LEN, e_tok)                            # :142
    for bi in range(cfg["residual_blocks"]):                    # :143-144 (always on)
        z = residual_block(z, P, f"residual_attention_blocks.{bi}.0", train)
        z = self_attention(z, P, f"residual_attention_blocks.{bi}.1", cfg["num_attention_heads"], noise, train)
    z = torch.cat([z.reshape(b, latent), emb], dim=1)           # :145-146
    h = z
    for li in range(len(hidden)):                               # decoder :117-125
        h = linear(h, P[f"decoder.{3 * li}.weight"], P[f"decoder.{3 * li}.bias"])
        h = torch.relu(batchnorm(h, P, f"decoder.{3 * li + 1}", train))
    out = linear(h, P[f"decoder.{3 * len(hidden)}.weight"], P[f"decoder.{3 * len(hidden)}.bias"])
    if cfg.get("preprocess_type", "counts") == "counts":        # :148-155
        out = torch.relu(out)
        out = out * (cfg["library_size"] / (out.sum(dim=1, keepdim=True) + 1e-10))
    return out, mu, log_var


# --------------------------------------------------------------------------- #
# discriminator / gradient penalty                                             #
# --------------------------------------------------------------------------- #
def discriminator_forward(P: Dict[str, Tensor], x: Tensor, cat_covs: Optional[Tensor], noise,
                          train: bool = True) -> Tuple[Tensor, Tensor]:
    """GeneExpressionDiscriminator.forward, GAN_functions.py:178-191; returns
    (validity (B,1), features (B,256))."""
    emb_dim = sum(P[k].shape[1] for k in P if k.startswith("embeddings."))
    emb = embed(P, cat_covs, x.shape[0], emb_dim, x.dtype)
    h = torch.cat([x, emb], dim=1)
    h = linear(h, P["model.0.weight"], P["model.0.bias"])       # :165
    h = leaky_relu(batchnorm(h, P, "model.1", train), 0.2)      # :166-167
    if train:
        h = h * noise.keep(h.shape, 0.3, h.dtype)               # :168
    h = leaky_relu(linear(h, P["model.4.weight"], P["model.4.bias"]), 0.2)   # :169-170
    if train:
        h = h * noise.keep(h.shape, 0.3, h.dtype)               # :171
    feat = leaky_relu(linear(h, P["model.7.weight"], P["model.7.bias"]), 0.2)  # :172-173
    out = linear(feat, P["model.9.weight"], P["model.9.bias"])  # :174
    return out, feat


def gradient_penalty(P, real: Tensor, fake: Tensor, cat_covs, noise, train=True) -> Tensor:
    """compute_gradient_penalty, GAN_functions.py:194-204 (fake is NOT detached)."""
    alpha = noise.rand((real.shape[0], 1)).to(real.dtype)
    inter = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    d_out, _ = discriminator_forward(P, inter, cat_covs, noise, train)
    grads = torch.autograd.grad(outputs=d_out, inputs=inter, grad_outputs=torch.ones_like(d_out),
                                create_graph=True, retain_graph=True, only_inputs=True)[0]
    return ((grads.norm(2, dim=1) - 1) ** 2).mean()


def poisson_loss(pred: Tensor, target: Tensor) -> Tensor:
    """GAN_functions.py:216-217 (defined, not called by the step)."""
    return torch.mean(pred - target * torch.log(pred + 1e-10))


# --------------------------------------------------------------------------- #
# Adam (amsgrad), torch's single-tensor CPU path                               #
# --------------------------------------------------------------------------- #
class AdamAmsgrad:
    """optim.Adam(params, lr, betas=(0.5, 0.9), amsgrad=True), GAN_functions.py:223-225,
    as torch's single-tensor loop executes it; parameters whose grad is None are
    skipped and keep their step count."""

    def __init__(self, params: List[Tensor], lr: float, betas=(0.5, 0.9), eps=1e-8, amsgrad=True):
        self.amsgrad = amsgrad      # False: torch.optim.Adam's default (GAN_optimize_optuna.ipynb cell-1)
        self.params = params
        self.lr = lr
        self.b1, self.b2 = betas
        self.eps = eps
        self.state = [None] * len(params)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            g = p.grad
            if self.state[i] is None:
                self.state[i] = dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p), vmax=torch.zeros_like(p))
            st = self.state[i]
            st["step"] += 1
            t = st["step"]
            st["m"].lerp_(g, 1 - self.b1)
            st["v"].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            if self.amsgrad:
                torch.maximum(st["vmax"], st["v"], out=st["vmax"])
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            denom = ((st["vmax"] if self.amsgrad else st["v"]).sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(st["m"], denom, value=-(self.lr / bc1))


def lr_decay(initial_lr, epoch, decay_start_epoch, total_epochs, decay_factor, decay_type):
    """LRDecayScheduler.step, GAN_functions.py:236-247.  Returns (value returned
    by step(), new lr to write into every optimizer or None when unchanged)."""
    if epoch < decay_start_epoch:
        return initial_lr, None
    progress = (epoch - decay_start_epoch) / (total_epochs - decay_start_epoch)
    if decay_type == "cosine":
        new_lr = initial_lr * (1 + math.cos(math.pi * progress)) / 2 * decay_factor
    else:
        new_lr = initial_lr * (1 - progress * decay_factor)
    return new_lr, new_lr


# --------------------------------------------------------------------------- #
# the training step                                                            #
# --------------------------------------------------------------------------- #
PARAM_SKIP = ("running_mean", "running_var", "num_batches_tracked")


def param_names(P: Dict[str, Tensor]) -> List[str]:
    return [k for k in P if not k.endswith(PARAM_SKIP)]


class OracleTrainer:
    """State + loop body of train_enhanced_cycle_gan, GAN_functions.py:220-419.

    ``nets`` maps 'g_m2h','g_h2m','d_m','d_h' to state-dict-keyed tensor dicts
    (cloned; parameters get requires_grad).  ``centroids_*`` are (n_types, G)
    tables (row = label index) standing in for the per-type dicts of :207-213,
    :264-265."""

    def __init__(self, nets: Dict[str, Dict[str, Tensor]], cfg: dict,
                 mouse_centroids: Tensor, human_centroids: Tensor, noise=None):
        self.cfg = cfg
        self.noise = noise if noise is not None else GlobalRngNoise()
        self.nets = {}
        for name, sd in nets.items():
            P = {k: v.detach().clone() for k, v in sd.items()}
            for k in param_names(P):
                P[k].requires_grad_(True)
            self.nets[name] = P
        gm, gh = self.nets["g_m2h"], self.nets["g_h2m"]
        # :223-225 - one optimizer over both generators (m2h first), lr*1.5.  The opt-in keys restate the curriculum
        # training of GAN_optimize_optuna.ipynb cell-1 (plain Adam with the config's betas, one lr, Poisson cycle
        # loss, loss weights handed in per epoch); absent keys = the reference's step.
        kw = dict(betas=tuple(cfg.get("adam_betas", (0.5, 0.9))), amsgrad=bool(cfg.get("adam_amsgrad", True)))
        self.opt_g = AdamAmsgrad([gm[k] for k in param_names(gm)] + [gh[k] for k in param_names(gh)],
                                 cfg["lr"] * float(cfg.get("g_lr_mult", 1.5)), **kw)
        self.opt_dm = AdamAmsgrad([self.nets["d_m"][k] for k in param_names(self.nets["d_m"])], cfg["lr"], **kw)
        self.opt_dh = AdamAmsgrad([self.nets["d_h"][k] for k in param_names(self.nets["d_h"])], cfg["lr"], **kw)
        self.cycle_poisson = cfg.get("cycle_loss", "l1") == "poisson"
        self.ext_weights = None
        self.mouse_centroids = mouse_centroids
        self.human_centroids = human_centroids
        self.ema = dict(adv=0.0, cycle=0.0, celltype=0.0, fm=0.0, kl=0.0)   # :283-287
        self.scales = None

    def set_loss_weights(self, weights):
        """(adv, cycle, celltype, fm, kl) weights used as they are, rounded to float32 like the device copy."""
        self.ext_weights = [float(torch.tensor(float(w), dtype=torch.float32)) for w in weights]

    def set_lr(self, lr):
        for o in (self.opt_g, self.opt_dm, self.opt_dh):        # :244-246
            o.lr = lr

    def G(self, which, x, cat):
        return generator_forward(self.nets[which], x, cat, self.cfg, self.noise, train=True)

    def D(self, which, x, cat):
        return discriminator_forward(self.nets[which], x, cat, self.noise, train=True)

    def d_step(self, mouse, human, mouse_cat, human_cat, step=True):
        """GAN_functions.py:322-340.  step=False stops before the optimizer steps (gradients left in .grad)."""
        cfg = self.cfg
        self.opt_dm.zero_grad()
        self.opt_dh.zero_grad()
        fake_human, _, _ = self.G("g_m2h", mouse, mouse_cat)
        fake_mouse, _, _ = self.G("g_h2m", human, human_cat)

        real_m, _ = self.D("d_m", mouse, mouse_cat)
        fake_m, _ = self.D("d_m", fake_mouse.detach(), human_cat)
        gp_m = gradient_penalty(self.nets["d_m"], mouse, fake_mouse, human_cat, self.noise)
        d_m_loss = -real_m.mean() + fake_m.mean() + cfg["lambda_gp"] * gp_m

        real_h, _ = self.D("d_h", human, human_cat)
        fake_h, _ = self.D("d_h", fake_human.detach(), mouse_cat)
        gp_h = gradient_penalty(self.nets["d_h"], human, fake_human, mouse_cat, self.noise)
        d_h_loss = -real_h.mean() + fake_h.mean() + cfg["lambda_gp"] * gp_h

        d_m_loss.backward()
        d_h_loss.backward()
        if step:
            self.opt_dm.step()
            self.opt_dh.step()
        return dict(d_m_loss=d_m_loss.item(), d_h_loss=d_h_loss.item(), gp_m=gp_m.item(), gp_h=gp_h.item())

    def g_step(self, mouse, human, mouse_cat, human_cat):
        """GAN_functions.py:343-419."""
        L = self.g_losses(mouse, human, mouse_cat, human_cat)
        if self.ext_weights is None:
            self.g_scales({k: L[k].item() for k in ("g_adv", "g_cycle", "g_celltype", "g_fm", "kl")})
        return self.g_finish(L)

    def g_losses(self, mouse, human, mouse_cat, human_cat):
        """:343-376: forward passes and the loss terms (tensors with their graphs)."""
        cfg = self.cfg
        F = torch.nn.functional
        self.opt_g.zero_grad()
        fake_human, mu_h, lv_h = self.G("g_m2h", mouse, mouse_cat)
        fake_mouse, mu_m, lv_m = self.G("g_h2m", human, human_cat)
        g_adv = -self.D("d_h", fake_human, human_cat)[0].mean() - self.D("d_m", fake_mouse, mouse_cat)[0].mean()
        recov_mouse, _, _ = self.G("g_h2m", fake_human, human_cat)
        lam_id = float(cfg.get("lambda_identity", 0.0))
        g_identity = torch.zeros((), dtype=mouse.dtype)
        if lam_id > 0:      # identity-mapping term of CycleGAN: an extension, GAN_functions.py has none
            idt_mouse, _, _ = self.G("g_h2m", mouse, mouse_cat)
            g_identity = g_identity + (idt_mouse - mouse).abs().mean()
        recov_human, _, _ = self.G("g_m2h", fake_mouse, mouse_cat)
        if lam_id > 0:
            idt_human, _, _ = self.G("g_m2h", human, human_cat)
            g_identity = g_identity + (idt_human - human).abs().mean()
        if self.cycle_poisson:
            g_cycle = poisson_loss(recov_mouse, mouse) + poisson_loss(recov_human, human)
        else:
            g_cycle = (recov_mouse - mouse).abs().mean() + (recov_human - human).abs().mean()
        g_celltype = torch.zeros((), dtype=mouse.dtype)
        if mouse_cat.shape[0] > 0 and human_cat.shape[0] > 0:   # :360-364
            g_celltype = g_celltype + F.mse_loss(fake_human, self.human_centroids[mouse_cat[:, 0]])
            g_celltype = g_celltype + F.mse_loss(fake_mouse, self.mouse_centroids[human_cat[:, 0]])
        _, real_h_fm = self.D("d_h", human, human_cat)          # :366-373
        _, fake_h_fm = self.D("d_h", fake_human, human_cat)
        fm_h = F.mse_loss(real_h_fm.mean(dim=0), fake_h_fm.mean(dim=0))
        _, real_m_fm = self.D("d_m", mouse, mouse_cat)
        _, fake_m_fm = self.D("d_m", fake_mouse, mouse_cat)
        fm_m = F.mse_loss(real_m_fm.mean(dim=0), fake_m_fm.mean(dim=0))
        g_fm = (fm_h + fm_m) / 2
        kl = -0.5 * torch.mean(1 + lv_h - mu_h.pow(2) - lv_h.exp()) \
             - 0.5 * torch.mean(1 + lv_m - mu_m.pow(2) - lv_m.exp())   # :375-376

        return dict(g_adv=g_adv, g_cycle=g_cycle, g_celltype=g_celltype, g_fm=g_fm, kl=kl, g_identity=g_identity,
                    fake_human=fake_human, fake_mouse=fake_mouse)

    def g_scales(self, v):
        """:380-399: moving averages of the five loss values `v` (Python floats) and the scales derived from them."""
        a = 0.3                                                  # :288, :380-384
        e = self.ema
        e["adv"] = a * v["g_adv"] + (1 - a) * e["adv"]
        e["cycle"] = a * v["g_cycle"] + (1 - a) * e["cycle"]
        e["celltype"] = a * v["g_celltype"] + (1 - a) * e["celltype"]
        e["fm"] = a * v["g_fm"] + (1 - a) * e["fm"]
        e["kl"] = a * v["kl"] + (1 - a) * e["kl"]
        s = dict(adv=1 / (abs(e["adv"]) + 1e-8), cycle=1 / (e["cycle"] + 1e-8),       # :387-391
                 celltype=1 / (e["celltype"] + 1e-8), fm=1 / (e["fm"] + 1e-8), kl=1 / (e["kl"] + 1e-8))
        mn = min(s.values())                                     # :394-399
        s = {k: val / mn for k, val in s.items()}
        self.scales = s
        return s

    def g_finish(self, L, step=True):
        """:402-419: weighted sum with the scales of g_scales(), backward, optimizer step (step=False stops before
        it), metrics."""
        cfg = self.cfg
        g_adv, g_cycle, g_celltype, g_fm, kl, g_identity = (L[k] for k in ("g_adv", "g_cycle", "g_celltype", "g_fm",
                                                                             "kl", "g_identity"))
        fake_human, fake_mouse = L["fake_human"], L["fake_mouse"]
        lam_id = float(cfg.get("lambda_identity", 0.0))
        if self.ext_weights is not None:
            w = self.ext_weights
            g_loss = w[0] * g_adv + w[1] * g_cycle + w[2] * g_celltype + w[3] * g_fm + w[4] * kl + lam_id * g_identity
            g_loss.backward()
            if step:
                self.opt_g.step()
            return dict(g_loss=g_loss.item(), g_adv=g_adv.item(), cycle_loss=g_cycle.item(),
                        celltype_loss=g_celltype.item(), fm_loss=g_fm.item(), kl_loss=kl.item(),
                        fake_human_var=torch.var(fake_human.detach(), dim=0).mean().item(),
                        fake_mouse_var=torch.var(fake_mouse.detach(), dim=0).mean().item(),
                        fake_human=fake_human.detach(), fake_mouse=fake_mouse.detach())
        s, e = self.scales, self.ema
        g_loss = (cfg["lambda_adv"] * s["adv"] * g_adv + cfg["lambda_cycle"] * s["cycle"] * g_cycle +
                  cfg["lambda_celltype"] * s["celltype"] * g_celltype + cfg["lambda_fm"] * s["fm"] * g_fm +
                  cfg["lambda_kl"] * s["kl"] * kl)               # :402-406
        if lam_id > 0:
            g_loss = g_loss + lam_id * g_identity
        g_loss.backward()
        if step:
            self.opt_g.step()
        return dict(g_loss=g_loss.item(), g_adv=g_adv.item(), cycle_loss=g_cycle.item(),
                    celltype_loss=g_celltype.item(), fm_loss=g_fm.item(), kl_loss=kl.item(),
                    identity_loss=g_identity.item(),
                    fake_human_var=torch.var(fake_human.detach(), dim=0).mean().item(),   # :418-419
                    fake_mouse_var=torch.var(fake_mouse.detach(), dim=0).mean().item(),
                    scales=dict(s), ema=dict(e),
                    fake_human=fake_human.detach(), fake_mouse=fake_mouse.detach())

    def iteration(self, i, mouse, human, mouse_cat, human_cat, critic=True):
        """One pass of the inner loop body, GAN_functions.py:298-419 (critic=False: generators only)."""
        out = self.d_step(mouse, human, mouse_cat, human_cat) if critic else {}
        if i % self.cfg["nb_critic"] == 0:                       # :342
            out.update(self.g_step(mouse, human, mouse_cat, human_cat))
        return out


# --------------------------------------------------------------------------- #
# helpers shared by tests / bench                                              #
# --------------------------------------------------------------------------- #
def default_config(**over) -> dict:
    """create_default_config(), GAN_main.py:47-77 (use_wandb forced off, device
    left to the caller)."""
    cfg = {
        'epochs': 800, 'batch_size': 64, 'lr': 2e-4, 'hidden_dims': [2048, 1024, 512], 'latent_dim': 256,
        'num_attention_heads': 4, 'residual_blocks': 4, 'nb_critic': 4, 'lambda_adv': 1, 'lambda_cycle': 1,
        'lambda_celltype': 0.2, 'lambda_gp': 13, 'lambda_fm': 0.3, 'lambda_kl': 0.1,
        'use_self_attention': False, 'use_sparse_attention': False, 'sparsity_target': 0.9, 'temperature': 0.5,
        'library_size': 10000, 'use_lr_decay': True, 'decay_start_epoch': 100, 'decay_type': 'cosine',
        'decay_factor': 0.1, 'use_wandb': False, 'device': 'cpu', 'preprocess_type': 'counts',
        'cycle_loss_type': 'poisson',
    }
    cfg.update(over)
    return cfg


def synthetic_expression(n_cells: int, n_genes: int, n_types: int, seed: int):
    """Synthetic log-normalised matrix: log1p(max(expm1(0.5+0.8 n),0)) * [u<0.10],
    ~10 % non-zeros in about [0, 9]; labels uniform over n_types."""
    g = torch.Generator().manual_seed(seed)
    n = torch.randn(n_cells, n_genes, generator=g)
    u = torch.rand(n_cells, n_genes, generator=g)
    x = torch.log1p(torch.clamp(torch.expm1(0.5 + 0.8 * n), min=0)) * (u < 0.10).float()
    labels = torch.randint(0, n_types, (n_cells,), generator=g)
    return x, labels


def is_gauge_param(key: str, state: dict) -> bool:
    """True for parameters whose gradient is mathematically zero because a
    train-mode BatchNorm removes them again: the bias of a Linear that feeds a
    BatchNorm (GAN_functions.py:59-60,62-63,89,92-93,117,120-121,165-166) and the
    last LayerNorm bias, which is a constant offset into decoder.0 -> BatchNorm.
    Adam divides their rounding-noise gradients by their own magnitude, so their
    trajectories are not reproducible between implementations and do not
    influence any output."""
    if key.endswith(".attention.in_proj_bias"):
        # its key third shifts every score of a query by the same amount, which
        # softmax removes; the tensor is treated as gauge as a whole
        return True
    if not key.endswith(".bias"):
        return False
    stem = key[:-len(".bias")]
    head, _, idx = stem.rpartition(".")
    if idx.isdigit() and f"{head}.{int(idx) + 1}.running_mean" in state:
        return True
    if stem.endswith(".norm"):
        blocks = [int(k.split(".")[1]) for k in state if k.startswith("residual_attention_blocks.")]
        return int(stem.split(".")[1]) == max(blocks)
    return False


def balanced_labels(n_cells: int, n_types: int, seed: int) -> Tensor:
    """Labels in which every type occurs (the reference's centroid lookup,
    GAN_functions.py:363-364, raises KeyError for a type absent from the other
    species' data)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.arange(n_cells) % n_types)[torch.randperm(n_cells, generator=g)]


def centroid_table(x: Tensor, labels: Tensor, n_types: int) -> Tensor:
    """compute_cell_type_centroids, GAN_functions.py:207-213, as a dense table."""
    tab = torch.zeros(n_types, x.shape[1], dtype=x.dtype)
    for t in range(n_types):
        m = labels == t
        if m.sum() > 0:
            tab[t] = x[m].mean(dim=0)
    return tab
