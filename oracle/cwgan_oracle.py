"""CPU restatement of the conditional WGAN-GP of cWGAN_my.ipynb (cells 1-3): Generator (cell 1), Discriminator
(cell 2), compute_gradient_penalty and the train_gan loop body (cell 3).

TEST INFRASTRUCTURE ONLY: checker for tests/ and generator of golden fixtures; nothing under the product package
imports it.  Pinned to the reference by oracle/make_golden_cwgan.py, which executes the notebook's own cells in the
build container and compares outputs, penalty and every gradient in float64.

Functional style like cyclegan_oracle: parameters are plain dicts keyed like the notebook modules' state_dict,
noise comes from a source object so the device path can replay it.
"""
from __future__ import annotations

from typing import Dict, List

import torch
from torch import Tensor

from .cyclegan_oracle import AdamAmsgrad, GlobalRngNoise, batchnorm, linear

G_HIDDEN = (256, 512, 1024)      # cell 1: fixed generator widths
D_HIDDEN = (1024, 512, 256)      # cell 2: fixed critic widths
D_DROP = 0.3
D_SLOPE = 0.2


def embed(P: Dict[str, Tensor], cat: Tensor) -> Tensor:
    """cell 1/2 forward: cat([emb_i(cat[:, i])], dim=1)."""
    return torch.cat([P[f"embeddings.{i}.weight"][cat[:, i]] for i in range(cat.shape[1])], dim=1)


def generator_forward(P: Dict[str, Tensor], z: Tensor, cat: Tensor, num: Tensor, library_size: float, train=True):
    """Generator.forward + normalize_to_library_size (cell 1): three Linear-BatchNorm-ReLU layers, Linear, then
    relu(x) * library_size / (sum_row relu(x) + 1e-10)."""
    h = torch.cat([z, embed(P, cat), num], dim=1)
    for li in range(3):
        h = linear(h, P[f"network.{3 * li}.weight"], P[f"network.{3 * li}.bias"])
        h = torch.relu(batchnorm(h, P, f"network.{3 * li + 1}", train))
    raw = linear(h, P["network.9.weight"], P["network.9.bias"])
    x = torch.relu(raw)
    return x * (library_size / (x.sum(dim=1, keepdim=True) + 1e-10))


def discriminator_forward(P: Dict[str, Tensor], x: Tensor, cat: Tensor, num: Tensor, noise, train=True):
    """Discriminator.forward without the optional negative-value branch (cell 2, use_neg_detector=False, the
    value main() passes): three Linear-LeakyReLU(0.2)-Dropout(0.3) layers and a Linear head."""
    h = torch.cat([x, embed(P, cat), num], dim=1)
    for li in range(3):
        h = linear(h, P[f"main_network.{3 * li}.weight"], P[f"main_network.{3 * li}.bias"])
        h = torch.nn.functional.leaky_relu(h, D_SLOPE)
        if train:
            h = h * noise.keep(tuple(h.shape), D_DROP, h.dtype)
    return linear(h, P["main_network.9.weight"], P["main_network.9.bias"])


def gradient_penalty(PD, real: Tensor, fake: Tensor, cat, num, noise, train=True):
    """compute_gradient_penalty (cell 3): alpha ~ U(0,1) per row, mean((||d D(xhat)/d xhat||_2 - 1)^2)."""
    alpha = noise.rand((real.shape[0], 1)).to(real.dtype)
    xhat = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    d = discriminator_forward(PD, xhat, cat, num, noise, train)
    grads = torch.autograd.grad(d, xhat, grad_outputs=torch.ones_like(d), create_graph=True, retain_graph=True)[0]
    return ((grads.norm(2, dim=1) - 1) ** 2).mean()


def param_keys(sd: Dict[str, Tensor]) -> List[str]:
    return [k for k in sd if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


class CwganOracleTrainer:
    """train_gan's loop body (cell 3): nb_critic critic updates, then one generator update."""

    def __init__(self, g_state, d_state, config, noise=None):
        self.cfg = dict(config)
        self.G = {k: v.clone() for k, v in g_state.items()}
        self.D = {k: v.clone() for k, v in d_state.items()}
        for k in param_keys(self.G):
            self.G[k].requires_grad_(True)
        for k in param_keys(self.D):
            self.D[k].requires_grad_(True)
        kw = dict(lr=config["lr"], betas=(config.get("beta1", 0.5), config.get("beta2", 0.9)), eps=config.get("eps", 1e-8))
        self.opt_g = AdamAmsgrad([self.G[k] for k in param_keys(self.G)], **kw)
        self.opt_d = AdamAmsgrad([self.D[k] for k in param_keys(self.D)], **kw)
        self.noise = noise or GlobalRngNoise()
        self.dtype = next(iter(self.G.values())).dtype

    def d_step(self, real, cat, num):
        cfg, nz = self.cfg, self.noise
        self.opt_d.zero_grad()
        z = nz.randn((real.shape[0], cfg["latent_dim"]), torch.float32).to(self.dtype)
        fake = generator_forward(self.G, z, cat, num, cfg.get("library_size", 10000), True).detach()
        real_v = discriminator_forward(self.D, real, cat, num, nz)
        fake_v = discriminator_forward(self.D, fake, cat, num, nz)
        gp = gradient_penalty(self.D, real, fake, cat, num, nz)
        loss = -real_v.mean() + fake_v.mean() + cfg.get("lambda_gp", 10) * gp
        loss.backward()
        self.opt_d.step()
        return dict(d_loss=float(loss.detach()), gp=float(gp.detach()))

    def g_step(self, batch, cat, num, neg_weight=0.0):
        cfg, nz = self.cfg, self.noise
        self.opt_g.zero_grad()
        z = nz.randn((batch, cfg["latent_dim"]), torch.float32).to(self.dtype)
        fake = generator_forward(self.G, z, cat, num, cfg.get("library_size", 10000), True)
        fake_v = discriminator_forward(self.D, fake, cat, num, nz)
        g_main = -fake_v.mean()
        neg_mask = (fake < 0).to(fake.dtype)                 # Generator.get_negative_penalty (cell 1)
        neg_prop = neg_mask.mean()
        neg_mag = (fake * neg_mask).abs().mean()
        loss = g_main + neg_weight * neg_mag
        for k in param_keys(self.D):                         # the notebook lets these accumulate; they are zeroed
            self.D[k].grad = None                            # before the next critic update and never used
        loss.backward()
        self.opt_g.step()
        return dict(g_loss=float(loss.detach()), g_loss_main=float(g_main.detach()), neg_prop=float(neg_prop.detach()),
                    neg_mag=float(neg_mag.detach()),
                    fake=fake.detach())

    def iteration(self, real, cat, num, neg_weight=0.0):
        out = {}
        for _ in range(self.cfg["nb_critic"]):
            out.update(self.d_step(real, cat, num))
        out.update(self.g_step(real.shape[0], cat, num, neg_weight))
        return out


def negative_penalty_weight(cfg, epoch):
    """get_negative_penalty_weight (cell 3)."""
    start = cfg.get("negative_penalty_start_epoch", 50)
    if epoch < start:
        return 0.0
    ramp = (epoch - start) / cfg.get("negative_penalty_ramp_epochs", 50)
    return cfg.get("max_negative_penalty", 10.0) * min(1.0, max(0.0, ramp))


class DropLastBatches:
    """DataLoader(TensorDataset(x), batch_size, shuffle=True, drop_last=True) with num_workers=0 (cell 4), including
    its draws from the global generator: a base seed when the iterator is created, the sampler's seed when it is
    first advanced."""

    def __init__(self, x, batch_size):
        self.x, self.bs = x, batch_size

    def __len__(self):
        return self.x.shape[0] // self.bs

    def __iter__(self):
        torch.empty((), dtype=torch.int64).random_()
        perm = None
        for b in range(len(self)):
            if perm is None:
                gen = torch.Generator()
                gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
                perm = torch.randperm(self.x.shape[0], generator=gen)
            yield self.x[perm[b * self.bs:(b + 1) * self.bs]]


def train(trainer: CwganOracleTrainer, x_train: Tensor, cat_covs: Tensor, num_covs: Tensor):
    """train_gan's epoch/batch scaffolding with the RNG draws of DataLoader(shuffle=True, drop_last=True) on a
    TensorDataset and of the covariate sampling, in the reference's order.  Returns the per-epoch averages the
    notebook logs."""
    cfg = trainer.cfg
    loader = DropLastBatches(x_train, cfg["batch_size"])
    log = []
    for epoch in range(cfg["epochs"]):
        d_losses, g_losses, g_main, neg = [], [], [], []
        w = negative_penalty_weight(cfg, epoch)
        for real in loader:
            idx = torch.randint(0, cat_covs.shape[0], (real.shape[0],))
            cat, num = cat_covs[idx], num_covs[idx]
            for _ in range(cfg["nb_critic"]):
                d_losses.append(trainer.d_step(real, cat, num)["d_loss"])
            o = trainer.g_step(real.shape[0], cat, num, w)
            g_losses.append(o["g_loss"])
            g_main.append(o["g_loss_main"])
            neg.append((o["neg_prop"], o["neg_mag"]))
        mean = lambda v: float(sum(v) / len(v))  # noqa: E731
        log.append(dict(epoch=epoch, d_loss=mean(d_losses), g_loss=mean(g_losses), g_loss_main=mean(g_main),
                        negative_proportion=mean([n[0] for n in neg]), negative_magnitude=mean([n[1] for n in neg]),
                        negative_penalty_weight=w))
    return log
