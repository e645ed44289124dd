"""Drop-in for cells 1-3 of the reference's cWGAN_my.ipynb (conditional WGAN-GP, noise -> expression) on MI355X.

Same names, constructor/forward signatures, state_dict keys and config dict as the notebook:

  Generator(x_dim, vocab_sizes, nb_numeric, z_dim, library_size=10000)          (cell 1)
  Discriminator(x_dim, vocab_sizes, nb_numeric, use_neg_detector=False)         (cell 2)
  compute_gradient_penalty(discriminator, real, fake, cat_covs, num_covs, device)   (cell 3)
  train_gan(generator, discriminator, dataloader, cat_covs, num_covs, config, device, score_fn=None, save_fn=None)

The modules are ordinary torch.nn containers (the same torch.manual_seed gives the same initial weights and the same
state_dict as the notebook's classes); their arithmetic runs in the HIP kernels of libcgrna.so through one
autograd.Function per call, and train_gan runs on the fused step engine (cwgan_engine.py).  Everything needs a HIP
device; there is no CPU or eager-PyTorch fallback.  The optional negative-value branch of the critic
(use_neg_detector=True, which the notebook's main() never enables) is not built.

Extra config keys (optional, defaults keep the notebook's behaviour):
  'noise': 'device' (default; noise drawn in HBM, steps replayed as HIP graphs) or 'host_replay' (the notebook's CPU
           draws, call by call - parity mode)
  'progress': False silences the prints (and the device reads they need)
"""
from __future__ import annotations

import weakref
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .cwgan_engine import CriticW, CwganEngine, GenW
from .engine import Arena, Net, split_state
from .autograd_guard import first_order_only
from .noise import DeviceNoise, HostReplayNoise

try:                                   # optional, as in the notebook (cell 0 imports it, cell 3 checks wandb.run)
    import wandb
except ImportError:
    wandb = None


def _ops(device):
    from .GAN_functions import _ops as shared
    return shared(device)


class _Backend:
    """Device state of one module: an arena holding its parameters (the nn.Parameters are re-pointed at views of
    it), BatchNorm buffers shared with the module, and workspaces by batch size."""

    def __init__(self, module, kind):
        self._module = weakref.ref(module)      # the module owns this object: no cycle, no garbage-collector delay
        self.kind = kind
        self.ops = None
        self.keys = []
        self.cfg = {}

    @property
    def module(self):
        return self._module()

    def noise(self, device):
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
        m.to(ops.device)
        params = OrderedDict(m.named_parameters())
        self.keys = list(params.keys())
        self.arena = Arena(ops, [(("net", k), tuple(params[k].shape)) for k in self.keys], second_grad=self.kind == "disc")
        _, buffers = split_state(m.state_dict())
        self.net = Net("net", self.arena, self.keys, OrderedDict(), ops)
        for k in self.keys:
            self.net.P[k].copy_(params[k].detach().float())
            params[k].data = self.net.P[k]
        named_buffers = dict(m.named_buffers())
        for k in buffers:
            if k.endswith("num_batches_tracked"):
                self.net.nbt[k] = int(named_buffers[k])
            else:
                self.net.B[k] = named_buffers[k].data
        self.emb_dims = [e.embedding_dim for e in m.embeddings]

    def make_pass(self, b):
        m = self.module
        if self.kind == "gen":
            return GenW(self, self.net, m.z_dim, self.emb_dims, m.nb_numeric, m.x_dim, b)
        return CriticW(self, self.net, m.x_dim, self.emb_dims, m.nb_numeric, b, 1)

    def sync_counters(self):
        nb = dict(self.module.named_buffers())
        for k, v in self.net.nbt.items():
            nb[k].fill_(v)


def _f(t, device):
    return t.detach().to(device).contiguous().float()


class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, be, p, z, cat, num, *params):
        out = p.forward(_f(z, be.ops.device), cat, num, float(be.module.library_size), train=be.module.training)
        be.sync_counters()
        ctx.be, ctx.p = be, p
        return out.clone()

    @staticmethod
    @first_order_only
    def backward(ctx, d_out):
        be, p = ctx.be, ctx.p
        if not be.module.training:
            raise RuntimeError("Generator backward is implemented for train mode")
        be.arena.g.zero_()
        p.backward(d_out.contiguous().float(), float(be.module.library_size))
        return (None, None, None, None, None, *[be.net.G[k].clone() for k in be.keys])


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, be, p, x, cat, num, *params):
        ops = be.ops
        b = x.shape[0]
        p.load(0, b, _f(x, ops.device), cat, num)
        masks = None
        if be.module.training:
            noise = be.noise(x.device)
            masks = [noise.keep((b, w), 0.3) for w in (1024, 512, 256)]
        val = p.forward(b, 1, masks)
        ctx.be, ctx.p, ctx.b, ctx.need_dx = be, p, b, x.requires_grad
        return val.clone()

    @staticmethod
    @first_order_only
    def backward(ctx, d_val):
        be, p, b = ctx.be, ctx.p, ctx.b
        be.arena.g.zero_()
        dx = torch.empty(b, p.x_dim, device=be.ops.device) if ctx.need_dx else None
        p._chain(slice(0, b), d_val.contiguous().float().view(b, 1), be.net.G, True, ctx.need_dx, dx)
        return (None, None, dx, None, None, *[be.net.G[k].clone() for k in be.keys])


class _GPFn(torch.autograd.Function):
    """mean((||d D(xhat)/d xhat||_2 - 1)^2) with its gradient w.r.t. the critic's parameters (hand-derived double
    backward).  The gradient w.r.t. real/fake is not provided: the notebook passes them detached."""

    @staticmethod
    def forward(ctx, be, p, real, fake, cat, num, *params):
        ops = be.ops
        b = real.shape[0]
        noise = be.noise(real.device)
        alpha = noise.rand((b, 1))
        p.load(0, b, None, cat, num)
        ops.gp_interp(p.x_seg(0, b), _f(real, ops.device), _f(fake, ops.device), alpha)
        masks = [noise.keep((b, w), 0.3) for w in (1024, 512, 256)] if be.module.training else None
        p.forward(b, 1, masks)
        be.arena.g2.zero_()
        red = (torch.zeros(1, 1024, device=ops.device), torch.zeros(1, dtype=torch.int32, device=ops.device))
        p.gp_backward(b, 0, 1.0, red, 0)
        tot = torch.zeros(1, dtype=torch.float64, device=ops.device)
        ops.sum_parts(tot, red, torch.zeros(1, dtype=torch.int32, device=ops.device),
                      torch.full((1,), 1.0 / b, dtype=torch.float64, device=ops.device))
        ctx.grads = [be.net.G2[k].clone() for k in be.keys]
        return tot.float().reshape(())

    @staticmethod
    @first_order_only
    def backward(ctx, dgp):
        s = dgp.float()
        return (None, None, None, None, None, None, *[g * s for g in ctx.grads])


class Generator(nn.Module):
    def __init__(self, x_dim, vocab_sizes, nb_numeric, z_dim, library_size=10000):
        super().__init__()
        self.library_size = library_size
        self.embeddings = nn.ModuleList([nn.Embedding(v, min(50, v)) for v in vocab_sizes])
        input_dim = z_dim + sum(min(50, v) for v in vocab_sizes) + nb_numeric
        self.network = nn.Sequential(
            nn.Linear(input_dim, 256), nn.BatchNorm1d(256), nn.ReLU(),
            nn.Linear(256, 512), nn.BatchNorm1d(512), nn.ReLU(),
            nn.Linear(512, 1024), nn.BatchNorm1d(1024), nn.ReLU(),
            nn.Linear(1024, x_dim))
        self.x_dim, self.z_dim, self.nb_numeric = x_dim, z_dim, nb_numeric
        object.__setattr__(self, "_backend", _Backend(self, "gen"))

    def normalize_to_library_size(self, x):
        ops = _ops(x.device)
        xc = x.detach().contiguous().float()
        y, rs = torch.empty_like(xc), torch.empty(xc.shape[0], device=x.device)
        ops.libsize_fwd(y, xc, rs, float(self.library_size))
        return y

    def forward(self, z, cat_covs, num_covs):
        be = self._backend
        be.ensure(z.device)
        dev = be.ops.device
        params = [dict(self.named_parameters())[k] for k in be.keys]
        num = num_covs.to(dev).float() if self.nb_numeric else None
        return _GenFn.apply(be, be.make_pass(z.shape[0]), z, cat_covs.to(dev), num, *params)

    def get_negative_penalty(self, generated_data):
        negative_mask = (generated_data < 0).float()
        return (generated_data * negative_mask).abs().mean(), negative_mask.mean()


class Discriminator(nn.Module):
    def __init__(self, x_dim, vocab_sizes, nb_numeric, use_neg_detector=False):
        super().__init__()
        if use_neg_detector:
            raise NotImplementedError("the negative-value branch of the critic (use_neg_detector=True) is not built")
        self.use_neg_detector = False
        self.embeddings = nn.ModuleList([nn.Embedding(v, min(50, v)) for v in vocab_sizes])
        input_dim = x_dim + sum(min(50, v) for v in vocab_sizes) + nb_numeric
        self.main_network = nn.Sequential(
            nn.Linear(input_dim, 1024), nn.LeakyReLU(0.2), nn.Dropout(0.3),
            nn.Linear(1024, 512), nn.LeakyReLU(0.2), nn.Dropout(0.3),
            nn.Linear(512, 256), nn.LeakyReLU(0.2), nn.Dropout(0.3),
            nn.Linear(256, 1))
        self.x_dim, self.nb_numeric = x_dim, nb_numeric
        object.__setattr__(self, "_backend", _Backend(self, "disc"))

    def forward(self, x, cat_covs, num_covs):
        be = self._backend
        be.ensure(x.device)
        dev = be.ops.device
        params = [dict(self.named_parameters())[k] for k in be.keys]
        num = num_covs.to(dev).float() if self.nb_numeric else None
        return _DiscFn.apply(be, be.make_pass(x.shape[0]), x, cat_covs.to(dev), num, *params)


def compute_gradient_penalty(discriminator, real_samples, fake_samples, cat_covs, num_covs, device):
    be = discriminator._backend
    be.ensure(real_samples.device)
    dev = be.ops.device
    params = [dict(discriminator.named_parameters())[k] for k in be.keys]
    num = num_covs.to(dev).float() if discriminator.nb_numeric else None
    return _GPFn.apply(be, be.make_pass(real_samples.shape[0]), real_samples, fake_samples, cat_covs.to(dev), num, *params)


# --------------------------------------------------------------------------- #
# train_gan (cell 3)                                                           #
# --------------------------------------------------------------------------- #
class _ResidentBatches:
    """A DataLoader(TensorDataset(x), batch_size, shuffle=True, drop_last=...) replaced by shuffled index_select on a
    copy of x in HBM, with the loader's draws from torch's global generator (one int64 when the iterator is
    created, one when the sampler is first advanced)."""

    def __init__(self, x, batch_size, drop_last):
        self.x, self.bs, self.n, self.drop_last = x, batch_size, x.shape[0], drop_last

    def __len__(self):
        return self.n // self.bs if self.drop_last else (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        torch.empty((), dtype=torch.int64).random_()
        perm = None
        for k in range(len(self)):
            if perm is None:
                gen = torch.Generator()
                gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
                perm = torch.randperm(self.n, generator=gen).to(self.x.device)
            yield (self.x.index_select(0, perm[k * self.bs:(k + 1) * self.bs]),)


def _resident(dataloader, device):
    """The loader's data in HBM when it is the notebook's kind of loader (cell 4); None otherwise."""
    from torch.utils.data import RandomSampler, TensorDataset
    ds = getattr(dataloader, "dataset", None)
    if (isinstance(ds, TensorDataset) and len(ds.tensors) == 1 and isinstance(getattr(dataloader, "sampler", None), RandomSampler)
            and dataloader.num_workers == 0 and dataloader.batch_size is not None and not dataloader.sampler.replacement):
        return _ResidentBatches(ds.tensors[0].float().to(device), dataloader.batch_size, dataloader.drop_last)
    return None


def train_gan(generator, discriminator, dataloader, cat_covs, num_covs, config, device, score_fn=None, save_fn=None):
    device = torch.device(device)
    ops = _ops(device)
    verbose = config.get('progress', True)
    neg_start = config.get('negative_penalty_start_epoch', 50)
    neg_ramp = config.get('negative_penalty_ramp_epochs', 50)
    max_neg = config.get('max_negative_penalty', 10.0)
    cat_covs = torch.tensor(np.asarray(cat_covs), dtype=torch.long).to(device)
    num_covs = torch.tensor(np.asarray(num_covs), dtype=torch.float32).to(device)
    loader = _resident(dataloader, device) or dataloader
    total_batches = len(loader)
    eng = CwganEngine(generator.state_dict(), discriminator.state_dict(), config, ops, dataloader.batch_size)
    if config.get('noise', 'device') == 'host_replay':
        noise = HostReplayNoise(device)
    else:
        noise = DeviceNoise(device, seed=int(torch.empty((), dtype=torch.int64).random_().item() % (2 ** 31)),
                            default_generator=True)
        if config.get('concurrent_streams', True):
            eng.enable_concurrency()
        if config.get('hip_graphs', True):
            eng.enable_graphs()
    if verbose:
        print(f"Starting training for {config['epochs']} epochs...")
        print(f"Total batches per epoch: {total_batches}")
        print(f"Using device: {device}")
        print(f"Using negative detector: {discriminator.use_neg_detector}")
        print(f"Negative penalty starts at epoch: {neg_start}")

    def neg_weight(epoch):
        if epoch < neg_start:
            return 0.0
        return max_neg * min(1.0, max(0.0, (epoch - neg_start) / neg_ramp))

    def hand_back():
        for name, m in (("g", generator), ("d", discriminator)):
            sd = eng.state_dict(name)
            m.load_state_dict({k: v.to(next(m.parameters()).device) for k, v in sd.items()})

    try:      # the engine's HIP graphs and events are released when training ends, however it ends
        for epoch in range(config['epochs']):
            if verbose:
                print(f"\nEpoch [{epoch + 1}/{config['epochs']}]")
            w = neg_weight(epoch)
            eng.begin_log()
            for batch_idx, (real,) in enumerate(loader):
                real = real.to(device).float()
                b = real.size(0)
                idx = torch.randint(0, cat_covs.size(0), (b,)).to(device)
                bc = cat_covs.index_select(0, idx)
                bn = num_covs.index_select(0, idx) if num_covs.shape[1] else None
                for _ in range(config['nb_critic']):
                    if hasattr(noise, "begin_step"):
                        noise.begin_step()
                    eng.d_step(real, bc, bn, noise)
                    eng.log("d")
                if hasattr(noise, "begin_step"):
                    noise.begin_step()
                eng.g_step(b, bc, bn, noise)
                eng.log("g")
                if verbose and batch_idx % 10 == 0:
                    lo = eng.losses()
                    print(f"  Batch [{batch_idx}/{total_batches}] D_loss: {lo['d_loss']:.4f}, G_loss: {lo['g_loss']:.4f}, "
                          f"G_main: {lo['g_loss_main']:.4f}, Neg_prop: 0.000, Neg_mag: 0.000, Neg_weight: {w:.3f}")
            d_losses, g_losses = eng.end_log()
            avg_d, avg_g = float(np.mean(d_losses)), float(np.mean(g_losses))
            if verbose:
                print(f"\nEpoch {epoch + 1} Summary:")
                print(f"  Average D_loss: {avg_d:.4f}")
                print(f"  Average G_loss: {avg_g:.4f}")
                print(f"  Average G_main: {avg_g:.4f}")
                print("  Average Neg_prop: 0.0000")
                print("  Average Neg_mag: 0.0000")
                print(f"  Negative Penalty Weight: {w:.4f}")
            if wandb is not None and getattr(wandb, "run", None) is not None:
                # the generator's output is relu(x) times a positive scale: the negative-value terms are identically zero
                wandb.log({'epoch': epoch, 'd_loss': avg_d, 'g_loss': avg_g, 'g_loss_main': avg_g,
                           'negative_proportion': 0.0, 'negative_magnitude': 0.0, 'negative_penalty_weight': w})
            last = epoch == config['epochs'] - 1
            if (score_fn is not None and epoch % 10 == 0) or (save_fn is not None and (epoch % 20 == 0 or last)):
                hand_back()
            if score_fn is not None and epoch % 10 == 0:
                score = score_fn(generator)
                if verbose:
                    print(f'Epoch {epoch}: Score = {score:.4f}')
            if save_fn is not None and epoch % 20 == 0:
                save_fn(generator, discriminator, epoch)
            if last and save_fn is not None:
                save_fn(generator, discriminator, epoch)
        hand_back()          # the notebook trains its modules in place
    finally:
        eng.close()
