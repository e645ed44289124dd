"""CPU oracle for the loop scaffolding of train_enhanced_cycle_gan.

TEST INFRASTRUCTURE ONLY (see cyclegan_oracle.py).  Restates
/root/reference/GAN_functions.py:254-262 (paired shuffled loaders), :290-319
(restart-on-exhaustion, truncation to the smaller batch), :421-432 (LR schedule,
epoch means) and the RNG draws of the periodic eval pass (:460-468) around
OracleTrainer.iteration, without files, plots or wandb.
"""
from __future__ import annotations

import numpy as np
import torch

from .cyclegan_oracle import OracleTrainer, lr_decay


class _ShuffledIter:
    def __init__(self, owner):
        self.o = owner
        # DataLoader iterators draw a base seed from the global generator when created ...
        torch.empty((), dtype=torch.int64).random_()
        self.perm = None
        self.pos = 0

    def __next__(self):
        o = self.o
        if self.perm is None:
            # ... and RandomSampler draws its own seed when first advanced.
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            gen = torch.Generator()
            gen.manual_seed(seed)
            self.perm = torch.randperm(o.x.shape[0], generator=gen)
        if self.pos >= o.x.shape[0]:
            raise StopIteration
        idx = self.perm[self.pos:self.pos + o.bs]
        self.pos += o.bs
        return o.x[idx], o.labels[idx]


class ShuffledBatches:
    """DataLoader(TensorDataset(x, labels), batch_size, shuffle=True) with
    num_workers=0 on CPU, including its draws from the global generator."""

    def __init__(self, x, labels, batch_size):
        self.x, self.labels, self.bs = x, labels, batch_size

    def __len__(self):
        return (self.x.shape[0] + self.bs - 1) // self.bs

    def __iter__(self):
        return _ShuffledIter(self)


METRIC_KEYS = ('g_loss', 'd_m_loss', 'd_h_loss', 'fake_human_var', 'fake_mouse_var',
               'cycle_loss', 'celltype_loss', 'fm_loss', 'kl_loss', 'current_lr')


def oracle_train(trainer: OracleTrainer, mouse, human, mouse_labels, human_labels, cfg, on_iteration=None):
    """mouse_labels / human_labels: int64 (N,1).  Returns the metrics dict of
    GAN_functions.py:267-270 filled like :423-432."""
    mouse_loader = ShuffledBatches(mouse, mouse_labels, cfg['batch_size'])
    human_loader = ShuffledBatches(human, human_labels, cfg['batch_size'])
    metrics = {k: [] for k in METRIC_KEYS}
    for epoch in range(cfg['epochs']):
        acc = {k: [] for k in METRIC_KEYS if k != 'current_lr'}
        mouse_iter, human_iter = iter(mouse_loader), iter(human_loader)
        for i in range(max(len(mouse_loader), len(human_loader))):
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
            n = min(md.shape[0], hd.shape[0])
            md, hd, mc, hc = md[:n], hd[:n], mc[:n], hc[:n]
            out = trainer.iteration(i, md, hd, mc, hc)
            if on_iteration is not None:
                on_iteration(epoch, i, out)
            if 'g_loss' in out:                                   # :411-419 - only on G steps
                for k in acc:
                    acc[k].append(out[k])
        if cfg['use_lr_decay']:                                   # :421, :236-247
            cur, new = lr_decay(cfg['lr'], epoch, cfg['decay_start_epoch'], cfg['epochs'],
                                cfg['decay_factor'], cfg['decay_type'])
            if new is not None:
                trainer.set_lr(new)
        else:
            cur = cfg['lr']
        for k in acc:
            metrics[k].append(float(np.mean(acc[k])))
        metrics['current_lr'].append(float(cur))
        if (epoch + 1) % 10 == 0 or epoch == cfg['epochs'] - 1:   # :460-468: eval forward draws eps
            sub = min(3000, mouse.shape[0])
            torch.randn(sub, cfg['latent_dim'])
    return metrics
