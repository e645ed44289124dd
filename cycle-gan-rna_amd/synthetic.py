"""Synthetic workload of the training step: log-normalised expression with ~10 % non-zeros, cell-type labels,
centroid tables and randomly initialised networks of the reference architecture.  Used by bench.py and by
examples; everything is produced with the package's own modules (no dataset, no checkpoint)."""
from __future__ import annotations

import torch

from . import GAN_functions as GF


def synthetic_expression(n_cells, n_genes, n_types, seed, device="cpu"):
    """(n_cells, n_genes) float32 matrix log1p(max(expm1(0.5 + 0.8 n), 0)) * [u < 0.10] - about 10 % non-zeros in
    [0, 9], the range of log-normalised counts - and labels in which every type occurs."""
    g = torch.Generator(device=device).manual_seed(seed)
    n = torch.randn(n_cells, n_genes, generator=g, device=device)
    u = torch.rand(n_cells, n_genes, generator=g, device=device)
    x = torch.log1p(torch.clamp(torch.expm1(0.5 + 0.8 * n), min=0)) * (u < 0.10).float()
    labels = (torch.arange(n_cells, device=device) % n_types)[torch.randperm(n_cells, generator=g, device=device)]
    return x, labels


def centroid_table(x, labels, n_types):
    """Mean expression per cell type as a dense (n_types, genes) table (GAN_functions.py:207-213)."""
    tab = torch.zeros(n_types, x.shape[1], dtype=x.dtype, device=x.device)
    cnt = torch.zeros(n_types, dtype=x.dtype, device=x.device)
    tab.index_add_(0, labels, x)
    cnt.index_add_(0, labels, torch.ones_like(labels, dtype=x.dtype))
    return tab / cnt.clamp_min(1)[:, None]


def default_config(**over):
    cfg = GF.create_default_config()
    cfg.update(use_wandb=False)
    cfg.update(over)
    return cfg


def random_networks(g_mouse, g_human, vocab_sizes, cfg, seed=0):
    """state_dicts of the four networks, initialised by torch.nn in the reference's construction order
    (GAN_main.py:134-137) under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    mods = {
        "g_m2h": GF.GeneExpressionGenerator(g_mouse, g_human, vocab_sizes, cfg),
        "g_h2m": GF.GeneExpressionGenerator(g_human, g_mouse, vocab_sizes, cfg),
        "d_m": GF.GeneExpressionDiscriminator(g_mouse, vocab_sizes, cfg),
        "d_h": GF.GeneExpressionDiscriminator(g_human, vocab_sizes, cfg),
    }
    return {k: {n: t.detach().clone() for n, t in m.state_dict().items()} for k, m in mods.items()}
