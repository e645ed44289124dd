"""Shared helpers for the parity tests."""
import torch

from oracle import cyclegan_oracle as O
from oracle import init_params


def small_cfg(**over):
    return O.default_config(hidden_dims=[96, 64, 32], use_wandb=False, **over)


def make_problem(gm, gh, n_cells, cfg, init_seed=3, data_seed=100, dtype=torch.float32):
    """Initial nets (drawn like the reference draws them) + synthetic data."""
    nets = init_params.build_nets(gm, gh, [12], cfg, init_seed)
    if dtype != torch.float32:
        nets = {n: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()} for n, sd in nets.items()}
    xm, _ = O.synthetic_expression(n_cells, gm, 12, seed=data_seed)
    xh, _ = O.synthetic_expression(n_cells, gh, 12, seed=data_seed + 1)
    lm = O.balanced_labels(n_cells, 12, data_seed + 2)
    lh = O.balanced_labels(n_cells, 12, data_seed + 3)
    return nets, xm.to(dtype), xh.to(dtype), lm, lh


def rel(a, b, floor=0.0):
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor, 1e-30))


def update_rel_l2(final, ref_final, init):
    """||(final-init) - (ref-init)|| / ||ref-init||: compares the UPDATE a run
    applied to a tensor, which is what distinguishes a correct optimizer step
    from a skipped or mis-scaled one after only a few steps."""
    d = (final.double() - ref_final.double()).norm()
    n = (ref_final.double() - init.double()).norm()
    return float(d / max(float(n), 1e-30))


LOSS_KEYS = ('d_m_loss', 'd_h_loss', 'g_loss', 'g_adv', 'cycle_loss', 'celltype_loss', 'fm_loss', 'kl_loss',
             'fake_human_var', 'fake_mouse_var')
