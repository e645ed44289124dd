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


_WideNoise = O.WideNoise


def oracle_iteration_grads(nets, cfg, cm, ch, xm, xh, lm, lh, dtype, seed):
    """One critic step + one generator step of the oracle in `dtype` on the noise that torch.manual_seed(seed) gives
    in float32.  Returns (metrics, {(net, key): gradient}) - critic gradients as the critic step leaves them,
    generator gradients as the generator step (run after the critic update, like the loop body) leaves them."""
    cast = lambda t: t.to(dtype) if t.is_floating_point() else t  # noqa: E731
    tr = O.OracleTrainer({n: {k: cast(v) for k, v in sd.items()} for n, sd in nets.items()}, cfg, cast(cm), cast(ch),
                         noise=_WideNoise())
    torch.manual_seed(seed)
    out = tr.d_step(cast(xm), cast(xh), lm, lh, step=False)
    grads = {(n, k): tr.nets[n][k].grad.clone() for n in ("d_m", "d_h") for k in O.param_names(tr.nets[n])
             if tr.nets[n][k].grad is not None}
    tr.opt_dm.step()
    tr.opt_dh.step()
    L = tr.g_losses(cast(xm), cast(xh), lm, lh)
    tr.g_scales({k: L[k].item() for k in ("g_adv", "g_cycle", "g_celltype", "g_fm", "kl")})
    out.update(tr.g_finish(L, step=False))
    grads.update({(n, k): tr.nets[n][k].grad.clone() for n in ("g_m2h", "g_h2m") for k in O.param_names(tr.nets[n])
                  if tr.nets[n][k].grad is not None})
    return out, grads
