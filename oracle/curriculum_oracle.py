"""CPU oracle for the curriculum training of the hyper-parameter search.

TEST INFRASTRUCTURE ONLY (see cyclegan_oracle.py).  Restates the defined part of
/root/reference/GAN_optimize_optuna.ipynb cell-1 (class GrokOptimizer): get_curriculum_weights,
train_with_curriculum's epoch loop (per-epoch weights, moving averages with alpha = 0.3 updated after every
epoch, LR schedule), train_epoch (adaptive scales 1/(1+avg), critics only while lambda_adv > 0, generator step when
i % nb_critic == 0, Poisson cycle loss on counts data, plain Adam with the trial's betas) and evaluate_models'
centroid correlation and cycle L1.  The notebook itself does not run (it calls names GAN_functions.py does not
define), so nothing of the reference pins these numbers: PARITY UNPINNED.  The loss terms the notebook cannot
compute as written (cross-entropy on a one-column critic output, `return_features`) are the ones
GAN_functions.py:347-376 defines.
"""
from __future__ import annotations

import numpy as np
import torch

from . import cyclegan_oracle as O
from .train_loop import ShuffledBatches


def curriculum_weights(epoch, c, base):
    """cell-1 get_curriculum_weights, phase by phase."""
    w = {}
    if epoch < c['warmup_epochs']:
        w['lambda_adv'] = 0.0
        w['lambda_cycle'] = c['initial_cycle_weight']
        w['lambda_celltype'] = c['initial_celltype_weight'] * (epoch / c['warmup_epochs'])
        w['lambda_fm'] = 0.0
        w['lambda_kl'] = base['lambda_kl'] * 0.1
    elif epoch < c['cycle_focus_epochs']:
        w.update(lambda_adv=0.1, lambda_cycle=base['lambda_cycle'], lambda_celltype=base['lambda_celltype'],
                 lambda_fm=0.0, lambda_kl=base['lambda_kl'] * 0.5)
    elif epoch < c['full_training_start']:
        p = (epoch - c['cycle_focus_epochs']) / (c['full_training_start'] - c['cycle_focus_epochs'])
        w.update(lambda_adv=c['final_adv_weight'] * p, lambda_cycle=base['lambda_cycle'],
                 lambda_celltype=base['lambda_celltype'], lambda_fm=c['final_fm_weight'] * p,
                 lambda_kl=base['lambda_kl'])
    else:
        w.update(lambda_adv=c['final_adv_weight'], lambda_cycle=base['lambda_cycle'],
                 lambda_celltype=base['lambda_celltype'], lambda_fm=c['final_fm_weight'], lambda_kl=base['lambda_kl'])
    return w


def trainer_config(config):
    cfg = dict(config)
    cfg.update(adam_betas=(config['beta1'], config['beta2']), adam_amsgrad=False, g_lr_mult=1.0,
               cycle_loss='poisson' if config['preprocess_type'] == 'counts' else 'l1')
    cfg.setdefault('lambda_adv', 0.0)
    cfg.setdefault('lambda_fm', 0.0)
    return cfg


def train(nets, config, curriculum, mouse, human, mouse_cat, human_cat, cent_m, cent_h, epochs):
    """Returns (per-epoch metrics, trainer)."""
    tr = O.OracleTrainer(nets, trainer_config(config), cent_m, cent_h)
    ml = ShuffledBatches(mouse, mouse_cat, config['batch_size'])
    hl = ShuffledBatches(human, human_cat, config['batch_size'])
    avg = dict(adv=0.0, cycle=0.0, celltype=0.0, fm=0.0, kl=0.0)
    alpha = 0.3
    hist = []
    for epoch in range(epochs):
        w = curriculum_weights(epoch, curriculum, config)
        sc = {k: (1.0 / (1.0 + v) if v > 0 else 1.0) for k, v in avg.items()}
        tr.set_loss_weights([w['lambda_adv'] * sc['adv'], w['lambda_cycle'] * sc['cycle'],
                             w['lambda_celltype'] * sc['celltype'], w['lambda_fm'] * sc['fm'],
                             w['lambda_kl'] * sc['kl']])
        acc = {k: [] for k in ('g_loss', 'd_m_loss', 'd_h_loss', 'cycle_loss', 'celltype_loss', 'fm_loss', 'kl_loss',
                               'g_adv')}
        mi, hi = iter(ml), iter(hl)
        for i in range(max(len(ml), len(hl))):
            try:
                md, mc = next(mi)
            except StopIteration:
                mi = iter(ml)
                md, mc = next(mi)
            try:
                hd, hc = next(hi)
            except StopIteration:
                hi = iter(hl)
                hd, hc = next(hi)
            n = min(md.shape[0], hd.shape[0])
            out = tr.iteration(i, md[:n], hd[:n], mc[:n], hc[:n], critic=w['lambda_adv'] > 0)
            for k in acc:
                if k in out:
                    acc[k].append(out[k])
        mean = lambda k: float(np.mean(acc[k])) if acc[k] else 0.0  # noqa: E731
        m = {'avg_g_loss': mean('g_loss'), 'avg_d_m_loss': mean('d_m_loss'), 'avg_d_h_loss': mean('d_h_loss'),
             'avg_cycle_loss': mean('cycle_loss'), 'avg_celltype_loss': mean('celltype_loss'),
             'avg_fm_loss': mean('fm_loss') if w['lambda_fm'] > 0 else 0.0, 'avg_kl_loss': mean('kl_loss'),
             'avg_adv_loss': mean('g_adv') if w['lambda_adv'] > 0 else 0.0}
        for k in avg:
            avg[k] = alpha * m[f'avg_{k}_loss'] + (1 - alpha) * avg[k]
        if config['use_lr_decay']:
            cur, new = O.lr_decay(config['lr'], epoch, config['decay_start_epoch'], config['epochs'],
                                  config['decay_factor'], config.get('decay_type', 'cosine'))
            if new is not None:
                tr.set_lr(new)
            m['lr'] = cur
        hist.append(m)
    return hist, tr


def evaluate(tr, config, mouse, mouse_cat, cent_h, idx):
    """cell-1 evaluate_models without the embedding-based terms: eval-mode translation of mouse[idx], np.corrcoef
    between real and translated centroids per cell type, L1 of the cycle."""
    noise = O.GlobalRngNoise()
    with torch.no_grad():
        x, lab = mouse[idx], mouse_cat[idx]
        fake, _, _ = O.generator_forward(tr.nets["g_m2h"], x, lab, tr.cfg, noise, train=False)
        cors = []
        for t in range(cent_h.shape[0]):
            m = lab[:, 0] == t
            if int(m.sum()) == 0:
                continue
            c = np.corrcoef(cent_h[t].double().numpy(), fake[m].mean(0).double().numpy())[0, 1]
            if not np.isnan(c):
                cors.append(c)
        cyc, _, _ = O.generator_forward(tr.nets["g_h2m"], fake, lab, tr.cfg, noise, train=False)
        cycle = float((x - cyc).abs().mean())
    corr = float(np.mean(cors)) if cors else 0.0
    return {'cell_type_preservation': corr, 'final_cycle_loss': cycle,
            'biological_score': -corr * 100 + 50 + cycle * 10}
