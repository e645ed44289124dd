"""Curriculum training of the hyper-parameter search (GAN_optimize_optuna.ipynb cell-1) without a GPU: the host
functions against their restatement, and the whole epoch loop on the CPU stand-in backend (float64) against the
autograd oracle.  Parity unpinned: the notebook does not execute, so the oracle is this build's reading of it."""
import numpy as np
import torch

from oracle import cyclegan_oracle as O
from oracle import curriculum_oracle as CO
from oracle.ops_cpu import CpuOps
from tests.helpers import make_problem, rel


def tiny_config(**over):
    cfg = dict(epochs=6, batch_size=16, lr=3e-4, hidden_dims=[48, 32], latent_dim=128, num_attention_heads=4,
               residual_blocks=2, lambda_cycle=8.0, lambda_celltype=0.3, lambda_gp=9.0, lambda_kl=0.1, nb_critic=2,
               use_self_attention=True, use_sparse_attention=False, beta1=0.4, beta2=0.97, use_lr_decay=True,
               decay_start_epoch=3, decay_factor=0.4, decay_type='linear', library_size=10000,
               preprocess_type='counts', temperature=0.5, sparsity_target=0.9, device='cpu', use_wandb=False)
    cfg.update(over)
    return cfg


CURRICULUM = dict(warmup_epochs=2, cycle_focus_epochs=3, adversarial_rampup=1, full_training_start=5,
                  initial_cycle_weight=6.0, initial_celltype_weight=0.5, final_adv_weight=0.9, final_fm_weight=0.3,
                  final_kl_weight=0.1)


def test_curriculum_weights_and_trial_sampling():
    from cycle_gan_rna_amd import curriculum as C
    for number in range(6):
        t = C.RandomTrial(number, seed=7)
        cfg = C.suggest_hyperparameters(t, "cpu")
        cur = C.create_curriculum_schedule(t)
        assert cfg['batch_size'] in (16, 32, 64) and 1e-5 <= cfg['lr'] <= 5e-4 and cfg['latent_dim'] in (128, 256, 512)
        assert cfg['num_attention_heads'] in (4, 8, 12, 16) and 3 <= cfg['residual_blocks'] <= 6
        assert list(cfg['hidden_dims']) in [list(h) for h in C.HIDDEN_CHOICES] and 3 <= cfg['nb_critic'] <= 7
        assert 10 <= cur['warmup_epochs'] <= 50 and 50 <= cur['full_training_start'] <= 150
        again = C.RandomTrial(number, seed=7)
        assert C.suggest_hyperparameters(again, "cpu") == cfg and t.params == dict(t.params)
        for epoch in (0, 1, 9, 10, 25, 49, 50, 79, 80, 120, 149, 150, 299):
            assert C.get_curriculum_weights(epoch, cur, cfg) == CO.curriculum_weights(epoch, cur, cfg), epoch
    assert C.adaptive_scale(0.0) == 1.0 and C.adaptive_scale(-0.3) == 1.0 and C.adaptive_scale(1.0) == 0.5
    assert C.check_gan_failure(dict(avg_g_loss=0.05, avg_d_m_loss=6.0))
    assert C.check_gan_failure(dict(avg_g_loss=float('nan'), avg_d_m_loss=0.0))
    assert not C.check_gan_failure(dict(avg_g_loss=1.0, avg_d_m_loss=1.0))


def test_curriculum_training_equals_oracle_in_float64():
    from cycle_gan_rna_amd import curriculum as C
    dtype = torch.float64
    cfg = tiny_config()
    gm, gh, n = 36, 40, 40
    nets, xm, xh, lm, lh = make_problem(gm, gh, n, O.default_config(**{k: cfg[k] for k in (
        'hidden_dims', 'latent_dim', 'num_attention_heads', 'residual_blocks', 'preprocess_type')}), dtype=dtype)
    xm, xh = torch.expm1(xm).round(), torch.expm1(xh).round()
    cm, ch = O.centroid_table(xm, lm, 12), O.centroid_table(xh, lh, 12)
    torch.manual_seed(21)
    hist, tr = CO.train(nets, cfg, CURRICULUM, xm, xh, lm[:, None], lh[:, None], cm, ch, epochs=6)
    trainer = C.CurriculumTrainer(xm, xh, lm, lh, 12, device="cpu", ops=CpuOps(dtype))
    assert rel(trainer.human_centroids, ch) <= 1e-12
    seen = []
    torch.manual_seed(21)
    # evaluation happens inside at epoch 10, 20, ...: not within these 6 epochs
    final, eng = trainer.train_with_curriculum(nets, cfg, CURRICULUM, epochs=6,
                                               on_epoch=lambda e, m, w: seen.append((e, w['lambda_adv'] > 0)))
    assert seen == [(0, False), (1, False), (2, True), (3, False), (4, True), (5, True)]      # ramp starts at weight 0
    got = final['training_history']
    assert len(got) == len(hist) == 6
    for e, (g, x) in enumerate(zip(got, hist)):
        for k in x:
            assert rel(g[k], x[k], 1e-3) <= 2e-5, (e, k, g[k], x[k])
    assert got[0]['avg_d_m_loss'] == 0 and got[0]['avg_adv_loss'] == 0 and got[2]['avg_d_m_loss'] != 0
    assert got[4]['avg_fm_loss'] > 0 and got[3]['avg_fm_loss'] == 0
    for name in ("g_m2h", "g_h2m", "d_m", "d_h"):
        sd = eng.state_dict(name)
        for k, v in sd.items():
            if O.is_gauge_param(k, sd):
                continue
            tol = 1e-4 if k.endswith("running_mean") else 2e-4
            assert rel(v, tr.nets[name][k]) <= tol, (name, k, rel(v, tr.nets[name][k]))
    # evaluation: same cells, same eval-mode noise
    np.random.seed(5)
    idx = torch.as_tensor(np.random.choice(n, n, replace=False))
    torch.manual_seed(9)
    exp = CO.evaluate(tr, cfg, xm, lm[:, None], ch, idx)
    from cycle_gan_rna_amd.noise import HostReplayNoise
    np.random.seed(5)
    torch.manual_seed(9)
    ev = trainer.evaluate_models(eng, HostReplayNoise("cpu", dtype))
    for k in exp:
        assert rel(ev[k], exp[k], 1e-3) <= 1e-4, (k, ev[k], exp[k])
