"""Curriculum training and the per-GPU trial runner of the reference's hyper-parameter search
(GAN_optimize_optuna.ipynb cell-1, class GrokOptimizer) on the MI355X step engine.

The notebook cannot run as checked in: it calls `compute_poisson_loss`, `LRDecayScheduler`-by-keyword,
`itertools` and `d(..., return_features=True)`, none of which exist in GAN_functions.py, and applies
`F.cross_entropy` to the critic's one-column output.  What is restated here is the part of it that is defined:

  * `create_curriculum_schedule`, `get_curriculum_weights`, `suggest_hyperparameters` - host functions, as written;
  * `train_with_curriculum` / `train_epoch` - per-epoch loss weights from the curriculum, epoch-level adaptive
    scales 1 / (1 + moving average) with alpha = 0.3, critics trained only while lambda_adv > 0, a generator step
    every nb_critic-th iteration, plain Adam with the trial's betas and one learning rate, the LR schedule, the
    GAN-failure rule, evaluation every 10 epochs;
  * the five loss terms are those GAN_functions.py defines (adversarial :347, cycle, centroid cell-type MSE
    :359-364, feature matching :366-373, KL :375-376); the cycle term is `poisson_loss` (:216-217) on counts
    data and L1 otherwise, as the notebook's train_epoch chooses.

Nothing in the reference pins these results (no fixture, and the notebook does not execute): they are checked
against this build's CPU restatement only.

A trial object is anything with optuna's `suggest_int / suggest_float / suggest_categorical`, `report` and
`should_prune`; `RandomTrial` is a seeded stand-in, because optuna is not a dependency.  `run_trials` is the
scheduler: independent processes, one per GPU, that pull trial numbers from a shared queue - replicas only, no
collective.
"""
from __future__ import annotations

import json
import math
import os
import time

import numpy as np
import torch

from .engine import GenPass, StepEngine
from .noise import DeviceNoise, HostReplayNoise

HIDDEN_CHOICES = ([2048, 1024, 512], [1024, 512, 256], [4096, 2048, 1024])


# --------------------------------------------------------------------------- #
# host functions of GrokOptimizer                                              #
# --------------------------------------------------------------------------- #
def create_curriculum_schedule(trial):
    """GAN_optimize_optuna.ipynb cell-1, create_curriculum_schedule."""
    return {
        'warmup_epochs': trial.suggest_int('warmup_epochs', 10, 50),
        'cycle_focus_epochs': trial.suggest_int('cycle_focus_epochs', 20, 80),
        'adversarial_rampup': trial.suggest_int('adversarial_rampup', 10, 40),
        'full_training_start': trial.suggest_int('full_training_start', 50, 150),
        'initial_cycle_weight': trial.suggest_float('initial_cycle_weight', 5.0, 20.0),
        'initial_celltype_weight': trial.suggest_float('initial_celltype_weight', 0.1, 1.0),
        'final_adv_weight': trial.suggest_float('final_adv_weight', 0.5, 2.0),
        'final_fm_weight': trial.suggest_float('final_fm_weight', 0.1, 0.5),
        'final_kl_weight': trial.suggest_float('final_kl_weight', 0.05, 0.2),
    }


def get_curriculum_weights(epoch, curriculum_config, base_config):
    """GAN_optimize_optuna.ipynb cell-1, get_curriculum_weights: four phases (reconstruction warm-up, cycle focus,
    adversarial ramp, full training)."""
    c, b = curriculum_config, base_config
    if epoch < c['warmup_epochs']:
        progress = epoch / c['warmup_epochs']
        return dict(lambda_adv=0.0, lambda_cycle=c['initial_cycle_weight'],
                    lambda_celltype=c['initial_celltype_weight'] * progress, lambda_fm=0.0,
                    lambda_kl=b['lambda_kl'] * 0.1)
    if epoch < c['cycle_focus_epochs']:
        return dict(lambda_adv=0.1, lambda_cycle=b['lambda_cycle'], lambda_celltype=b['lambda_celltype'],
                    lambda_fm=0.0, lambda_kl=b['lambda_kl'] * 0.5)
    if epoch < c['full_training_start']:
        progress = (epoch - c['cycle_focus_epochs']) / (c['full_training_start'] - c['cycle_focus_epochs'])
        return dict(lambda_adv=c['final_adv_weight'] * progress, lambda_cycle=b['lambda_cycle'],
                    lambda_celltype=b['lambda_celltype'], lambda_fm=c['final_fm_weight'] * progress,
                    lambda_kl=b['lambda_kl'])
    return dict(lambda_adv=c['final_adv_weight'], lambda_cycle=b['lambda_cycle'],
                lambda_celltype=b['lambda_celltype'], lambda_fm=c['final_fm_weight'], lambda_kl=b['lambda_kl'])


def suggest_hyperparameters(trial, device="cuda"):
    """GAN_optimize_optuna.ipynb cell-1, suggest_hyperparameters (same names, ranges and draw order)."""
    config = {
        'epochs': 300,
        'batch_size': trial.suggest_categorical('batch_size', [16, 32, 64]),
        'lr': trial.suggest_float('lr', 1e-5, 5e-4, log=True),
        'hidden_dims': trial.suggest_categorical('hidden_dims', [list(h) for h in HIDDEN_CHOICES]),
        'latent_dim': trial.suggest_categorical('latent_dim', [128, 256, 512]),
        'num_attention_heads': trial.suggest_int('num_attention_heads', 4, 16, step=4),
        'residual_blocks': trial.suggest_int('residual_blocks', 3, 6),
        'lambda_cycle': trial.suggest_float('lambda_cycle', 5.0, 20.0),
        'lambda_celltype': trial.suggest_float('lambda_celltype', 0.1, 0.5),
        'lambda_gp': trial.suggest_float('lambda_gp', 5.0, 20.0),
        'lambda_kl': trial.suggest_float('lambda_kl', 0.05, 0.2),
        'nb_critic': trial.suggest_int('nb_critic', 3, 7),
        'use_self_attention': trial.suggest_categorical('use_self_attention', [True, False]),
        'use_sparse_attention': False,
        'beta1': trial.suggest_float('beta1', 0.0, 0.9),
        'beta2': trial.suggest_float('beta2', 0.9, 0.999),
        'use_lr_decay': trial.suggest_categorical('use_lr_decay', [True, False]),
        'decay_start_epoch': 100,
        'decay_factor': trial.suggest_float('decay_factor', 0.1, 0.5),
        'library_size': 10000,
        'preprocess_type': 'counts',
        'temperature': 0.5,
        'sparsity_target': 0.9,
        'device': str(device),
        'use_wandb': False,
    }
    if config['use_lr_decay']:
        config['decay_type'] = trial.suggest_categorical('decay_type', ['linear', 'cosine'])
    return config


class TrialPruned(Exception):
    """Raised where the notebook raises optuna.TrialPruned."""


class RandomTrial:
    """Seeded stand-in for optuna.trial.Trial: independent uniform (or log-uniform) draws, never prunes."""

    def __init__(self, number=0, seed=42):
        self.number = number
        self.params = {}
        self.reports = []
        self._rng = np.random.RandomState((int(seed) * 1000003 + int(number)) % (2 ** 31 - 1))

    def suggest_int(self, name, low, high, step=1):
        v = int(low + step * self._rng.randint(0, (high - low) // step + 1))
        self.params[name] = v
        return v

    def suggest_float(self, name, low, high, log=False):
        u = float(self._rng.uniform())
        v = math.exp(math.log(low) + u * (math.log(high) - math.log(low))) if log else low + u * (high - low)
        self.params[name] = v
        return v

    def suggest_categorical(self, name, choices):
        v = choices[int(self._rng.randint(0, len(choices)))]
        self.params[name] = v
        return v

    def report(self, value, step):
        self.reports.append((int(step), float(value)))

    def should_prune(self):
        return False


def adaptive_scale(moving_avg):
    """train_epoch: 1 / (1 + moving average) once the average is positive, else 1."""
    return 1.0 / (1.0 + moving_avg) if moving_avg > 0 else 1.0


def check_gan_failure(m):
    """GAN_optimize_optuna.ipynb cell-1, check_gan_failure."""
    if m['avg_g_loss'] < 0.1 and m['avg_d_m_loss'] > 5.0:
        return True
    return bool(m['avg_g_loss'] > 10.0 or m['avg_d_m_loss'] > 10.0 or np.isnan(m['avg_g_loss']))


def engine_config(config):
    """The step-engine view of a trial config: plain Adam with the trial's betas and one learning rate for all
    three optimizers (train_with_curriculum), Poisson cycle loss on counts data (train_epoch)."""
    cfg = dict(config)
    cfg['adam_betas'] = (config['beta1'], config['beta2'])
    cfg['adam_amsgrad'] = False
    cfg['g_lr_mult'] = 1.0
    cfg['cycle_loss'] = 'poisson' if config['preprocess_type'] == 'counts' else 'l1'
    for k in ('lambda_adv', 'lambda_fm'):            # placeholders: every weight is handed in per epoch
        cfg.setdefault(k, 0.0)
    return cfg


# --------------------------------------------------------------------------- #
# training                                                                     #
# --------------------------------------------------------------------------- #
class CurriculumTrainer:
    """Data of one study in HBM + train_with_curriculum / evaluate_models of GrokOptimizer.

    mouse, human: (N, G) float tensors as preprocess_data returns them; mouse_labels, human_labels: int64 (N,) or
    (N, 1) cell-type indices; `ops`: the kernel backend (default: the HIP library on `device`)."""

    def __init__(self, mouse, human, mouse_labels, human_labels, n_types, device="cuda:0", ops=None, precision="fp32",
                 noise="device"):
        if ops is None:
            from .GAN_functions import _ops
            ops = _ops(torch.device(device), precision)
        self.ops = ops
        dev = ops.device
        dtype = getattr(ops, "dtype", torch.float32)        # float32 on the device; a test backend may use float64
        self.mouse = ops.to_device(mouse.to(dtype))
        self.human = ops.to_device(human.to(dtype))
        self.mouse_cat = ops.to_device(mouse_labels.reshape(-1, 1).long())
        self.human_cat = ops.to_device(human_labels.reshape(-1, 1).long())
        self.n_types = int(n_types)
        self.vocab_sizes = [self.n_types]
        self.device = dev
        self.noise_kind = noise
        self.mouse_centroids = self._centroids(self.mouse, self.mouse_cat)
        self.human_centroids = self._centroids(self.human, self.human_cat)

    def _centroids(self, x, cat):
        """compute_cell_type_centroids (GAN_functions.py:207-213) as a dense (n_types, G) table."""
        tab = torch.zeros(self.n_types, x.shape[1], dtype=x.dtype, device=x.device)
        cnt = torch.zeros(self.n_types, dtype=x.dtype, device=x.device)
        tab.index_add_(0, cat[:, 0], x)
        cnt.index_add_(0, cat[:, 0], torch.ones_like(cat[:, 0], dtype=x.dtype))
        return tab / cnt.clamp_min(1)[:, None]

    # -- build_models ---------------------------------------------------------------------------------------
    def build_models(self, config):
        """Initial state dicts of the four networks, drawn like the reference's constructors draw them
        (GAN_optimize_optuna.ipynb cell-1, build_models: g_m2h, g_h2m, d_m, d_h in this order)."""
        from . import GAN_functions as F
        gm, gh = self.mouse.shape[1], self.human.shape[1]
        mods = dict(g_m2h=F.GeneExpressionGenerator(gm, gh, self.vocab_sizes, config),
                    g_h2m=F.GeneExpressionGenerator(gh, gm, self.vocab_sizes, config),
                    d_m=F.GeneExpressionDiscriminator(gm, self.vocab_sizes, config),
                    d_h=F.GeneExpressionDiscriminator(gh, self.vocab_sizes, config))
        return {k: m.state_dict() for k, m in mods.items()}

    def make_engine(self, nets, config):
        eng = StepEngine(nets, engine_config(config), self.mouse_centroids, self.human_centroids, self.ops,
                         config['batch_size'])
        if self.noise_kind == "host_replay" or self.device.type != "cuda":
            noise = HostReplayNoise(self.device, self.mouse.dtype)        # torch's CPU draws, call by call
        else:
            noise = DeviceNoise(self.device, seed=int(torch.empty((), dtype=torch.int64).random_().item() % (2 ** 31)),
                                default_generator=True)
            eng.enable_concurrency()
            eng.enable_graphs()
        return eng, noise

    # -- train_epoch ----------------------------------------------------------------------------------------
    def train_epoch(self, eng, noise, loaders, config, weights, moving):
        """One epoch with fixed weights lambda_k(epoch) x scale_k(moving averages).  Loss values are summed on the
        device and read once at the end of the epoch."""
        mouse_loader, human_loader = loaders
        critic = weights['lambda_adv'] > 0
        w = [weights['lambda_adv'] * adaptive_scale(moving['adv']), weights['lambda_cycle'] * adaptive_scale(moving['cycle']),
             weights['lambda_celltype'] * adaptive_scale(moving['celltype']),
             weights['lambda_fm'] * adaptive_scale(moving['fm']), weights['lambda_kl'] * adaptive_scale(moving['kl'])]
        eng.set_loss_weights(w)
        g_sum = torch.zeros_like(eng.gloss)
        d_sum = torch.zeros_like(eng.d_tot)
        n_g = n_d = 0
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
            n = min(md.size(0), hd.size(0))
            if hasattr(noise, "begin_step"):
                noise.begin_step()
            eng.iteration(i, md[:n], hd[:n], mc[:n], hc[:n], noise, metrics=False, critic=critic)
            if critic:
                d_sum += eng.d_tot
                n_d += 1
            if i % config['nb_critic'] == 0:
                g_sum += eng.gloss
                n_g += 1
        g = (g_sum / max(n_g, 1)).cpu().tolist()
        d = (d_sum / max(n_d, 1)).cpu().tolist()
        fm_on, adv_on = weights['lambda_fm'] > 0, critic
        return {'avg_g_loss': g[5] if n_g else 0, 'avg_d_m_loss': d[0] + d[1] + d[2] if n_d else 0,
                'avg_d_h_loss': d[3] + d[4] + d[5] if n_d else 0, 'avg_cycle_loss': g[1] if n_g else 0,
                'avg_celltype_loss': g[2] if n_g else 0, 'avg_fm_loss': g[3] if (n_g and fm_on) else 0,
                'avg_kl_loss': g[4] if n_g else 0, 'avg_adv_loss': g[0] if (n_g and adv_on) else 0}

    # -- train_with_curriculum ------------------------------------------------------------------------------
    def train_with_curriculum(self, nets, config, curriculum_config, trial=None, epochs=None, on_epoch=None):
        """Returns (final evaluation metrics incl. 'training_history', engine)."""
        epochs = config['epochs'] if epochs is None else int(epochs)
        eng, noise = self.make_engine(nets, config)
        from .GAN_functions import _ShuffledBatches          # DataLoader(shuffle=True) over device-resident data
        loaders = (_ShuffledBatches(self.mouse, self.mouse_cat, config['batch_size']),
                   _ShuffledBatches(self.human, self.human_cat, config['batch_size']))
        moving = dict(adv=0.0, cycle=0.0, celltype=0.0, fm=0.0, kl=0.0)
        alpha = 0.3
        history = []
        best = float('inf')
        for epoch in range(epochs):
            weights = get_curriculum_weights(epoch, curriculum_config, config)
            m = self.train_epoch(eng, noise, loaders, config, weights, moving)
            for k in moving:
                moving[k] = alpha * m[f'avg_{k}_loss'] + (1 - alpha) * moving[k]
            if config['use_lr_decay']:
                m['lr'] = self._lr_step(eng, config, epoch)
            history.append(m)
            if on_epoch is not None:
                on_epoch(epoch, m, weights)
            if epoch % 10 == 0 and epoch > 0:
                ev = self.evaluate_models(eng, noise)
                if trial is not None:
                    trial.report(ev['biological_score'], epoch)
                    if trial.should_prune():
                        raise TrialPruned(f"pruned at epoch {epoch}")
                if check_gan_failure(m):
                    raise TrialPruned("GAN instability")
                best = min(best, ev['biological_score'])
        final = self.evaluate_models(eng, noise)
        final['training_history'] = history
        return final, eng

    @staticmethod
    def _lr_step(eng, config, epoch):
        """LRDecayScheduler.step (GAN_functions.py:236-247) with the trial's decay settings."""
        if epoch < config['decay_start_epoch']:
            return config['lr']
        progress = (epoch - config['decay_start_epoch']) / (config['epochs'] - config['decay_start_epoch'])
        if config.get('decay_type', 'cosine') == 'cosine':
            new_lr = config['lr'] * (1 + math.cos(math.pi * progress)) / 2 * config['decay_factor']
        else:
            new_lr = config['lr'] * (1 - progress * config['decay_factor'])
        eng.set_lr(new_lr)
        return new_lr

    # -- evaluate_models ------------------------------------------------------------------------------------
    def evaluate_models(self, eng, noise, embed=None):
        """Eval-mode translation of up to 1000 mouse cells, centroid correlation per cell type, L1 of the cycle,
        and the combined score (GAN_optimize_optuna.ipynb cell-1, evaluate_models).  `embed`: optional callable
        mapping the translated matrix (numpy) to 2-D coordinates for the silhouette term, which is 0 without it
        (the notebook's own value when it has no PCA / scaler)."""
        n = self.mouse.shape[0]
        sub = min(1000, n)
        idx = torch.as_tensor(np.random.choice(n, sub, replace=False), device=self.device)
        x, lab = self.mouse.index_select(0, idx), self.mouse_cat.index_select(0, idx)
        gm, gh = eng.g_mouse, eng.g_human
        fwd = GenPass(eng, eng.nets["g_m2h"], gm, gh, sub)
        fake = fwd.forward(x, lab, noise, train=False)
        fake_cent = self._centroids(fake, lab)
        present = torch.zeros(self.n_types, dtype=torch.bool, device=self.device)
        present[lab[:, 0]] = True
        r, f = self.human_centroids, fake_cent
        rc, fc = r - r.mean(1, keepdim=True), f - f.mean(1, keepdim=True)
        corr = (rc * fc).sum(1) / torch.sqrt((rc * rc).sum(1) * (fc * fc).sum(1))
        ok = present & ~torch.isnan(corr)
        avg_corr = float(corr[ok].mean()) if bool(ok.any()) else 0.0
        back = GenPass(eng, eng.nets["g_h2m"], gh, gm, sub)
        cyc = back.forward(fake, lab, noise, train=False)
        cycle_loss = float((x - cyc).abs().mean())
        silhouette = 0.0
        if embed is not None:
            from sklearn.metrics import silhouette_score
            labels = lab[:, 0].cpu().numpy()
            if len(set(labels.tolist())) > 1:
                silhouette = float(silhouette_score(embed(fake.cpu().numpy()), labels))
        score = -avg_corr * 100 + (1 - silhouette) * 50 + cycle_loss * 10
        return {'biological_score': score, 'cell_type_preservation': avg_corr, 'silhouette_score': silhouette,
                'final_cycle_loss': cycle_loss}

    # -- objective ------------------------------------------------------------------------------------------
    def objective(self, trial, epochs=None, seed=None, curriculum=None):
        """One trial: sample, build, train, evaluate (GAN_optimize_optuna.ipynb cell-1, objective).  `curriculum`
        replaces the drawn schedule's entries (measurements that want a given phase from epoch 0)."""
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed % (2 ** 32))
        config = suggest_hyperparameters(trial, self.device)
        curriculum = dict(create_curriculum_schedule(trial), **(curriculum or {}))
        nets = self.build_models(config)
        t0 = time.time()
        final, eng = self.train_with_curriculum(nets, config, curriculum, trial, epochs)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        eng.close()             # the trial's HIP graphs and events
        final['train_seconds'] = time.time() - t0
        final['config'] = {k: v for k, v in config.items()}
        final['curriculum'] = curriculum
        return final


# --------------------------------------------------------------------------- #
# trial scheduler: one process per GPU, replicas only                          #
# --------------------------------------------------------------------------- #
def _trial_worker(rank, device_index, make_trainer, trial_queue, result_queue, out_dir, seed, epochs, curriculum):
    device = torch.device("cuda", device_index)
    torch.cuda.set_device(device)
    trainer = make_trainer(device)
    while True:
        number = trial_queue.get()
        if number is None:
            break
        trial = RandomTrial(number, seed)
        rec = {'trial_number': number, 'worker': rank, 'device': str(device)}
        try:
            final = trainer.objective(trial, epochs=epochs, seed=seed + number, curriculum=curriculum)
            rec.update(status='complete', params=trial.params, reports=trial.reports,
                       metrics={k: v for k, v in final.items() if k != 'training_history'},
                       last_epoch=final['training_history'][-1] if final['training_history'] else None,
                       cells=len(final['training_history']) * max(trainer.mouse.shape[0], trainer.human.shape[0]))
        except TrialPruned as e:
            rec.update(status='pruned', reason=str(e), params=trial.params)
        except Exception as e:        # a failed trial is recorded and the study goes on (objective's except branch)
            rec.update(status='failed', reason=f"{type(e).__name__}: {e}", params=trial.params)
        if out_dir:
            with open(os.path.join(out_dir, f"trial_{number}_results.json"), "w") as f:
                json.dump(rec, f, indent=1, default=str)
        result_queue.put(rec)


def run_trials(make_trainer, n_trials, out_dir=None, gpus=None, workers_per_gpu=1, seed=42, epochs=None,
               curriculum=None):
    """Run `n_trials` independent trials on `gpus` devices, `workers_per_gpu` processes each.  `make_trainer(device)`
    (a picklable top-level function) builds the CurriculumTrainer inside each worker.  No process talks to
    another: trial numbers come from one queue, results go back on another.  Returns the records by trial number."""
    import torch.multiprocessing as mp
    gpus = torch.cuda.device_count() if gpus is None else int(gpus)
    if gpus < 1:
        raise RuntimeError("run_trials needs at least one HIP device")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    ctx = mp.get_context("spawn")
    trial_queue, result_queue = ctx.Queue(), ctx.Queue()
    n_workers = gpus * int(workers_per_gpu)
    for t in range(n_trials):
        trial_queue.put(t)
    for _ in range(n_workers):
        trial_queue.put(None)
    procs = [ctx.Process(target=_trial_worker, args=(r, r % gpus, make_trainer, trial_queue, result_queue, out_dir,
                                                     seed, epochs, curriculum)) for r in range(n_workers)]
    for p in procs:
        p.start()
    records = {}
    while len(records) < n_trials:
        try:
            rec = result_queue.get(timeout=5.0)
            records[rec['trial_number']] = rec
        except Exception:
            if not any(p.is_alive() for p in procs) and result_queue.empty():
                break
    for p in procs:
        p.join()
    return [records.get(t) for t in range(n_trials)]
