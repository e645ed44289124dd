"""Generate the golden fixtures under tests/golden/ from the imported reference.

Runs ONLY in the build container (needs /root/reference, which never travels).
The reference is imported unmodified with six absent third-party modules
(anndata, scanpy[.experimental[.pp]], umap, seaborn) registered as empty stubs;
its classes and functions are then called through their public API:

  module level   GeneExpressionGenerator / GeneExpressionDiscriminator forward,
                 compute_gradient_penalty, and parameter gradients
  function level the unmodified train_enhanced_cycle_gan on synthetic data

Every fixture records torch.__version__, the seeds, and (for module level) the
noise tensors in draw order, so that the fixtures stay valid on another torch
build.  The same run cross-checks oracle/cyclegan_oracle.py against the
reference and prints the largest deviations.

    python oracle/make_golden.py [--only fp64|module|func|func_unequal|traj]
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def import_reference():
    for name in ['anndata', 'scanpy', 'scanpy.experimental', 'scanpy.experimental.pp', 'umap', 'seaborn']:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['scanpy.experimental.pp'].normalize_pearson_residuals = lambda *a, **k: None
    sys.modules['scanpy'].experimental = sys.modules['scanpy.experimental']
    sys.modules['scanpy.experimental'].pp = sys.modules['scanpy.experimental.pp']
    sys.modules['seaborn'].scatterplot = lambda *a, **k: None
    sys.modules['seaborn'].histplot = lambda *a, **k: None
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, '/root/reference')
    import GAN_functions as R
    R.has_wandb = False
    return R


from oracle import cyclegan_oracle as O          # noqa: E402
from oracle import init_params                   # noqa: E402
from oracle.train_loop import oracle_train       # noqa: E402


def relerr(a, b, floor=0.0):
    """max|a-b| / max(max|b|, floor).  ``floor`` keeps tensors that are zero up to
    rounding (e.g. the gradient of a bias that feeds a BatchNorm) from reading as
    O(1) relative error."""
    a = torch.as_tensor(a).detach().to(torch.float64)
    b = torch.as_tensor(b).detach().to(torch.float64)
    return float((a - b).abs().max() / max(float(b.abs().max()), floor, 1e-30))


def grad_floor(params):
    return 1e-3 * max(float(p.grad.abs().max()) for p in params if p.grad is not None)


def sample(t, n=2048):
    """Fixed strided sample + sum + L2 of a tensor (keeps fixtures small)."""
    f = t.detach().reshape(-1).double()
    stride = max(1, f.numel() // n)
    return dict(numel=f.numel(), stride=stride, values=f[::stride].float().clone(),
                sum=float(f.sum()), l2=float(f.norm()))


def small_cfg(**over):
    return O.default_config(hidden_dims=[96, 64, 32], use_wandb=False, **over)


def check_init(R, cfg, gm, gh, seed):
    torch.manual_seed(seed)
    ref = [R.GeneExpressionGenerator(gm, gh, [12], cfg), R.GeneExpressionGenerator(gh, gm, [12], cfg),
           R.GeneExpressionDiscriminator(gm, [12], cfg), R.GeneExpressionDiscriminator(gh, [12], cfg)]
    nets = init_params.build_nets(gm, gh, [12], cfg, seed)
    for m, (name, sd) in zip(ref, nets.items()):
        rsd = m.state_dict()
        assert list(rsd.keys()) == list(sd.keys()), name
        for k in rsd:
            assert torch.equal(rsd[k], sd[k]), (name, k)
    return ref, nets


# --------------------------------------------------------------------------- #
def module_level(R):
    out = {}
    for tag, gm, gh, B, ptype in [("a", 48, 40, 8, "log_normalized"), ("b", 200, 200, 32, "counts")]:
        cfg = small_cfg(preprocess_type=ptype)
        ref, nets = check_init(R, cfg, gm, gh, seed=7)
        g_m2h, g_h2m, d_m, d_h = ref
        xm, lm = O.synthetic_expression(B, gm, 12, seed=11)
        xh, lh = O.synthetic_expression(B, gh, 12, seed=12)
        lm, lh = lm[:, None], lh[:, None]
        gen = torch.Generator().manual_seed(5)
        w_out = torch.randn(B, gh, generator=gen)
        w_mu = torch.randn(B, 256, generator=gen)
        w_lv = torch.randn(B, 256, generator=gen)
        fx = dict(torch_version=str(torch.__version__), cfg_preprocess_type=ptype, g_mouse=gm, g_human=gh, B=B,
                  init_seed=7, noise_seed=21, xm=xm, xh=xh, lm=lm, lh=lh, w_out=w_out, w_mu=w_mu, w_lv=w_lv)

        # ---- generator forward + backward (reference) ----
        torch.manual_seed(21)
        o, mu, lv = g_m2h(xm, lm)
        loss = (o * w_out).sum() + (mu * w_mu).sum() + (lv * w_lv).sum()
        g_m2h.zero_grad()
        loss.backward()
        # ---- same with the oracle, recording the noise ----
        P = {k: v.clone() for k, v in nets['g_m2h'].items()}
        for k in O.param_names(P):
            P[k].requires_grad_(True)
        torch.manual_seed(21)
        rec = O.RecordingNoise()
        o2, mu2, lv2 = O.generator_forward(P, xm, lm, cfg, rec, train=True)
        loss2 = (o2 * w_out).sum() + (mu2 * w_mu).sum() + (lv2 * w_lv).sum()
        loss2.backward()
        errs = [relerr(o2, o), relerr(mu2, mu), relerr(lv2, lv)]
        rp = dict(g_m2h.named_parameters())
        fl = grad_floor(rp.values())
        for k in O.param_names(P):
            errs.append(relerr(P[k].grad, rp[k].grad, fl))
        rb = dict(g_m2h.named_buffers())
        for k in rb:
            errs.append(relerr(P[k].double(), rb[k].double()))
        print(f"[module {tag}] generator oracle-vs-reference max rel err {max(errs):.3e}")
        fx['gen'] = dict(noise=[t.clone() for t in rec.tensors], out=o.detach(), mu=mu.detach(), log_var=lv.detach(),
                         grads={k: rp[k].grad.clone() for k in rp},
                         buffers={k: v.clone() for k, v in rb.items()})

        # ---- generator eval-mode forward ----
        g_m2h.eval()
        torch.manual_seed(22)
        with torch.no_grad():
            oe, mue, lve = g_m2h(xm, lm)
        g_m2h.train()
        Pe = {k: v.clone() for k, v in g_m2h.state_dict().items()}
        torch.manual_seed(22)
        rec = O.RecordingNoise()
        with torch.no_grad():
            oe2, _, _ = O.generator_forward(Pe, xm, lm, cfg, rec, train=False)
        print(f"[module {tag}] generator eval oracle-vs-reference {relerr(oe2, oe):.3e}")
        fx['gen_eval'] = dict(noise=[t.clone() for t in rec.tensors], out=oe, mu=mue, log_var=lve,
                              state={k: v.clone() for k, v in Pe.items()})

        # ---- discriminator: real pass, GP, full critic loss (reference) ----
        fake = o.detach()[:, :gm] if gh >= gm else torch.cat([o.detach(), o.detach()[:, :gm - gh]], 1)
        fake = fake.clone().requires_grad_(True)
        torch.manual_seed(23)
        d_m.zero_grad()
        val, feat = d_m(xm, lm, return_features=True)
        val_f, _ = d_m(fake.detach(), lh, return_features=True)
        gp = R.compute_gradient_penalty(d_m, xm, fake, torch.device('cpu'), lh)
        d_loss = -val.mean() + val_f.mean() + 13 * gp
        d_loss.backward()
        Pd = {k: v.clone() for k, v in nets['d_m'].items()}
        for k in O.param_names(Pd):
            Pd[k].requires_grad_(True)
        fake2 = fake.detach().clone().requires_grad_(True)
        torch.manual_seed(23)
        rec = O.RecordingNoise()
        v2, f2 = O.discriminator_forward(Pd, xm, lm, rec)
        vf2, _ = O.discriminator_forward(Pd, fake2.detach(), lh, rec)
        gp2 = O.gradient_penalty(Pd, xm, fake2, lh, rec)
        (-v2.mean() + vf2.mean() + 13 * gp2).backward()
        rpd = dict(d_m.named_parameters())
        errs = [relerr(v2, val), relerr(f2, feat), relerr(gp2, gp), relerr(fake2.grad, fake.grad)]
        fl = grad_floor(rpd.values())
        for k in O.param_names(Pd):
            if rpd[k].grad is not None:
                errs.append(relerr(Pd[k].grad, rpd[k].grad, fl))
        rbd = dict(d_m.named_buffers())
        for k in rbd:
            errs.append(relerr(Pd[k].double(), rbd[k].double()))
        print(f"[module {tag}] critic+GP oracle-vs-reference max rel err {max(errs):.3e}  gp={float(gp):.6f}")
        fx['disc'] = dict(noise=[t.clone() for t in rec.tensors], fake=fake.detach(), validity=val.detach(),
                          features=feat.detach(), validity_fake=val_f.detach(), gp=float(gp), d_loss=float(d_loss),
                          grad_fake=fake.grad.clone(),
                          grads={k: sample(rpd[k].grad) for k in rpd if rpd[k].grad is not None},
                          buffers={k: v.clone() for k, v in rbd.items()})
        out[tag] = fx
    torch.save(out, os.path.join(GOLD, "module_level.pt"))


# --------------------------------------------------------------------------- #
class Obs:
    """Stand-in for AnnData: only .obs['myannotations'] is read
    (GAN_functions.py:209-210, :481-485)."""

    def __init__(self, names):
        import pandas as pd
        self.obs = {'myannotations': pd.Series(list(names))}


class TwoCol:
    def transform(self, x):
        return np.asarray(x)[:, :2]


def run_reference_function(R, cfg, gm, gh, N, init_seed, data_seed, train_seed):
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import StandardScaler
    ref, nets = check_init(R, cfg, gm, gh, init_seed)
    g_m2h, g_h2m, d_m, d_h = ref
    n_mouse, n_human = (N, N) if isinstance(N, int) else N       # the two species may have different cell counts
    xm, lm = O.synthetic_expression(n_mouse, gm, 12, seed=data_seed)
    xh, lh = O.synthetic_expression(n_human, gh, 12, seed=data_seed + 1)
    lm, lh = O.balanced_labels(n_mouse, 12, data_seed + 2), O.balanced_labels(n_human, 12, data_seed + 3)
    names = [f"type{i:02d}" for i in range(12)]
    idx_to_ct = {i: n for i, n in enumerate(names)}
    mouse_ad, human_ad = Obs([names[i] for i in lm.tolist()]), Obs([names[i] for i in lh.tolist()])
    cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]
    scaler = StandardScaler().fit(xh.numpy())
    pca = PCA(n_components=4).fit(scaler.transform(xh.numpy()))
    coords = pca.transform(scaler.transform(xh.numpy()))[:, :2]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(train_seed)
            np.random.seed(0)
            metrics, _ = R.train_enhanced_cycle_gan(g_m2h, g_h2m, d_m, d_h, mouse_ad, human_ad, xm, xh, cat, coords,
                                                    TwoCol(), idx_to_ct, cfg, torch.device('cpu'),
                                                    pca_model=pca, scaler=scaler)
        finally:
            os.chdir(cwd)
    return ref, nets, (xm, lm, xh, lh), metrics


def run_oracle_function(nets, cfg, data, train_seed):
    xm, lm, xh, lh = data
    tr = O.OracleTrainer(nets, cfg, O.centroid_table(xm, lm, 12), O.centroid_table(xh, lh, 12))
    torch.manual_seed(train_seed)
    metrics = oracle_train(tr, xm, xh, lm[:, None], lh[:, None], cfg)
    return tr, metrics


def compare_function(tag, ref, tr, metrics, metrics2):
    errs = {k: relerr(metrics2[k], [float(v) for v in metrics[k]]) for k in metrics}
    perr, gauge_err = 0.0, 0.0
    for m, name in zip(ref, ['g_m2h', 'g_h2m', 'd_m', 'd_h']):
        for k, v in m.state_dict().items():
            e = relerr(tr.nets[name][k].detach().double(), v.double())
            if O.is_gauge_param(k, m.state_dict()):
                gauge_err = max(gauge_err, float((tr.nets[name][k].detach().double() - v.double()).abs().max()))
            else:
                perr = max(perr, e)
    print(f"[func {tag}] oracle-vs-reference metrics max rel err {max(errs.values()):.3e}; final params {perr:.3e}"
          f" (gauge params, abs: {gauge_err:.3e})")
    return errs


def function_level(R):
    out = {}
    scen = [
        ("one_iter_epochs", dict(epochs=6, batch_size=32, nb_critic=4, preprocess_type='log_normalized',
                                 decay_start_epoch=3, decay_type='cosine'), 200, 200, 32),
        ("cadence_linear", dict(epochs=3, batch_size=16, nb_critic=2, preprocess_type='log_normalized',
                                decay_start_epoch=1, decay_type='linear'), 64, 56, 72),
        ("counts", dict(epochs=3, batch_size=16, nb_critic=1, preprocess_type='counts', use_lr_decay=False), 64, 64, 40),
    ]
    for tag, over, gm, gh, N in scen:
        cfg = small_cfg(**over)
        ref, nets, data, metrics = run_reference_function(R, cfg, gm, gh, N, init_seed=3, data_seed=100, train_seed=9)
        tr, metrics2 = run_oracle_function(nets, cfg, data, train_seed=9)
        compare_function(tag, ref, tr, metrics, metrics2)
        fx = dict(torch_version=str(torch.__version__), cfg={k: v for k, v in cfg.items() if k != 'device'},
                  g_mouse=gm, g_human=gh, N=N, init_seed=3, data_seed=100, train_seed=9,
                  metrics={k: [float(v) for v in vals] for k, vals in metrics.items()},
                  final={name: {k: sample(v) for k, v in m.state_dict().items()}
                         for m, name in zip(ref, ['g_m2h', 'g_h2m', 'd_m', 'd_h'])})
        out[tag] = fx
    torch.save(out, os.path.join(GOLD, "function_level.pt"))


def function_level_unequal(R):
    """Different cell counts for the two species: the shorter loader runs out in the middle of an epoch, is
    restarted and re-shuffled (GAN_functions.py:299-309), and ragged last batches of either loader are truncated
    to the smaller of the pair (:315-319).  40 mouse cells = batches of 16, 16, 8; 88 human cells = 16 x 5, 8;
    six iterations per epoch, the mouse loader restarting at the fourth."""
    cfg = small_cfg(epochs=3, batch_size=16, nb_critic=2, preprocess_type='log_normalized', decay_start_epoch=1,
                    decay_type='cosine')
    gm, gh, N = 64, 56, (40, 88)
    ref, nets, data, metrics = run_reference_function(R, cfg, gm, gh, N, init_seed=3, data_seed=200, train_seed=13)
    tr, metrics2 = run_oracle_function(nets, cfg, data, train_seed=13)
    compare_function("unequal_restart", ref, tr, metrics, metrics2)
    fx = dict(torch_version=str(torch.__version__), cfg={k: v for k, v in cfg.items() if k != 'device'},
              g_mouse=gm, g_human=gh, N=list(N), init_seed=3, data_seed=200, train_seed=13,
              metrics={k: [float(v) for v in vals] for k, vals in metrics.items()},
              final={name: {k: sample(v) for k, v in m.state_dict().items()}
                     for m, name in zip(ref, ['g_m2h', 'g_h2m', 'd_m', 'd_h'])})
    torch.save({"unequal_restart": fx}, os.path.join(GOLD, "function_level_unequal.pt"))


def trajectory_level(R):
    """BASELINE configuration 1 shape: G=2000, B=128, default hidden dims, 100
    iterations (25 epochs of 4 iterations, nb_critic=4)."""
    cfg = O.default_config(epochs=25, batch_size=128, preprocess_type='log_normalized', use_wandb=False)
    ref, nets, data, metrics = run_reference_function(R, cfg, 2000, 2000, 512, init_seed=0, data_seed=1234, train_seed=1)
    tr, metrics2 = run_oracle_function(nets, cfg, data, train_seed=1)
    compare_function("traj", ref, tr, metrics, metrics2)
    # the same run of the oracle in float64 on the same (float32-drawn) noise: the arbiter between float32
    # implementations, whose trajectories drift apart by amplified rounding
    cast = lambda t: t.double() if t.is_floating_point() else t  # noqa: E731
    xm, lm, xh, lh = data
    tr64 = O.OracleTrainer({n: {k: cast(v) for k, v in sd.items()} for n, sd in nets.items()}, cfg,
                           O.centroid_table(xm, lm, 12).double(), O.centroid_table(xh, lh, 12).double(),
                           noise=O.WideNoise())
    torch.manual_seed(1)
    metrics64 = oracle_train(tr64, xm.double(), xh.double(), lm[:, None], lh[:, None], cfg)
    e_ref = {k: relerr(metrics64[k], [float(v) for v in metrics[k]]) for k in metrics}
    e_orc = {k: relerr(metrics64[k], metrics2[k]) for k in metrics}
    print("[traj] against the float64 oracle: reference max rel err %.3e, float32 oracle %.3e"
          % (max(e_ref.values()), max(e_orc.values())))
    fx = dict(torch_version=str(torch.__version__), cfg={k: v for k, v in cfg.items() if k != 'device'},
              oracle_metrics_f64={k: [float(v) for v in vals] for k, vals in metrics64.items()},
              g_mouse=2000, g_human=2000, N=512, init_seed=0, data_seed=1234, train_seed=1,
              metrics={k: [float(v) for v in vals] for k, vals in metrics.items()},
              # the float32 oracle on the same seeds: the gap between two CPU float32
              # implementations of the same mathematics is the rounding-noise floor
              # that any third implementation has to be judged against
              oracle_metrics={k: [float(v) for v in vals] for k, vals in metrics2.items()})
    torch.save(fx, os.path.join(GOLD, "trajectory_cfg1.pt"))


def verify_fp64(R):
    """The restatement against the reference with both in float64: agreement to
    ~1e-11 shows the two are the same mathematics, so the ~1e-4 float32
    deviations printed elsewhere are rounding, not a modelling difference."""
    for ptype in ['log_normalized', 'counts']:
        cfg = small_cfg(preprocess_type=ptype)
        ref, nets = check_init(R, cfg, 48, 40, 7)
        g, d = ref[0].double(), ref[2].double()
        xm, lm = O.synthetic_expression(8, 48, 12, seed=11)
        xh, lh = O.synthetic_expression(8, 40, 12, seed=12)
        xm, lm, lh = xm.double(), lm[:, None], lh[:, None]

        def to64(sd):
            P = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
            for k in O.param_names(P):
                P[k].requires_grad_(True)
            return P
        torch.manual_seed(21)
        o, mu, lv = g(xm, lm)
        (o.sum() + (mu * mu).sum() + lv.exp().sum()).backward()
        P = to64(nets['g_m2h'])
        torch.manual_seed(21)
        o2, mu2, lv2 = O.generator_forward(P, xm, lm, cfg, O.GlobalRngNoise(), True)
        (o2.sum() + (mu2 * mu2).sum() + lv2.exp().sum()).backward()
        rp = dict(g.named_parameters())
        fl = grad_floor(rp.values())
        errs = [relerr(o2, o), relerr(mu2, mu), relerr(lv2, lv)] + [relerr(P[k].grad, rp[k].grad, fl) for k in O.param_names(P)]
        fake = torch.randn(8, 48, dtype=torch.float64).requires_grad_(True)
        torch.manual_seed(23)
        val, _ = d(xm, lm, return_features=True)
        vf, _ = d(fake.detach(), lh, return_features=True)
        gp = R.compute_gradient_penalty(d, xm, fake, torch.device('cpu'), lh)
        (-val.mean() + vf.mean() + 13 * gp).backward()
        Pd = to64(nets['d_m'])
        fake2 = fake.detach().clone().requires_grad_(True)
        torch.manual_seed(23)
        nz = O.GlobalRngNoise()
        v2, _ = O.discriminator_forward(Pd, xm, lm, nz)
        vf2, _ = O.discriminator_forward(Pd, fake2.detach(), lh, nz)
        gp2 = O.gradient_penalty(Pd, xm, fake2, lh, nz)
        (-v2.mean() + vf2.mean() + 13 * gp2).backward()
        rpd = dict(d.named_parameters())
        fl = grad_floor(rpd.values())
        errs += [relerr(v2, val), relerr(gp2, gp), relerr(fake2.grad, fake.grad)]
        errs += [relerr(Pd[k].grad, rpd[k].grad, fl) for k in O.param_names(Pd) if rpd[k].grad is not None]
        print(f"[fp64 {ptype}] oracle-vs-reference max rel err {max(errs):.3e}")
        assert max(errs) < 1e-9


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    R = import_reference()
    torch.set_num_threads(8)
    if a.only in (None, "fp64"):
        verify_fp64(R)
    if a.only in (None, "module"):
        module_level(R)
    if a.only in (None, "func"):
        function_level(R)
    if a.only in (None, "func_unequal"):
        function_level_unequal(R)
    if a.only in (None, "traj"):
        trajectory_level(R)
