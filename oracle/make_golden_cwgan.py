"""Golden fixtures for the conditional WGAN-GP of cWGAN_my.ipynb, made by running the notebook's own cells.

Run in the build container (the reference does not travel):  python -m oracle.make_golden_cwgan

Cells 1-3 of /root/reference/cWGAN_my.ipynb (Generator, Discriminator, compute_gradient_penalty, train_gan) are
executed unmodified in a namespace that provides what cell 0 imports (torch, nn, optim, numpy) and a recording
stand-in for the absent `wandb` module.  The script
  1. checks oracle/cwgan_oracle.py == notebook in float64 (outputs, penalty, every gradient),
  2. writes tests/golden/cwgan.pt: a module-level case (inputs, recorded noise, outputs, gradients; float32) and a
     function-level case (train_gan on synthetic data: the per-epoch metrics it logs and checksums of the final
     weights).
Only data is written; no reference source is copied.
"""
import json
import os
import types

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import cwgan_oracle as W
from .cyclegan_oracle import RecordingNoise, synthetic_expression

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOTEBOOK = "/root/reference/cWGAN_my.ipynb"


class WandbStub(types.ModuleType):
    def __init__(self):
        super().__init__("wandb")
        self.run = object()
        self.logged = []

    def log(self, metrics):
        self.logged.append(dict(metrics))


def load_notebook():
    nb = json.load(open(NOTEBOOK))
    ns = dict(torch=torch, nn=nn, optim=optim, np=np, wandb=WandbStub(), print=lambda *a, **k: None)
    for ci in (1, 2, 3):
        exec(compile("".join(nb["cells"][ci]["source"]), f"cWGAN_my.ipynb:cell{ci}", "exec"), ns)
    return ns


def relerr(a, b, floor=0.0):
    return float((a - b).abs().max() / max(float(b.abs().max()), floor, 1e-300))


def problem(dtype, seed=0, x_dim=300, vocab=(3, 12), batch=24, z_dim=64):
    ns = load_notebook()
    torch.manual_seed(seed)
    gen = ns["Generator"](x_dim, list(vocab), 0, z_dim).to(dtype)
    disc = ns["Discriminator"](x_dim, list(vocab), 0).to(dtype)
    x, _ = synthetic_expression(batch, x_dim, 12, seed=seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    cat = torch.stack([torch.randint(0, v, (batch,), generator=g) for v in vocab], dim=1)
    num = torch.zeros(batch, 0, dtype=dtype)
    z = torch.randn(batch, z_dim, generator=g).to(dtype)
    return ns, gen, disc, x.to(dtype), cat, num, z


def verify_fp64():
    ns, gen, disc, x, cat, num, z = problem(torch.float64)
    gs = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    ds = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    # reference
    torch.manual_seed(7)
    fake = gen(z, cat, num)
    real_v = disc(x, cat, num)
    fake_v = disc(fake.detach(), cat, num)
    gp = ns["compute_gradient_penalty"](disc, x, fake.detach(), cat, num, "cpu")
    loss = -real_v.mean() + fake_v.mean() + 10 * gp
    loss.backward()
    ref_d = {k: p.grad.clone() for k, p in disc.named_parameters()}
    disc.zero_grad()
    gl = -disc(fake, cat, num).mean()
    gl.backward()
    ref_g = {k: p.grad.clone() for k, p in gen.named_parameters()}
    # oracle
    torch.manual_seed(7)
    tr = W.CwganOracleTrainer(gs, ds, dict(lr=1e-4, latent_dim=64, nb_critic=1, lambda_gp=10))
    nz = tr.noise
    ofake = W.generator_forward(tr.G, z, cat, num, 10000, True)
    oreal_v = W.discriminator_forward(tr.D, x, cat, num, nz)
    ofake_v = W.discriminator_forward(tr.D, ofake.detach(), cat, num, nz)
    ogp = W.gradient_penalty(tr.D, x, ofake.detach(), cat, num, nz)
    (-oreal_v.mean() + ofake_v.mean() + 10 * ogp).backward()
    errs = {"fake": relerr(ofake, fake), "real_v": relerr(oreal_v, real_v), "fake_v": relerr(ofake_v, fake_v),
            "gp": abs(float(ogp) - float(gp)) / abs(float(gp))}
    for k in ref_d:
        errs["dD " + k] = relerr(tr.D[k].grad, ref_d[k], floor=1e-12)
    for k in W.param_keys(tr.D):
        tr.D[k].grad = None
    (-W.discriminator_forward(tr.D, ofake, cat, num, nz).mean()).backward()
    for k in ref_g:
        if ref_g[k].abs().max() > 1e-9:
            errs["dG " + k] = relerr(tr.G[k].grad, ref_g[k])
    worst = max(errs.values())
    print("float64 oracle vs notebook: worst relative error %.2e" % worst)
    assert worst < 1e-9, errs
    return worst


def sample(v, n=256):
    """Strided sample of a tensor plus its norm: enough to pin it without storing it."""
    stride = max(1, v.numel() // n)
    return dict(stride=stride, values=v.detach().reshape(-1)[::stride].clone(), norm=float(v.detach().double().norm()))


def module_level():
    ns, gen, disc, x, cat, num, z = problem(torch.float32, seed=3)
    # the networks are rebuilt by the tests from the seed (torch.manual_seed(3); Generator(...); Discriminator(...))
    out = dict(x_dim=x.shape[1], vocab=[3, 12], z_dim=64, init_seed=3, x=x, cat=cat, z=z)
    tr = W.CwganOracleTrainer({k: v.clone() for k, v in gen.state_dict().items()},
                              {k: v.clone() for k, v in disc.state_dict().items()},
                              dict(lr=1e-4, latent_dim=64, nb_critic=1, lambda_gp=10), noise=RecordingNoise())
    torch.manual_seed(11)
    fake = gen(z, cat, num)
    torch.manual_seed(12)
    real_v = disc(x, cat, num)
    fake_v = disc(fake.detach(), cat, num)
    gp = ns["compute_gradient_penalty"](disc, x, fake.detach(), cat, num, "cpu")
    (-real_v.mean() + fake_v.mean() + 10 * gp).backward()
    # record the same noise through the oracle (same seed, same draw order)
    torch.manual_seed(12)
    nz = tr.noise
    W.discriminator_forward(tr.D, x, cat, num, nz)
    W.discriminator_forward(tr.D, fake.detach(), cat, num, nz)
    W.gradient_penalty(tr.D, x, fake.detach(), cat, num, nz)
    out.update(fake=fake.detach(), real_v=real_v.detach(), fake_v=fake_v.detach(), gp=float(gp),
               noise=[t.clone() for t in nz.tensors],
               d_grads={k: sample(p.grad) for k, p in disc.named_parameters()},
               g_weight_sample=sample(gen.state_dict()["network.9.weight"]),
               g_buffers_after={k: v.clone() for k, v in gen.state_dict().items() if "running" in k})
    return out


def function_level():
    ns = load_notebook()
    cfg = dict(epochs=3, latent_dim=64, batch_size=32, lr=1e-4, beta1=0.5, beta2=0.9, eps=1e-8, nb_critic=5, lambda_gp=10,
               negative_penalty_start_epoch=1, negative_penalty_ramp_epochs=2, max_negative_penalty=5.0,
               library_size=10000)
    x_dim, vocab, N = 400, [3, 12], 200
    x, _ = synthetic_expression(N, x_dim, 12, seed=21)
    g = torch.Generator().manual_seed(22)
    cat = torch.stack([torch.randint(0, v, (N,), generator=g) for v in vocab], dim=1).numpy()
    num = np.zeros((N, 0))
    torch.manual_seed(5)
    gen = ns["Generator"](x_dim, vocab, 0, 64)
    disc = ns["Discriminator"](x_dim, vocab, 0)
    g0 = {k: v.clone() for k, v in gen.state_dict().items()}
    d0 = {k: v.clone() for k, v in disc.state_dict().items()}
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x), batch_size=cfg["batch_size"], shuffle=True,
                                         drop_last=True)
    final = {}
    torch.manual_seed(99)
    ns["train_gan"](gen, disc, loader, cat, num, cfg, "cpu",
                    save_fn=lambda G, D, e: final.update(g={k: v.clone() for k, v in G.state_dict().items()},
                                                         d={k: v.clone() for k, v in D.state_dict().items()}))
    ref_log = ns["wandb"].logged
    # the oracle's loop on the same seed
    torch.manual_seed(99)
    tr = W.CwganOracleTrainer(g0, d0, cfg)
    olog = W.train(tr, x, torch.from_numpy(cat), torch.zeros(N, 0))
    worst = 0.0
    for r, o in zip(ref_log, olog):
        for k in ("d_loss", "g_loss", "g_loss_main"):
            worst = max(worst, abs(r[k] - o[k]) / max(abs(r[k]), 0.5))
    print("float32 oracle loop vs notebook train_gan: worst metric deviation %.2e over %d epochs" % (worst, len(olog)))
    assert len(ref_log) == len(olog) == cfg["epochs"] and worst < 1e-3, (ref_log, olog)

    def checksum(sd):
        return {k: sample(v, 64) for k, v in sd.items() if v.dtype.is_floating_point}
    return dict(cfg=cfg, x_dim=x_dim, vocab=vocab, N=N, data_seed=21, cat_seed=22, init_seed=5, train_seed=99,
                metrics=[{k: float(v) for k, v in m.items()} for m in ref_log],
                oracle_metrics=olog, final_g=checksum(final["g"]), final_d=checksum(final["d"]))


def main():
    worst = verify_fp64()
    out = dict(fp64_oracle_vs_reference=worst, module=module_level(), function=function_level(),
               torch_version=str(torch.__version__))
    path = os.path.join(ROOT, "tests", "golden", "cwgan.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
