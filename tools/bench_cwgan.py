"""Throughput of the conditional WGAN-GP step of cWGAN_my.ipynb (cells 1-3) on the HIP engine: one iteration =
nb_critic critic updates + one generator update on one batch (the notebook's defaults: nb_critic 5, latent 64)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from cycle_gan_rna_amd.cwgan_engine import CwganEngine
from cycle_gan_rna_amd.noise import DeviceNoise
from cycle_gan_rna_amd.ops import HipOps
from cycle_gan_rna_amd import synthetic as S
from cycle_gan_rna_amd.cWGAN import Discriminator, Generator

ap = argparse.ArgumentParser()
ap.add_argument("--genes", type=int, default=5000)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--eager", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = dict(latent_dim=64, lr=1e-4, beta1=0.5, beta2=0.9, eps=1e-8, nb_critic=5, lambda_gp=10, library_size=10000)
torch.manual_seed(0)
g, d = Generator(a.genes, [3, 12], 0, 64), Discriminator(a.genes, [3, 12], 0)
x, _ = S.synthetic_expression(4 * a.batch, a.genes, 12, seed=1, device=dev)
cat = torch.stack([torch.randint(0, v, (a.batch,)) for v in (3, 12)], dim=1).to(dev)
ops = HipOps(dev)
eng = CwganEngine(g.state_dict(), d.state_dict(), cfg, ops, a.batch)
noise = DeviceNoise(dev, seed=1, default_generator=True)
if not a.eager:
    eng.enable_concurrency()
    eng.enable_graphs()


def it(i):
    noise.begin_step()
    eng.iteration(x[(i % 4) * a.batch:(i % 4 + 1) * a.batch], cat, None, noise)


for i in range(6):
    it(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    it(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"cWGAN genes={a.genes} batch={a.batch}: {dt * 1e3:.3f} ms per iteration (5 critic + 1 generator update), "
      f"{a.batch / dt:.0f} cells/s, losses {eng.losses()}")
