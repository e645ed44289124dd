"""Conditional WGAN-GP of cWGAN_my.ipynb without a GPU.

  * the CPU oracle (oracle/cwgan_oracle.py) against the golden vectors captured by executing the notebook's own
    cells (oracle/make_golden_cwgan.py): module level with the recorded noise replayed, function level through
    the loop scaffolding on the recorded seeds;
  * the engine's hand-derived forward/backward/update sequence (cwgan_engine.py) driven with the per-kernel CPU
    restatements in float64 against the autograd oracle, which pins the derivation; on the GPU box the same
    engine runs on the HIP kernels (tests/test_cwgan_gpu.py).
"""
import os

import pytest
import torch

from oracle import cwgan_oracle as W
from oracle import cyclegan_oracle as O
from oracle import init_params
from oracle.ops_cpu import CpuOps
from tests.helpers import rel

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cwgan.pt")
CFG = dict(lr=1e-4, latent_dim=64, nb_critic=2, lambda_gp=10, library_size=10000)


def cwgan_problem(x_dim, vocab, batch, z_dim=64, seed=3, dtype=torch.float32):
    torch.manual_seed(seed)
    g = init_params.cwgan_generator_state(x_dim, vocab, 0, z_dim)
    d = init_params.cwgan_discriminator_state(x_dim, vocab, 0)
    cast = lambda sd: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}  # noqa: E731
    x, _ = O.synthetic_expression(batch, x_dim, 12, seed=seed + 1)
    gen = torch.Generator().manual_seed(seed + 2)
    cat = torch.stack([torch.randint(0, v, (batch,), generator=gen) for v in vocab], dim=1)
    return cast(g), cast(d), x.to(dtype), cat, torch.zeros(batch, 0, dtype=dtype)


class ZFloat32Noise:
    """The notebook draws z with torch.randn(batch, latent_dim) in float32 whatever the model's dtype."""

    def __init__(self, dtype):
        self.dtype = dtype

    def randn(self, shape):
        return torch.randn(shape, dtype=torch.float32).to(self.dtype)

    def keep(self, shape, p_drop):
        return torch.empty(shape, dtype=self.dtype).bernoulli_(1.0 - p_drop).div_(1.0 - p_drop)

    def rand(self, shape):
        return torch.rand(shape).to(self.dtype)


def test_oracle_module_level_golden():
    fx = torch.load(GOLD, weights_only=True)["module"]
    torch.manual_seed(fx["init_seed"])
    g = init_params.cwgan_generator_state(fx["x_dim"], fx["vocab"], 0, fx["z_dim"])
    d = init_params.cwgan_discriminator_state(fx["x_dim"], fx["vocab"], 0)
    s = fx["g_weight_sample"]
    assert torch.equal(g["network.9.weight"].reshape(-1)[::s["stride"]], s["values"])      # same initial draw
    x, cat, z = fx["x"], fx["cat"], fx["z"]
    num = torch.zeros(x.shape[0], 0)
    tr = W.CwganOracleTrainer(g, d, CFG, noise=O.RecordedNoise(fx["noise"]))
    fake = W.generator_forward(tr.G, z, cat, num, 10000, True)
    assert rel(fake, fx["fake"]) <= 1e-5
    for k, ref in fx["g_buffers_after"].items():
        assert rel(tr.G[k], ref) <= 1e-5, k
    nz = tr.noise
    real_v = W.discriminator_forward(tr.D, x, cat, num, nz)
    fake_v = W.discriminator_forward(tr.D, fx["fake"], cat, num, nz)
    gp = W.gradient_penalty(tr.D, x, fx["fake"], cat, num, nz)
    assert rel(real_v, fx["real_v"]) <= 1e-5 and rel(fake_v, fx["fake_v"]) <= 1e-5 and rel(gp, fx["gp"]) <= 1e-4
    (-real_v.mean() + fake_v.mean() + 10 * gp).backward()
    floor = 1e-3 * max(float(sp["values"].abs().max()) for sp in fx["d_grads"].values())
    for k, sp in fx["d_grads"].items():
        got = tr.D[k].grad.reshape(-1)[::sp["stride"]]
        assert rel(got, sp["values"], floor) <= 0.5, k         # i.e. 5e-4 of the largest gradient


def test_oracle_function_level_golden():
    """train_gan of the notebook (3 epochs, 6 batches, nb_critic 5, negative-penalty ramp) against the oracle's
    loop on the same seeds: the per-epoch metrics the notebook logs."""
    fx = torch.load(GOLD, weights_only=True)["function"]
    cfg = dict(fx["cfg"])
    x, _ = O.synthetic_expression(fx["N"], fx["x_dim"], 12, seed=fx["data_seed"])
    gen = torch.Generator().manual_seed(fx["cat_seed"])
    cat = torch.stack([torch.randint(0, v, (fx["N"],), generator=gen) for v in fx["vocab"]], dim=1)
    torch.manual_seed(fx["init_seed"])
    g = init_params.cwgan_generator_state(fx["x_dim"], fx["vocab"], 0, cfg["latent_dim"])
    d = init_params.cwgan_discriminator_state(fx["x_dim"], fx["vocab"], 0)
    torch.manual_seed(fx["train_seed"])
    tr = W.CwganOracleTrainer(g, d, cfg)
    log = W.train(tr, x, cat, torch.zeros(fx["N"], 0))
    assert len(log) == len(fx["metrics"]) == cfg["epochs"]
    for ep, (got, ref) in enumerate(zip(log, fx["metrics"])):
        for k in ("d_loss", "g_loss", "g_loss_main", "negative_proportion", "negative_magnitude",
                  "negative_penalty_weight"):
            assert rel(got[k], ref[k], 0.5) <= (1e-4 if ep == 0 else 2e-3), (ep, k, got[k], ref[k])
    for k, sp in fx["final_d"].items():
        got = tr.D[k].detach().reshape(-1)[::sp["stride"]]
        assert rel(got, sp["values"]) <= 2e-2, k


def run_engine(dtype, x_dim, vocab, batch, iters, cfg=CFG):
    from cycle_gan_rna_amd.cwgan_engine import CwganEngine
    g, d, x, cat, num = cwgan_problem(x_dim, vocab, batch, dtype=dtype)
    tr = W.CwganOracleTrainer(g, d, cfg)
    torch.manual_seed(11)
    exp = [tr.iteration(x, cat, num) for _ in range(iters)]
    eng = CwganEngine(g, d, cfg, CpuOps(dtype), batch)
    noise = ZFloat32Noise(dtype)
    torch.manual_seed(11)
    got = []
    for _ in range(iters):
        eng.iteration(x, cat, num, noise)
        got.append(dict(eng.losses(), fake=eng.fake.clone()))
    return tr, eng, exp, got


@pytest.mark.parametrize("x_dim,vocab", [(48, [3, 12]), (37, [5])])
def test_engine_explicit_backward_equals_autograd_in_float64(x_dim, vocab):
    tr, eng, exp, got = run_engine(torch.float64, x_dim, vocab, 16, iters=3)
    for e, g in zip(exp, got):
        for k in ("d_loss", "gp", "g_loss", "g_loss_main"):
            assert rel(g[k], e[k]) <= 5e-7, (k, g[k], e[k])     # losses are rounded to float32 like .item()
        assert rel(g["fake"], e["fake"]) <= 1e-9
    for name, ref in (("g", tr.G), ("d", tr.D)):
        sd = eng.state_dict(name)
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(ref[k]), k
            elif k in ("network.0.bias", "network.3.bias", "network.6.bias"):
                continue          # a bias feeding a BatchNorm: zero gradient, random walk on rounding noise
            else:
                assert rel(v, ref[k]) <= 1e-7, (name, k, rel(v, ref[k]))


def test_engine_float32_first_iteration():
    tr, eng, exp, got = run_engine(torch.float32, 48, [3, 12], 16, iters=1)
    for k in ("d_loss", "gp", "g_loss"):
        assert rel(got[0][k], exp[0][k], 0.5) <= 1e-4, (k, got[0][k], exp[0][k])
    assert rel(got[0]["fake"], exp[0]["fake"]) <= 1e-4
