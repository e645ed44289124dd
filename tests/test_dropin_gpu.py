"""The reference's Python API on the HIP kernels, against the golden fixtures
captured from the imported reference (oracle/make_golden.py).

  module level   - GeneExpressionGenerator / GeneExpressionDiscriminator /
                   compute_gradient_penalty called like the reference's own
                   modules, with the recorded noise replayed; outputs, BatchNorm
                   buffers and parameter gradients against the reference's
  function level - the drop-in train_enhanced_cycle_gan against the metrics the
                   unmodified reference function returned on the same seeds

Float32 tolerances: outputs 2e-4 of the largest value; gradients 1e-3 of the
largest gradient of the network (the reference's own float32 noise against the
float64-exact restatement is 2e-4, see make_golden.py); per-epoch metrics 3e-3 in
the first epoch and 3e-2 later (the reference and the CPU oracle are 7e-3 apart
by the third epoch of the 'counts' scenario).
"""
import os
import tempfile

import numpy as np
import pytest
import torch

from tests.helpers import rel

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build_modules(gm, gh, cfg, seed):
    from cycle_gan_rna_amd import GAN_functions as F
    torch.manual_seed(seed)      # same construction order as GAN_main.py:125-128
    return (F.GeneExpressionGenerator(gm, gh, [12], cfg), F.GeneExpressionGenerator(gh, gm, [12], cfg),
            F.GeneExpressionDiscriminator(gm, [12], cfg), F.GeneExpressionDiscriminator(gh, [12], cfg))


def sampled(t, spec):
    return t.detach().cpu().reshape(-1).double()[::spec["stride"]].float()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_module_level_against_reference(tag):
    from cycle_gan_rna_amd.noise import RecordedNoise
    from oracle import cyclegan_oracle as O
    from cycle_gan_rna_amd import GAN_functions as F
    fx = torch.load(os.path.join(GOLD, "module_level.pt"), weights_only=True)[tag]
    cfg = O.default_config(hidden_dims=[96, 64, 32], preprocess_type=fx["cfg_preprocess_type"])
    gm, gh = fx["g_mouse"], fx["g_human"]
    g_m2h, g_h2m, d_m, d_h = build_modules(gm, gh, cfg, fx["init_seed"])
    dev = torch.device("cuda:0")
    xm, lm, lh = fx["xm"].to(dev), fx["lm"].to(dev), fx["lh"].to(dev)
    # ---- generator, train mode ----
    g = fx["gen"]
    g_m2h._noise_source = RecordedNoise(g["noise"], dev)
    out, mu, lv = g_m2h(xm, lm)
    assert rel(out, g["out"]) <= 2e-4 and rel(mu, g["mu"]) <= 2e-4 and rel(lv, g["log_var"]) <= 2e-4
    loss = (out * fx["w_out"].to(dev)).sum() + (mu * fx["w_mu"].to(dev)).sum() + (lv * fx["w_lv"].to(dev)).sum()
    loss.backward()
    floor = 1e-3 * max(float(v.abs().max()) for v in g["grads"].values())
    params = dict(g_m2h.named_parameters())
    for k, ref in g["grads"].items():
        assert rel(params[k].grad, ref, floor) <= 1.0, (k, rel(params[k].grad, ref, floor))
    bufs = dict(g_m2h.named_buffers())
    for k, ref in g["buffers"].items():
        assert rel(bufs[k].double(), ref.double()) <= 2e-4, k
    # ---- generator, eval mode (BatchNorm running stats, no attention dropout, eps still drawn) ----
    ge = fx["gen_eval"]
    g_m2h.load_state_dict({k: v.to(dev) for k, v in ge["state"].items()})
    g_m2h.eval()
    g_m2h._noise_source = RecordedNoise(ge["noise"], dev)
    with torch.no_grad():
        oe, _, _ = g_m2h(xm, lm)
    assert rel(oe, ge["out"]) <= 2e-4
    # ---- critic: real pass, fake pass, gradient penalty, full critic loss ----
    d = fx["disc"]
    d_m._noise_source = RecordedNoise(d["noise"], dev)
    fake = d["fake"].to(dev).requires_grad_(True)
    val, feat = d_m(xm, lm, return_features=True)
    val_f, _ = d_m(fake.detach(), lh, return_features=True)
    gp = F.compute_gradient_penalty(d_m, xm, fake, dev, lh)
    assert rel(val, d["validity"]) <= 2e-4 and rel(feat, d["features"]) <= 2e-4
    assert rel(val_f, d["validity_fake"]) <= 2e-4 and rel(gp, d["gp"]) <= 2e-4
    d_loss = -val.mean() + val_f.mean() + 13 * gp
    assert rel(d_loss, d["d_loss"]) <= 2e-4
    d_loss.backward()
    assert rel(fake.grad, d["grad_fake"], 1e-3 * float(d["grad_fake"].abs().max())) <= 1.0
    pd = dict(d_m.named_parameters())
    floor = 1e-3 * max(float(s["values"].abs().max()) for s in d["grads"].values())
    for k, spec in d["grads"].items():
        assert rel(sampled(pd[k].grad, spec), spec["values"], floor) <= 1.0, k
        # whole-tensor checksum; tensors whose gradient is identically zero hold rounding noise in the reference
        assert abs(float(pd[k].grad.double().sum()) - spec["sum"]) <= 2e-3 * max(spec["l2"], floor * spec["numel"] ** 0.5), k
    bd = dict(d_m.named_buffers())
    for k, ref in d["buffers"].items():
        assert rel(bd[k].double(), ref.double()) <= 2e-4, k


def test_double_differentiation_through_a_module_is_refused():
    """The modules' hand-written backward passes are first-order only.  The reference's own penalty code,
    torch.autograd.grad(d(x), x, create_graph=True) followed by .backward() (GAN_functions.py:199-203), must not run
    through them with the second-order terms silently dropped: it raises, and compute_gradient_penalty is the
    supported route (its gradients are checked against the reference in test_module_level_against_reference)."""
    from cycle_gan_rna_amd import GAN_functions as F
    from cycle_gan_rna_amd import cWGAN as W
    from oracle import cyclegan_oracle as O
    cfg = O.default_config(hidden_dims=[96, 64, 32], preprocess_type="log_normalized")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    d = F.GeneExpressionDiscriminator(40, [12], cfg).to(dev)
    g = F.GeneExpressionGenerator(40, 40, [12], cfg).to(dev)
    x = torch.rand(16, 40, device=dev, requires_grad=True)
    lab = torch.randint(0, 12, (16, 1), device=dev)
    for fn in (lambda: d(x, lab), lambda: g(x, lab)[0]):
        out = fn()
        with pytest.raises(RuntimeError, match="first-order only"):
            torch.autograd.grad(out, x, grad_outputs=torch.ones_like(out), create_graph=True)
    wd = W.Discriminator(40, [12], 0).to(dev)          # the notebook's own penalty (cWGAN_my.ipynb cell 3) likewise
    out = wd(x, lab, None)
    with pytest.raises(RuntimeError, match="first-order only"):
        torch.autograd.grad(out, x, grad_outputs=torch.ones_like(out), create_graph=True)
    gp = F.compute_gradient_penalty(d, x.detach(), torch.rand(16, 40, device=dev), dev, lab)
    gp.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in d.named_parameters() if "9.bias" not in n)


class Obs:
    def __init__(self, names):
        import pandas as pd
        self.obs = {"myannotations": pd.Series(list(names))}


class TwoCol:
    def transform(self, x):
        return np.asarray(x)[:, :2]


@pytest.mark.parametrize("tag", ["one_iter_epochs", "cadence_linear", "counts", "unequal_restart"])
def test_train_function_against_reference(tag):
    """unequal_restart: the species have 40 and 88 cells - the shorter loader is restarted and re-shuffled within
    every epoch and ragged batches are truncated to the smaller of the pair (GAN_functions.py:299-319)."""
    from sklearn.decomposition import PCA
    from sklearn.preprocessing import StandardScaler
    from cycle_gan_rna_amd import GAN_functions as F
    from oracle import cyclegan_oracle as O
    fx = torch.load(os.path.join(GOLD, "function_level_unequal.pt" if tag == "unequal_restart" else "function_level.pt"),
                    weights_only=True)[tag]
    cfg = dict(fx["cfg"], device="cuda", noise="host_replay", progress=False)
    gm, gh, N = fx["g_mouse"], fx["g_human"], fx["N"]
    nm, nh = (N, N) if isinstance(N, int) else N
    mods = build_modules(gm, gh, cfg, fx["init_seed"])
    xm, _ = O.synthetic_expression(nm, gm, 12, seed=fx["data_seed"])
    xh, _ = O.synthetic_expression(nh, gh, 12, seed=fx["data_seed"] + 1)
    lm, lh = O.balanced_labels(nm, 12, fx["data_seed"] + 2), O.balanced_labels(nh, 12, fx["data_seed"] + 3)
    names = [f"type{i:02d}" for i in range(12)]
    idx_to_ct = dict(enumerate(names))
    cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]
    scaler = StandardScaler().fit(xh.numpy())
    pca = PCA(n_components=4).fit(scaler.transform(xh.numpy()))
    coords = pca.transform(scaler.transform(xh.numpy()))[:, :2]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(fx["train_seed"])
            np.random.seed(0)
            metrics, out_dir = F.train_enhanced_cycle_gan(
                *mods, Obs([names[i] for i in lm.tolist()]), Obs([names[i] for i in lh.tolist()]), xm, xh, cat, coords,
                TwoCol(), idx_to_ct, cfg, torch.device("cuda:0"), pca_model=pca, scaler=scaler)
            files = set(os.listdir(out_dir))
        finally:
            os.chdir(cwd)
    e = cfg["epochs"]
    assert {"config.json", f"g_m2h_epoch_{e}.pt", f"g_h2m_epoch_{e}.pt", f"d_m_epoch_{e}.pt", f"d_h_epoch_{e}.pt",
            f"umap_epoch_{e}.png", f"centroid_correlations_epoch_{e}_scaled.png"} <= files
    assert set(metrics) == set(fx["metrics"])
    for k, ref in fx["metrics"].items():
        assert len(metrics[k]) == len(ref) == e
        for ep, (a, b) in enumerate(zip(metrics[k], ref)):
            tol = 3e-3 if ep == 0 else 3e-2
            floor = 0.5 if k in ("g_loss", "d_m_loss", "d_h_loss") else 0.0
            assert rel(a, b, floor) <= tol, (k, ep, a, b)
    # the modules passed in hold the trained weights afterwards (the reference trains in place)
    ref_final = fx["final"]["d_m"]["model.4.weight"]
    got = mods[2].state_dict()["model.4.weight"].cpu().reshape(-1).double()[::ref_final["stride"]]
    init = build_modules(gm, gh, cfg, fx["init_seed"])[2].state_dict()["model.4.weight"].reshape(-1).double()[::ref_final["stride"]]
    upd_ref = ref_final["values"].double() - init
    assert float((got - ref_final["values"].double()).norm() / upd_ref.norm()) <= 0.35


@pytest.mark.parametrize("identity,iters_per_epoch", [(False, 1), (True, 1), (False, 7), (False, (3.25, 6.25))])
def test_train_function_default_mode_replays_graphs(identity, iters_per_epoch):
    """Default mode of the drop-in loop (noise drawn in HBM): the step runs as HIP graphs on several streams.  With
    the same seed, a run that launches every kernel eagerly on one stream gives the same metrics bit for bit.
    identity: the same with the identity-loss extension switched on (one gene list for both species)."""
    from cycle_gan_rna_amd import GAN_functions as F
    from oracle import cyclegan_oracle as O
    fx = torch.load(os.path.join(GOLD, "function_level.pt"), weights_only=True)["one_iter_epochs"]
    gm, gh, N = fx["g_mouse"], fx["g_human"], fx["N"]
    # several iterations per epoch: generator forwards are issued ahead, across batches; a pair of factors: the two
    # species differ in size, so one loader restarts within the epoch and the last batches are ragged
    fm_, fh_ = iters_per_epoch if isinstance(iters_per_epoch, tuple) else (iters_per_epoch, iters_per_epoch)
    nm, nh = int(N * fm_), int(N * fh_)
    if identity:
        gh = gm
    xm, _ = O.synthetic_expression(nm, gm, 12, seed=3)
    xh, _ = O.synthetic_expression(nh, gh, 12, seed=4)
    lm, lh = O.balanced_labels(nm, 12, 5), O.balanced_labels(nh, 12, 6)
    names = [f"type{i:02d}" for i in range(12)]
    cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]
    res = {}
    cwd = os.getcwd()
    for mode in ("fast", "plain"):
        # the plain run also reads every generator step's metrics at once instead of one generator step late
        cfg = dict(fx["cfg"], device="cuda", progress=False, epochs=4, hip_graphs=mode == "fast",
                   concurrent_streams=mode == "fast", deferred_metrics=mode == "fast")
        if identity:
            cfg["lambda_identity"] = 0.5
        mods = build_modules(gm, gh, cfg, fx["init_seed"])
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                torch.manual_seed(11)
                np.random.seed(0)
                metrics, _ = F.train_enhanced_cycle_gan(
                    *mods, Obs([names[i] for i in lm.tolist()]), Obs([names[i] for i in lh.tolist()]), xm, xh, cat,
                    xh.numpy()[:, :2], TwoCol(), dict(enumerate(names)), cfg, torch.device("cuda:0"))
            finally:
                os.chdir(cwd)
        assert all(np.isfinite(v).all() for v in metrics.values())
        assert ("identity_loss" in metrics) == identity and (not identity or min(metrics["identity_loss"]) > 0)
        res[mode] = (metrics, mods[0].state_dict()["encoder.0.weight"].cpu())
    differ = {k: [(i, float(a), float(b)) for i, (a, b) in enumerate(zip(res["fast"][0][k], res["plain"][0][k])) if a != b][:2]
              for k in res["fast"][0] if res["fast"][0][k] != res["plain"][0][k]}
    assert not differ, f"first (epoch, graphs, eager) that differ: {differ}"
    assert torch.equal(res["fast"][1], res["plain"][1])


def _oracle_state(module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


@pytest.mark.parametrize("sparse,with_labels,gm,gh,latent,heads", [
    (False, False, 37, 50, 256, 4), (True, True, 64, 41, 256, 4), (True, True, 45, 41, 128, 8), (False, True, 40, 40, 512, 4)])
def test_modules_against_oracle_edge_configurations(sparse, with_labels, gm, gh, latent, heads):
    """cat_covs=None (zero embeddings, GAN_functions.py:136-137,182-183), the sparse-attention class
    (extra unused `gate` parameters), and gene counts that are odd / not multiples of 4 (unaligned
    weight column slices take the scalar-load path of the GEMM), other latent widths and head counts."""
    from cycle_gan_rna_amd import GAN_functions as F
    from cycle_gan_rna_amd.noise import HostReplayNoise
    from oracle import cyclegan_oracle as O
    cfg = O.default_config(hidden_dims=[96, 64, 32], preprocess_type="log_normalized", use_sparse_attention=sparse,
                           latent_dim=latent, num_attention_heads=heads)
    g_m2h, g_h2m, d_m, d_h = build_modules(gm, gh, cfg, 5)
    if sparse:
        assert "residual_attention_blocks.0.1.gate.weight" in g_m2h.state_dict()
    dev = torch.device("cuda:0")
    B = 12
    xm, _ = O.synthetic_expression(B, gm, 12, seed=1)
    lab = O.balanced_labels(B, 12, 2)[:, None] if with_labels else None
    dlab = None if lab is None else lab.to(dev)
    Pg, Pd = _oracle_state(g_m2h), _oracle_state(d_h)
    for P in (Pg, Pd):
        for k in O.param_names(P):
            P[k].requires_grad_(True)
    # oracle
    torch.manual_seed(3)
    nz = O.GlobalRngNoise()
    out, mu, lv = O.generator_forward(Pg, xm, lab, cfg, nz, train=True)
    val, feat = O.discriminator_forward(Pd, out, lab, nz)
    (val.mean() + 0.1 * feat.pow(2).mean() + 0.01 * (mu * lv).mean()).backward()
    # drop-in modules on the GPU, same draws replayed from the host generator
    torch.manual_seed(3)
    src = HostReplayNoise(dev)
    g_m2h._noise_source = src
    d_h._noise_source = src
    o2, mu2, lv2 = g_m2h(xm.to(dev), dlab)
    v2, f2 = d_h(o2, dlab, return_features=True)
    (v2.mean() + 0.1 * f2.pow(2).mean() + 0.01 * (mu2 * lv2).mean()).backward()
    assert rel(o2, out) <= 2e-4 and rel(v2, val) <= 2e-4 and rel(f2, feat) <= 2e-4
    for mod, P in ((g_m2h, Pg), (d_h, Pd)):
        params = dict(mod.named_parameters())
        floor = 1e-3 * max(float(P[k].grad.abs().max()) for k in O.param_names(P) if P[k].grad is not None)
        for k in O.param_names(P):
            if P[k].grad is None:            # unused parameters (gate; embeddings without labels)
                assert params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, k
                continue
            assert rel(params[k].grad, P[k].grad, floor) <= 1.0, (k, rel(params[k].grad, P[k].grad, floor))


def test_modules_with_two_covariates_against_oracle():
    """vocab_sizes with two entries: two embedding tables concatenated into the decoder / critic input
    (GAN_functions.py:85-86,:135,:160-161,:182), gradients for both tables, and the gradient penalty."""
    from cycle_gan_rna_amd import GAN_functions as F
    from cycle_gan_rna_amd.noise import HostReplayNoise
    from oracle import cyclegan_oracle as O
    cfg = O.default_config(hidden_dims=[96, 64, 32], preprocess_type="log_normalized")
    gm, gh, B = 44, 52, 12
    torch.manual_seed(5)
    gen = F.GeneExpressionGenerator(gm, gh, [12, 5], cfg)
    dis = F.GeneExpressionDiscriminator(gh, [12, 5], cfg)
    dev = torch.device("cuda:0")
    xm, _ = O.synthetic_expression(B, gm, 12, seed=1)
    xh, _ = O.synthetic_expression(B, gh, 12, seed=2)
    g = torch.Generator().manual_seed(3)
    lab = torch.stack([O.balanced_labels(B, 12, 2), torch.randint(0, 5, (B,), generator=g)], 1)
    Pg, Pd = _oracle_state(gen), _oracle_state(dis)
    for P in (Pg, Pd):
        for k in O.param_names(P):
            P[k].requires_grad_(True)
    torch.manual_seed(3)
    nz = O.GlobalRngNoise()
    out, mu, lv = O.generator_forward(Pg, xm, lab, cfg, nz, train=True)
    val, feat = O.discriminator_forward(Pd, out, lab, nz)
    gp = O.gradient_penalty(Pd, xh, out.detach(), lab, nz)
    (val.mean() + 0.1 * feat.pow(2).mean() + 0.01 * (mu * lv).mean() + 2.0 * gp).backward()
    torch.manual_seed(3)
    src = HostReplayNoise(dev)
    gen._noise_source = src
    dis._noise_source = src
    o2, mu2, lv2 = gen(xm.to(dev), lab.to(dev))
    v2, f2 = dis(o2, lab.to(dev), return_features=True)
    gp2 = F.compute_gradient_penalty(dis, xh.to(dev), o2.detach(), dev, lab.to(dev))
    (v2.mean() + 0.1 * f2.pow(2).mean() + 0.01 * (mu2 * lv2).mean() + 2.0 * gp2).backward()
    assert rel(o2, out) <= 2e-4 and rel(v2, val) <= 2e-4 and rel(gp2, gp) <= 2e-4
    for mod, P in ((gen, Pg), (dis, Pd)):
        params = dict(mod.named_parameters())
        floor = 1e-3 * max(float(P[k].grad.abs().max()) for k in O.param_names(P) if P[k].grad is not None)
        for k in ("embeddings.0.weight", "embeddings.1.weight", "decoder.0.weight" if mod is gen else "model.0.weight"):
            assert rel(params[k].grad, P[k].grad, floor) <= 1.0, (k, rel(params[k].grad, P[k].grad, floor))


@pytest.mark.parametrize("E,heads", [(16, 4), (8, 2), (32, 8), (16, 16), (4, 1)])
def test_standalone_blocks_match_torch_modules(E, heads):
    """ResidualBlock and GeneSelfAttention used on their own (eval mode: no dropout) against the
    torch.nn layers they wrap, forward and backward."""
    from cycle_gan_rna_amd import GAN_functions as F
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(6, 16, E)
    rb = F.ResidualBlock(E).to(dev)
    xr = x.to(dev).requires_grad_(True)
    y = rb(xr)                                     # train mode: batch statistics
    y.square().sum().backward()
    xc = x.clone().requires_grad_(True)
    blk = torch.nn.Sequential(torch.nn.Linear(E, E), torch.nn.BatchNorm1d(E), torch.nn.ReLU(), torch.nn.Linear(E, E),
                              torch.nn.BatchNorm1d(E))
    sd = {k: v.cpu() for k, v in rb.block.state_dict().items()}
    sd["1.running_mean"].zero_(), sd["1.running_var"].fill_(1), sd["4.running_mean"].zero_(), sd["4.running_var"].fill_(1)
    blk.load_state_dict(sd)
    ye = xc + blk(xc.view(96, E)).view(6, 16, E)
    ye.square().sum().backward()
    assert rel(y, ye) <= 2e-4 and rel(xr.grad, xc.grad, 1e-3 * float(xc.grad.abs().max())) <= 1.0
    assert rel(rb.block[0].weight.grad, blk[0].weight.grad, 1e-3 * float(blk[0].weight.grad.abs().max())) <= 1.0
    at = F.GeneSelfAttention(E, heads).to(dev).eval()
    xa = x.to(dev).requires_grad_(True)
    ya = at(xa)
    ya.square().sum().backward()
    ref = torch.nn.MultiheadAttention(E, heads, dropout=0.1, batch_first=True).eval()
    ref.load_state_dict({k: v.cpu() for k, v in at.attention.state_dict().items()})
    ln = torch.nn.LayerNorm(E)
    ln.load_state_dict({k: v.cpu() for k, v in at.norm.state_dict().items()})
    xb = x.clone().requires_grad_(True)
    yb = ln(xb + ref(xb, xb, xb)[0])
    yb.square().sum().backward()
    assert rel(ya, yb) <= 2e-4 and rel(xa.grad, xb.grad, 1e-3 * float(xb.grad.abs().max())) <= 1.0
    assert rel(at.attention.in_proj_weight.grad, ref.in_proj_weight.grad, 1e-3 * float(ref.in_proj_weight.grad.abs().max())) <= 1.0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_hundred_iteration_trajectory_against_reference(precision):
    """BASELINE configuration 1 (2000 genes, batch 128, default architecture): 100 iterations (25 epochs of 4) of the
    drop-in loop with the reference's noise replayed, against the per-epoch metrics the unmodified reference
    returned on the same seeds.  Two float32 CPU implementations of this step (the reference and the oracle) are
    up to 2e-2 apart on these metrics over the 100 iterations (tests/golden/trajectory_cfg1.pt holds both), because
    Adam amplifies rounding differences step by step.  Bounds, every epoch:
      float32   5e-2 on every metric (1e-3 in the first epoch)
      bf16 MFMA 2e-2 on the smooth terms (cycle, cell-type, KL, output variances: the stated bf16 tolerance) and on
                everything in the first epoch; 2e-1 on the adversarial quantities (critic losses, feature matching and
                g_loss, which contains them): differences of critic outputs near zero that follow the critics'
                own noisy trajectory."""
    from cycle_gan_rna_amd import GAN_functions as F
    from oracle import cyclegan_oracle as O
    fx = torch.load(os.path.join(GOLD, "trajectory_cfg1.pt"), weights_only=True)
    cfg = dict(fx["cfg"], device="cuda", noise="host_replay", progress=False, precision=precision)
    gm, gh, N = fx["g_mouse"], fx["g_human"], fx["N"]
    mods = build_modules(gm, gh, cfg, fx["init_seed"])
    xm, _ = O.synthetic_expression(N, gm, 12, seed=fx["data_seed"])
    xh, _ = O.synthetic_expression(N, gh, 12, seed=fx["data_seed"] + 1)
    lm, lh = O.balanced_labels(N, 12, fx["data_seed"] + 2), O.balanced_labels(N, 12, fx["data_seed"] + 3)
    names = [f"type{i:02d}" for i in range(12)]
    cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(fx["train_seed"])
            np.random.seed(0)
            metrics, _ = F.train_enhanced_cycle_gan(
                *mods, Obs([names[i] for i in lm.tolist()]), Obs([names[i] for i in lh.tolist()]), xm, xh, cat, None,
                None, dict(enumerate(names)), cfg, torch.device("cuda:0"))
        finally:
            os.chdir(cwd)
    adversarial = ("g_loss", "d_m_loss", "d_h_loss", "fm_loss")
    first, worst = {}, {}
    for k, ref in fx["metrics"].items():
        assert len(metrics[k]) == len(ref) == 25
        floor = 0.5 if k in ("g_loss", "d_m_loss", "d_h_loss") else 0.0
        errs = [rel(a, b, floor) for a, b in zip(metrics[k], ref)]
        first[k], worst[k] = errs[0], max(errs)
    print(precision, {k: "%.1e" % v for k, v in worst.items()})
    if precision == "fp32":
        # The arbiter: the oracle run in float64 on the same noise.  Over 100 iterations every float32 implementation
        # drifts from it by amplified rounding - the reference itself by up to 3.7e-2, the float32 CPU oracle by up
        # to 5.4e-2 (fixture: metrics, oracle_metrics, oracle_metrics_f64).  The device path must stay as close to the
        # float64 trajectory as those two do, within a factor of two, metric by metric.
        f64 = fx["oracle_metrics_f64"]
        dev64 = {}
        for k, ref in fx["metrics"].items():
            floor = 0.5 if k in ("g_loss", "d_m_loss", "d_h_loss") else 0.0
            e_dev = max(rel(a, b, floor) for a, b in zip(metrics[k], f64[k]))
            e_cpu = max(max(rel(a, b, floor) for a, b in zip(ref, f64[k])),
                        max(rel(a, b, floor) for a, b in zip(fx["oracle_metrics"][k], f64[k])))
            dev64[k] = (e_dev, e_cpu)
        print("vs float64 (device, worse of the two CPU float32 runs):", {k: "%.1e / %.1e" % v for k, v in dev64.items()})
        for k, (e_dev, e_cpu) in dev64.items():
            assert e_dev <= max(1e-3, 2.0 * e_cpu), (k, e_dev, e_cpu)
    for k in worst:
        if precision == "fp32":
            # smooth terms stay within 2e-2 over the 100 iterations; the adversarial ones (differences of critic
            # outputs, and the feature-matching MSE of the critics' means, ~1e-3 in size) amplify rounding more:
            # two float32 summation orders of this very build differ by up to 5e-2 there
            assert first[k] <= 1e-3 and worst[k] <= (1e-1 if k in adversarial else 2e-2), (k, first[k], worst[k])
        else:
            assert first[k] <= 2e-2 and worst[k] <= (2e-1 if k in adversarial else 2e-2), (k, first[k], worst[k])
