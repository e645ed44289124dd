"""Is one generator forward pass a pure function of its inputs?  Runs GenPass.forward over and over on one stream
with the same batch, labels, parameters and generator seed and compares every workspace buffer with the first run's;
prints the first buffer (in the order the pass writes them) that differs.  Then the same for the critics' part of a
critic step (stacked forward, both backward chains; gradients compared).

    python tools/diag_repeat_forward.py [repetitions] [genes] [batch]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cycle_gan_rna_amd import synthetic as S  # noqa: E402
from cycle_gan_rna_amd.engine import StepEngine  # noqa: E402
from cycle_gan_rna_amd.noise import DeviceNoise  # noqa: E402
from cycle_gan_rna_amd.ops import HipOps  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)
cfg = S.default_config(preprocess_type="log_normalized", batch_size=B, hidden_dims=[96, 64, 32], latent_dim=256,
                       num_attention_heads=4, residual_blocks=4, lambda_gp=13)
nets = S.random_networks(G, G, [12], cfg, seed=0)
xm, lm = S.synthetic_expression(B, G, 12, seed=1, device=dev)
xh, lh = S.synthetic_expression(B, G, 12, seed=2, device=dev)
cm, ch = S.centroid_table(xm, lm, 12), S.centroid_table(xh, lh, 12)
ops = HipOps(dev)
eng = StepEngine(nets, cfg, cm, ch, ops, B)
lm, lh = lm[:, None].contiguous(), lh[:, None].contiguous()
noise = DeviceNoise(dev, seed=1, default_generator=True)


def gen_buffers(p):
    out = [("enc_h0", p.enc_h0), ("enc_a", p.enc_a), ("ml", [p.ml]), ("z", p.z), ("t1", p.t1), ("r", p.r), ("t2", p.t2),
           ("y", p.y), ("dec_in", [p.dec_in]), ("dec_h0", p.dec_h0), ("dec_a", p.dec_a), ("out_raw", [p.out_raw])]
    return [(f"{n}[{i}]", t) for n, ts in out for i, t in enumerate(ts)]


def one_forward(p, x, lab):
    torch.cuda.manual_seed(5)
    noise.begin_step()
    p.forward(x, lab, noise)
    return [(n, t.clone()) for n, t in gen_buffers(p)]


def critics():
    torch.cuda.manual_seed(7)
    noise.begin_step()
    eng._d_state = (xm, xh, lm, lh, noise, fake_h, fake_m)
    eng._d_critics()
    return [("arena_d.g", eng.arena_d.g.clone()), ("d_tot", eng.d_tot.clone())] + \
           [(f"{k}.{n}", getattr(eng.dstack[k], n).clone()) for k in ("d_m", "d_h") for n in ("h0", "a1", "a2", "a3", "val")]


for name, p, x, lab in (("g_m2h", eng.gen_a["g_m2h"], xm, lm), ("g_h2m", eng.gen_a["g_h2m"], xh, lh)):
    ref = one_forward(p, x, lab)
    bad = 0
    for rep in range(reps):
        cur = one_forward(p, x, lab)
        diff = [n for (n, a), (_, b) in zip(cur, ref) if not torch.equal(a, b)]
        if diff:
            bad += 1
            a, b = dict(cur)[diff[0]], dict(ref)[diff[0]]
            w = (a != b).nonzero()
            print(f"{name} rep {rep}: first differing buffer {diff[0]} ({len(w)} elements, first at {w[0].tolist()}, "
                  f"{a[tuple(w[0])].item()!r} vs {b[tuple(w[0])].item()!r}); all: {diff}", flush=True)
    print(f"{name}: {bad} of {reps} forward passes deviate", flush=True)
fake_h, fake_m = eng.gen_a["g_m2h"].out[:B].clone(), eng.gen_a["g_h2m"].out[:B].clone()
eng._d_scales_b = None
lam = float(cfg["lambda_gp"])
eng.d_scales.copy_(torch.tensor([-1.0 / B, 1.0 / B, lam / B] * 2, dtype=torch.float64))
ref = critics()
bad = 0
for rep in range(reps):
    cur = critics()
    diff = [n for (n, a), (_, b) in zip(cur, ref) if not torch.equal(a, b)]
    if diff:
        bad += 1
        print(f"critics rep {rep}: differing {diff}", flush=True)
print(f"critics: {bad} of {reps} deviate", flush=True)
