"""Diagnostic: per-tensor relative error of the gradients of one critic step + one generator step, HIP engine against
the CPU oracle (float32 and float64), at a given size.  usage: diag_grad_error.py GENES [BATCH] [precision]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cyclegan_oracle as O          # noqa: E402
from tests.helpers import make_problem           # noqa: E402


def oracle_grads(nets, cfg, cm, ch, xm, xh, lm, lh, dtype):
    cast = lambda t: t.to(dtype) if t.is_floating_point() else t  # noqa: E731
    tr = O.OracleTrainer({n: {k: cast(v) for k, v in sd.items()} for n, sd in nets.items()}, cfg, cast(cm), cast(ch))
    torch.manual_seed(11)
    if dtype == torch.float64:      # same noise values as the float32 run, widened
        class N64(O.GlobalRngNoise):
            def randn(self, shape, dtype):
                return torch.randn(shape, dtype=torch.float32).double()

            def keep(self, shape, p, dtype):
                return super().keep(shape, p, torch.float32).double()
        tr.noise = N64()
    tr.d_step(cast(xm), cast(xh), lm, lh, step=False)
    gd = {(n, k): tr.nets[n][k].grad.clone() for n in ("d_m", "d_h") for k in O.param_names(tr.nets[n])
          if tr.nets[n][k].grad is not None}
    tr.opt_dm.step(); tr.opt_dh.step()
    L = tr.g_losses(cast(xm), cast(xh), lm, lh)
    tr.g_scales({k: L[k].item() for k in ("g_adv", "g_cycle", "g_celltype", "g_fm", "kl")})
    tr.g_finish(L, step=False)
    gg = {(n, k): tr.nets[n][k].grad.clone() for n in ("g_m2h", "g_h2m") for k in O.param_names(tr.nets[n])
          if tr.nets[n][k].grad is not None}
    return gd, gg


def main():
    G = int(sys.argv[1])
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    from cycle_gan_rna_amd.engine import StepEngine
    from cycle_gan_rna_amd.noise import HostReplayNoise
    from cycle_gan_rna_amd.ops import HipOps
    cfg = O.default_config(preprocess_type="log_normalized", nb_critic=4)
    nets, xm, xh, lm, lh = make_problem(G, G, B, cfg)
    cm, ch = O.centroid_table(xm, lm, 12), O.centroid_table(xh, lh, 12)
    lm2, lh2 = lm[:, None], lh[:, None]
    g32 = oracle_grads(nets, cfg, cm, ch, xm, xh, lm2, lh2, torch.float32)
    g64 = oracle_grads(nets, cfg, cm, ch, xm, xh, lm2, lh2, torch.float64)
    eng = StepEngine(nets, cfg, cm, ch, HipOps("cuda:0", precision=prec), B)
    grabbed = {}
    real_adam = eng._adam

    def adam(arena, lo=0, hi=None):      # the optimizer step is issued range by range
        which = "d" if arena is eng.arena_d else "g"
        hi = arena.n if hi is None else hi
        grabbed.setdefault(which, torch.full_like(arena.g, float("nan")))[lo:hi] = arena.g[lo:hi]
        real_adam(arena, lo, hi)
    eng._adam = adam
    torch.manual_seed(11)
    eng.iteration(0, xm.cuda(), xh.cuda(), lm2.cuda(), lh2.cuda(), HostReplayNoise("cuda:0"))
    torch.cuda.synchronize()
    print(f"{'tensor':58s} {'hip-vs-f64':>11s} {'f32-vs-f64':>11s}   (relative L2 of the gradient)")
    for which, arena, (a32, a64) in (("d", eng.arena_d, (g32[0], g64[0])), ("g", eng.arena_g, (g32[1], g64[1]))):
        flat = grabbed[which].cpu()
        tot = [0.0, 0.0, 0.0]
        for (n, k), (off, cnt, shape) in arena.offsets.items():
            if (n, k) in a64 and not O.is_gauge_param(k, nets[n]):
                ref = a64[(n, k)].reshape(-1)
                tot[0] += float((flat[off:off + cnt].double() - ref).norm()) ** 2
                tot[1] += float((a32[(n, k)].reshape(-1).double() - ref).norm()) ** 2
                tot[2] += float(ref.norm()) ** 2
        print(f"SUMMARY {which}-step gradients, all tensors: hip-vs-f64 {(tot[0] / tot[2]) ** 0.5:.3e}   "
              f"f32-vs-f64 {(tot[1] / tot[2]) ** 0.5:.3e}")
        for (n, k), (off, cnt, shape) in arena.offsets.items():
            if (n, k) not in a64 or O.is_gauge_param(k, nets[n]):
                continue
            ref = a64[(n, k)].reshape(-1)
            den = float(ref.norm())
            if den == 0:
                continue
            e_hip = float((flat[off:off + cnt].double() - ref).norm()) / den
            e_32 = float((a32[(n, k)].reshape(-1).double() - ref).norm()) / den
            print(f"{n + '.' + k:58s} {e_hip:11.3e} {e_32:11.3e}")


if __name__ == "__main__":
    main()
