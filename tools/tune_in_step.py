"""Tune the (tile, split-K) table of cg_gemm inside the training step.

tools/sweep_gemm.py times one product replayed over and over, with its operands resident in L2; in the step the
operands of a product come from HBM and the choice that wins there can be a different one.  Here the step itself runs
(one stream, every launch bracketed by HIP events, as bench.py's roofline pass does) and, shape by shape, every
candidate is forced through CGRNA_GEMM_OVERRIDE and timed where it is used.  Prints the table rows of the winners.

    python tools/tune_in_step.py [bf16] [--thresh=0.97] [--genes=5000] [--batch=512] > gpurun_out/in_step_f32.csv
    python tools/tune_in_step.py --dump --shapes=512,5000,1024,0,0,0:512,5000,512,0,1,0     (these shapes only, every candidate's time)
"""
import os
os.environ.setdefault("CGRNA_GEMM_TUNING", "1")   # cg_gemm looks at its tuning hooks only when this is set at its first call
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cycle_gan_rna_amd import synthetic as S  # noqa: E402
from cycle_gan_rna_amd.engine import StepEngine  # noqa: E402
from cycle_gan_rna_amd.noise import DeviceNoise  # noqa: E402
from cycle_gan_rna_amd.ops import HipOps  # noqa: E402

G, B, NB = 5000, 512, 4
for _a in sys.argv[1:]:
    if _a.startswith("--genes="):
        G = int(_a.split("=", 1)[1])
    if _a.startswith("--batch="):
        B = int(_a.split("=", 1)[1])
half = "bf16" in sys.argv[1:]
THRESH = 0.97          # a candidate must beat the current choice by 3 % (--thresh=0.98 for 2 %), twice
for _a in sys.argv[1:]:
    if _a.startswith("--thresh="):
        THRESH = float(_a.split("=", 1)[1])
DUMP = "--dump" in sys.argv[1:]         # print the time of every candidate, not only the winners
dev = torch.device("cuda", 0)
cfg = S.default_config(preprocess_type="log_normalized", batch_size=B)
nets = S.random_networks(G, G, [12], cfg, seed=0)
xm, lm = S.synthetic_expression(4 * B, G, 12, seed=1, device=dev)
xh, lh = S.synthetic_expression(4 * B, G, 12, seed=2, device=dev)
cm, ch = S.centroid_table(xm, lm, 12), S.centroid_table(xh, lh, 12)
ops = HipOps(dev, precision="bf16" if half else "fp32")
eng = StepEngine(nets, cfg, cm, ch, ops, B)
lm, lh = lm[:, None].contiguous(), lh[:, None].contiguous()
noise = DeviceNoise(dev, seed=1, default_generator=True)
filler_a = torch.rand(4096, 4096, device=dev)
filler_c = torch.empty(4096, 4096, device=dev)


def measure(reps=2):
    """Mean microseconds per launch of every GEMM shape over `reps` groups of NB iterations."""
    agg = {}
    for _ in range(reps):
        ops.begin_gemm_timing()
        for i in range(NB):
            for _ in range(24):
                ops.gemm_untimed(filler_c, filler_a, filler_a, False, True)
            s = (i % 4) * B
            noise.begin_step()
            eng.iteration(i, xm[s:s + B], xh[s:s + B], lm[s:s + B], lh[s:s + B], noise, metrics=False)
        torch.cuda.synchronize()
        ops.end_gemm_timing()
        for shape, ms, fl in ops.last_gemm_detail:
            c = agg.setdefault(shape, [0, 0.0])
            c[0] += 1
            c[1] += ms
    return {k: (v[1] / v[0] * 1e3, v[0] / reps / NB) for k, v in agg.items()}


measure(1)
base = measure(3)
want = 1 if half else 0
shapes = sorted((k for k in base if (k[6] & 1) == want and min(k[0], k[1]) > 2 and k[2] > 16), key=lambda k: -base[k][0] * base[k][1])
for _a in sys.argv[1:]:
    if _a.startswith("--shapes="):
        _only = {tuple(int(v) for v in t.split(",")) for t in _a.split("=", 1)[1].split(":")}
        shapes = [k for k in shapes if tuple(k[:6]) in _only]
total_gain = 0.0
for shape in shapes:
    M, N, K, ta, tb, acc, dt = shape
    cur, n_it = base[shape]
    best = (cur, None)
    for t in range(3):
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):
            os.environ["CGRNA_GEMM_OVERRIDE"] = f"{M},{N},{K},{ta},{tb},{acc},{t},{sp}"
            us = measure(1)[shape][0]
            if DUMP:
                print(f"# cand {M},{N},{K},{ta},{tb},{acc} tile {t} splits {sp}: {us:.2f} us", flush=True)
            if us < best[0]:
                best = (us, (t, sp))
    os.environ.pop("CGRNA_GEMM_OVERRIDE", None)
    if best[1] is not None and best[0] < THRESH * cur:          # confirm a candidate that looks faster by the threshold
        os.environ["CGRNA_GEMM_OVERRIDE"] = f"{M},{N},{K},{ta},{tb},{acc},{best[1][0]},{best[1][1]}"
        again = measure(3)[shape][0]
        os.environ.pop("CGRNA_GEMM_OVERRIDE", None)
        recheck = measure(3)[shape][0]
        if again < THRESH * recheck:
            total_gain += (recheck - again) * n_it
            print(f"{M},{N},{K},{ta},{tb},{acc},{best[1][0]},{best[1][1]},{again:.2f},{recheck:.2f},{n_it:.2f}", flush=True)
            continue
    print(f"# keep {M},{N},{K},{ta},{tb},{acc}: {cur:.2f} us (best other {best[0]:.2f})", flush=True)
print(f"# estimated gain {total_gain:.1f} us per iteration", flush=True)
