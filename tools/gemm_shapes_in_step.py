"""Per-shape time of the products of the training step, measured where they run (the step on one stream, every
launch bracketed by HIP events): launches per iteration, microseconds per launch, TFLOP/s, share of the GEMM time.

    python tools/gemm_shapes_in_step.py [bf16] [--genes=5000] [--batch=512]
"""
import os
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
agg = {}
for rep in range(4):
    ops.begin_gemm_timing()
    for i in range(NB):
        s = (i % 4) * B
        noise.begin_step()
        eng.iteration(i, xm[s:s + B], xh[s:s + B], lm[s:s + B], lh[s:s + B], noise, metrics=False)
    torch.cuda.synchronize()
    ops.end_gemm_timing()
    if rep == 0:
        continue
    for shape, ms, fl in ops.last_gemm_detail:
        c = agg.setdefault(shape, [0, 0.0, fl])
        c[0] += 1
        c[1] += ms
total = sum(v[1] for v in agg.values())
print(f"GEMM time per iteration {total / 3 / NB:.3f} ms, launches per iteration {sum(v[0] for v in agg.values()) / 3 / NB:.1f}")
print("M,N,K,transA,transB,accumulate,dtype  launches/iter  us/launch  TFLOP/s  share  cumulative")
cum = 0.0
for shape, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    cum += ms
    print(f"{shape}  {n / 3 / NB:6.2f}  {ms / n * 1e3:8.1f}  {fl / (ms / n * 1e-3) / 1e12:7.1f}  {ms / total:6.3f}  {cum / total:6.3f}")
