"""Do two HIP graphs replayed on two streams overlap?  Times the critic part and the generator forward of a critic
step alone and side by side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cycle_gan_rna_amd.engine import StepEngine
from cycle_gan_rna_amd.noise import DeviceNoise
from cycle_gan_rna_amd.ops import HipOps
from cycle_gan_rna_amd import synthetic as S

dev = torch.device("cuda", 0)
G, B = 5000, 512
cfg = S.default_config(preprocess_type="log_normalized", batch_size=B)
nets = S.random_networks(G, G, [12], cfg, seed=0)
xm, lm = S.synthetic_expression(4 * B, G, 12, seed=1, device=dev)
xh, lh = S.synthetic_expression(4 * B, G, 12, seed=2, device=dev)
cm, ch = S.centroid_table(xm, lm, 12), S.centroid_table(xh, lh, 12)
ops = HipOps(dev)
eng = StepEngine(nets, cfg, cm, ch, ops, B)
lm, lh = lm[:, None].contiguous(), lh[:, None].contiguous()
noise = DeviceNoise(dev, seed=1, default_generator=True)
eng.enable_concurrency(); eng.enable_lookahead(); eng.enable_graphs()
batch = lambda i: (xm[(i % 4) * B:(i % 4 + 1) * B], xh[(i % 4) * B:(i % 4 + 1) * B], lm[(i % 4) * B:(i % 4 + 1) * B], lh[(i % 4) * B:(i % 4 + 1) * B])
for i in range(12):
    noise.begin_step()
    eng.iteration(i, *batch(i), noise, nxt=batch(i + 1), metrics=False)
torch.cuda.synchronize()
gs = eng._graphs
print({k: v["graph"] is not None for k, v in gs.items()})
main = torch.cuda.current_stream()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for k in sorted(gs):
    if gs[k]["graph"] is not None:
        print("%-8s %.3f ms" % (k, timed(lambda k=k: gs[k]["graph"].replay())))
