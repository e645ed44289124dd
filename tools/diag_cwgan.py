"""Which execution mode of the cWGAN engine departs from the plain run, and in which tensors."""
import torch
from cycle_gan_rna_amd.cwgan_engine import CwganEngine
from cycle_gan_rna_amd.noise import DeviceNoise
from cycle_gan_rna_amd.ops import HipOps
from tests.test_cwgan_host import CFG, cwgan_problem

ops = HipOps("cuda:0")
g, d, x, cat, num = cwgan_problem(64, [3, 12], 32)
dx, dcat, dnum = x.cuda(), cat.cuda(), num.cuda()
base = None
for graphs, streams in ((False, False), (False, False), (True, False), (False, True), (True, True)):
    for iters in (1, 2, 4):
        eng = CwganEngine(g, d, CFG, ops, 32)
        if streams:
            eng.enable_concurrency()
        if graphs:
            eng.enable_graphs()
        noise = DeviceNoise(ops.device, seed=7, default_generator=True)
        for _ in range(iters):
            noise.begin_step()
            eng.iteration(dx, dcat, dnum, noise)
        torch.cuda.synchronize()
        sd = {("d", k): v for k, v in eng.state_dict("d").items()}
        sd.update({("g", k): v for k, v in eng.state_dict("g").items()})
        if not graphs and not streams and iters not in (base or {}):
            base = base or {}
            base[iters] = sd
            continue
        bad = [(k, float((v.float() - base[iters][k].float()).abs().max())) for k, v in sd.items()
               if not torch.equal(v, base[iters][k])]
        print("graphs", graphs, "streams", streams, "iters", iters, "differing:", bad[:6], len(bad), flush=True)
