"""What a parked HIP graph costs: engines with graphs + side streams are built, run and closed one after the other;
prints host RSS, device memory in use and torch's reserved memory after each.  Usage: python tools/diag_parked_cost.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cycle_gan_rna_amd import engine as E                            # noqa: E402
from cycle_gan_rna_amd.noise import DeviceNoise                      # noqa: E402
from cycle_gan_rna_amd.ops import HipOps                             # noqa: E402
from oracle import cyclegan_oracle as O                              # noqa: E402
from tests.helpers import make_problem, small_cfg                    # noqa: E402


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024


n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ops = HipOps("cuda:0")
cfg = small_cfg(preprocess_type="log_normalized", nb_critic=2)
nets, xm, xh, lm, lh = make_problem(48, 48, 16, cfg)
cm, ch = O.centroid_table(xm, lm, 12), O.centroid_table(xh, lh, 12)
dx = [t.cuda() for t in (xm, xh, lm[:, None], lh[:, None])]
for k in range(n):
    eng = E.StepEngine(nets, cfg, cm, ch, ops, 16)
    eng.enable_concurrency()
    eng.enable_lookahead()
    eng.enable_graphs()
    noise = DeviceNoise(ops.device, seed=7, default_generator=True)
    for i in range(12):
        noise.begin_step()
        eng.iteration(i, *dx, noise, nxt=dx)
    torch.cuda.synchronize()
    made = len(eng._graphs)
    eng.close()
    del eng
    free, total = torch.cuda.mem_get_info()
    print("engine %2d: %d graphs, parked %3d | host RSS %7.1f MB | device in use %7.1f MB | torch reserved %7.1f MB" % (
        k, made, len(E._PARKED_FORKED_GRAPHS), rss_mb(), (total - free) / 2**20, torch.cuda.memory_reserved() / 2**20),
        flush=True)
