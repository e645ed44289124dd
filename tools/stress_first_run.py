"""One fresh process: the drop-in loop without and then with the identity extension, HIP graphs on several streams
against the eager one-stream run, as the first training runs of the process.  Prints which metrics differ and from
which epoch (they must never differ).  Usage: python tools/stress_first_run.py"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import faultdbg                                                     # noqa: E402
faultdbg.install()       # a crash prints the native stack of the faulting thread and the Python frames of every thread

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import tests.test_dropin_gpu as T                                   # noqa: E402
from cycle_gan_rna_amd import GAN_functions as F                    # noqa: E402
from oracle import cyclegan_oracle as O                             # noqa: E402

fx = torch.load(os.path.join(T.GOLD, "function_level.pt"), weights_only=True)["one_iter_epochs"]
gm, gh0, N = fx["g_mouse"], fx["g_human"], fx["N"]
names = [f"type{i:02d}" for i in range(12)]
NB = int(os.environ.get("STRESS_ITERS", 1)) * N          # cells per species: STRESS_ITERS iterations per epoch
lm, lh = O.balanced_labels(NB, 12, 5), O.balanced_labels(NB, 12, 6)
cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]


def run(identity, **over):
    gh = gm if identity else gh0
    xm, _ = O.synthetic_expression(NB, gm, 12, seed=3)
    xh, _ = O.synthetic_expression(NB, gh, 12, seed=4)
    cfg = dict(fx["cfg"], device="cuda", progress=False, epochs=int(os.environ.get("STRESS_EPOCHS", 4)))
    cfg.update(over)
    if identity:
        cfg["lambda_identity"] = 0.5
    mods = T.build_modules(gm, gh, cfg, fx["init_seed"])
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(11)
            np.random.seed(0)
            metrics, _ = F.train_enhanced_cycle_gan(
                *mods, T.Obs([names[i] for i in lm.tolist()]), T.Obs([names[i] for i in lh.tolist()]), xm, xh, cat,
                xh.numpy()[:, :2], T.TwoCol(), dict(enumerate(names)), cfg, torch.device("cuda:0"))
        finally:
            os.chdir(cwd)
    return metrics, {k: v.detach().cpu().clone() for k, v in mods[0].state_dict().items()}


from cycle_gan_rna_amd.engine import StepEngine                     # noqa: E402
_state_dict = StepEngine.state_dict


def _checked_state_dict(self, name):
    """Read the parameters three ways and report any disagreement: a host copy, a second host copy after a device
    synchronisation, and a host copy of a device-side clone."""
    first = _state_dict(self, name)
    torch.cuda.synchronize()
    second = _state_dict(self, name)
    clones = {k: v.detach().clone() for k, v in self.nets[name].P.items()}
    torch.cuda.synchronize()
    for k, v in clones.items():
        a, b, c = first[k], second[k], v.cpu()
        if not (torch.equal(a, b) and torch.equal(b, c)):
            ia = (a != c).nonzero().flatten().tolist()[:4]
            ib = (b != c).nonzero().flatten().tolist()[:4]
            print("    READBACK", name, k, "first copy differs from clone at", ia, [float(a.flatten()[i]) for i in ia],
                  [float(c.flatten()[i]) for i in ia], "second at", ib, flush=True)
    return first


if os.environ.get("STRESS_CHECK_READBACK"):
    StepEngine.state_dict = _checked_state_dict
# STRESS_OVER: a dict literal of configuration keys for the graph run, e.g. "dict(lookahead=False)"
over = eval(os.environ.get("STRESS_OVER", "{}"))
for identity in (False, True):
    print("stage identity", identity, "graphs+streams start", flush=True)
    fast, wf = run(identity, **over)
    print("stage identity", identity, "eager start", flush=True)
    plain, wp = run(identity, hip_graphs=False, concurrent_streams=False)
    print("stage identity", identity, "compare", flush=True)
    d = {k: [i for i, (x, y) in enumerate(zip(fast[k], plain[k])) if x != y] for k in fast}
    d = {k: v for k, v in d.items() if v}
    w = [k for k in wf if not torch.equal(wf[k], wp[k])]
    print("identity", identity, "DIFFERS" if d or w else "same", d, w[:4], flush=True)
    for k in w[:3]:      # where and by how much: element indices that differ, the largest difference, the size of an update
        a, b = wf[k].flatten(), wp[k].flatten()
        idx = (a != b).nonzero().flatten()
        print("   ", k, tuple(wf[k].shape), "differ:", idx.numel(), "of", a.numel(), "first/last", int(idx[0]), int(idx[-1]),
              "graphs", a[idx].tolist(), "eager", b[idx].tolist(), "mean |w| %.3g" % float(b.abs().mean()), flush=True)
