"""Repeat the drop-in training loop with the identity extension in several schedules and count the runs whose metrics
differ from the eager one-stream run (they must never differ).  Usage: python tools/stress_graph_vs_eager.py [reps]"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import tests.test_dropin_gpu as T                                   # noqa: E402
from cycle_gan_rna_amd import GAN_functions as F                    # noqa: E402
from cycle_gan_rna_amd.engine import StepEngine                     # noqa: E402
from oracle import cyclegan_oracle as O                             # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2 and sys.argv[2] == "poison":
    # every uninitialised float buffer on the device starts as NaN: a kernel that reads what nothing wrote shows up as
    # a non-finite metric or as a difference between two runs
    _empty, _empty_like = torch.empty, torch.empty_like

    def _poisoned(t):
        if t.is_cuda and t.is_floating_point() and t.numel():
            t.fill_(float("nan"))
        return t
    torch.empty = lambda *a, **k: _poisoned(_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poisoned(_empty_like(*a, **k))
fx = torch.load(os.path.join(T.GOLD, "function_level.pt"), weights_only=True)["one_iter_epochs"]
gm, N = fx["g_mouse"], fx["N"]
xm, _ = O.synthetic_expression(N, gm, 12, seed=3)
xh, _ = O.synthetic_expression(N, gm, 12, seed=4)
lm, lh = O.balanced_labels(N, 12, 5), O.balanced_labels(N, 12, 6)
names = [f"type{i:02d}" for i in range(12)]
cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]


def run(**over):
    cfg = dict(fx["cfg"], device="cuda", progress=False, epochs=4, lambda_identity=0.5, **over)
    mods = T.build_modules(gm, gm, cfg, fx["init_seed"])
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
    assert all(np.isfinite(v).all() for v in metrics.values()), {k: v for k, v in metrics.items() if not np.isfinite(v).all()}
    return metrics


def first_diff(a, b):
    out = {}
    for k in a:
        d = [i for i, (x, y) in enumerate(zip(a[k], b[k])) if x != y]
        if d:
            out[k] = d[0]
    return out


plain = run(hip_graphs=False, concurrent_streams=False)
variants = {"default": {}, "no_lookahead": dict(lookahead=False), "no_graphs": dict(hip_graphs=False),
            "streams_only": dict(hip_graphs=False, lookahead=False), "graphs_only": dict(concurrent_streams=False),
            "no_reuse": {}, "plain": dict(hip_graphs=False, concurrent_streams=False)}
bad = {k: 0 for k in variants}
for r in range(reps):
    for name, over in variants.items():
        StepEngine.reuse_encoder = name != "no_reuse"
        d = first_diff(run(**over), plain)
        if d:
            bad[name] += 1
            print(r, name, "DIFFERS, first epoch per metric:", d, flush=True)
    print("rep", r, bad, flush=True)
print("runs that differ, of", reps, ":", bad)
