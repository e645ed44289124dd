"""How often does the multi-stream eager first iteration differ from the single-stream one?"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cycle_gan_rna_amd import GAN_functions as F
from oracle import cyclegan_oracle as O
from tests.test_dropin_gpu import GOLD, Obs, TwoCol, build_modules

fx = torch.load(os.path.join(GOLD, "function_level.pt"), weights_only=True)["one_iter_epochs"]
gm, gh, N = fx["g_mouse"], fx["g_human"], fx["N"]
xm, _ = O.synthetic_expression(N, gm, 12, seed=3)
xh, _ = O.synthetic_expression(N, gh, 12, seed=4)
lm, lh = O.balanced_labels(N, 12, 5), O.balanced_labels(N, 12, 6)
names = [f"type{i:02d}" for i in range(12)]
cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]


def run(**over):
    cfg = dict(fx["cfg"], device="cuda", progress=False, epochs=2, **over)
    mods = build_modules(gm, gh, cfg, fx["init_seed"])
    cwd = os.getcwd()
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
    return [metrics[k][e] for e in range(2) for k in ("g_loss", "d_m_loss", "d_h_loss", "cycle_loss")]


ref = run(hip_graphs=False, concurrent_streams=False)
for name, over in (("streams+lookahead", dict(hip_graphs=False)), ("streams", dict(hip_graphs=False, lookahead=False)),
                   ("all", dict())):
    bad = 0
    for i in range(int(sys.argv[1])):
        got = run(**over)
        if got != ref:
            bad += 1
            print(name, i, [(a, b) for a, b in zip(got, ref) if a != b][:3], flush=True)
    print(name, "mismatches", bad, "of", sys.argv[1], flush=True)
