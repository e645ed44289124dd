"""Fast (HIP graphs, several streams, forwards issued ahead) against plain (eager, one stream) runs of the drop-in
training loop with one seed, many times over, alternating between two configurations (with and without the identity
term, which also changes the gene counts) as the test suite does.  Prints every run whose metrics differ from the
first run of its configuration: a difference within one mode is a race or a read of memory nobody wrote, a
difference between the modes a schedule that changes the arithmetic.

    python tools/diag_fast_plain.py [repetitions] [iters_per_epoch]
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_dropin_gpu import GOLD, Obs, TwoCol, build_modules  # noqa: E402

from cycle_gan_rna_amd import GAN_functions as F  # noqa: E402
from oracle import cyclegan_oracle as O  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
# "sync": device synchronize + garbage collection before every run; "check": compare an engine's parameters with the
# modules it was built from right after it was built (which also waits for the device)
variant = sys.argv[3] if len(sys.argv) > 3 else ""
flavour = sys.argv[4] if len(sys.argv) > 4 else "both"    # what a "fast" run switches on: "graphs", "streams" or both
ipe = float(sys.argv[2]) if len(sys.argv) > 2 else 1
fx = torch.load(os.path.join(GOLD, "function_level.pt"), weights_only=True)["one_iter_epochs"]
names = [f"type{i:02d}" for i in range(12)]


def train(identity, mode):
    gm, gh, N = fx["g_mouse"], fx["g_human"], fx["N"]
    nm = nh = int(N * ipe)
    if identity:
        gh = gm
    xm, _ = O.synthetic_expression(nm, gm, 12, seed=3)
    xh, _ = O.synthetic_expression(nh, gh, 12, seed=4)
    lm, lh = O.balanced_labels(nm, 12, 5), O.balanced_labels(nh, 12, 6)
    cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]
    cfg = dict(fx["cfg"], device="cuda", progress=False, epochs=4, hip_graphs=mode == "fast" and flavour != "streams",
               concurrent_streams=mode == "fast" and flavour != "graphs")
    if identity:
        cfg["lambda_identity"] = 0.5
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
    return metrics


# check the parameters an engine holds right after it was built against the modules it was built from
from cycle_gan_rna_amd.engine import StepEngine  # noqa: E402

_init = StepEngine.__init__


def _checked_init(self, nets, *a, **k):
    _init(self, nets, *a, **k)
    for name, sd in nets.items():
        got = self.state_dict(name)
        for key, v in sd.items():
            if not torch.equal(got[key].cpu().to(v.dtype), v.detach().cpu()):
                print(f"engine built with a wrong copy of {name}.{key}", flush=True)


if "check" in variant:
    StepEngine.__init__ = _checked_init

# fingerprints (float64 sums kept on the device, read when the run is over: no extra host synchronisation) of the
# optimizer state after the engine is built and after every iteration
prints = []
_iteration = StepEngine.iteration


def _fp(eng, tag):
    for an in ("arena_g", "arena_d"):
        ar = getattr(eng, an)
        for bn in ("p", "g", "m", "v", "vmax"):
            prints.append((f"{tag} {an}.{bn}", getattr(ar, bn).double().sum()))
    for name, net in eng.nets.items():
        prints.append((f"{tag} {name} buffers", sum(v.double().sum() for v in net.B.values())))
    prints.append((f"{tag} ema", eng.ema.sum()))
    prints.append((f"{tag} d_tot", eng.d_tot.sum()))
    prints.append((f"{tag} gloss", eng.gloss.sum()))


def _fp_init(self, *a, **k):
    (_checked_init if "check" in variant else _init)(self, *a, **k)
    self._it = 0
    _fp(self, "built")


def _fp_iteration(self, *a, **k):
    out = _iteration(self, *a, **k)
    _fp(self, f"iteration {self._it}")
    self._it += 1
    return out


if "fp" in variant:
    StepEngine.__init__ = _fp_init
    StepEngine.iteration = _fp_iteration
first_fp = {}

# what was uploaded, as the device saw it right after the upload (cloned there, no host synchronisation), compared with
# the host's copy when the run is over
uploads = []
_batches_init = F._ShuffledBatches.__init__


def _batches_init_rec(self, x, labels, batch_size):
    _batches_init(self, x, labels, batch_size)
    uploads.append(("dataset", x.clone(), None))
    uploads.append(("labels", labels.clone(), None))


def _upl_init(self, nets, *a, **k):
    (_fp_init if "fp" in variant else _init)(self, nets, *a, **k)
    for name, sd in nets.items():
        for key, v in sd.items():
            if key in self.nets[name].P:
                uploads.append((f"{name}.{key}", self.nets[name].P[key].clone(), v.detach().clone()))
    for an in ("arena_g", "arena_d"):
        ar = getattr(self, an)
        for bn in ("g", "m", "v", "vmax"):
            uploads.append((f"{an}.{bn} (zeros)", getattr(ar, bn).clone(), torch.zeros(ar.n)))


# when does a copy change?  integer checksums of all copies, taken on the device after every iteration and whenever
# the engine's state is read out (no host synchronisation), read when the run is over
sums = []


def _checksum(tag):
    if uploads:
        sums.append((tag, torch.stack([d.reshape(-1).view(torch.int32).sum(dtype=torch.int64) for _, d, h in uploads
                                       if h is not None and d.dtype == torch.float32]).sum()))


def _upl_iteration(self, *a, **k):
    self._it = getattr(self, "_it", 0)
    out = (_fp_iteration if "fp" in variant else _iteration)(self, *a, **k)
    _checksum(f"after iteration {self._it}")
    self._it += 1
    return out


_state_dict = StepEngine.state_dict


def _upl_state_dict(self, name):
    _checksum(f"before state_dict({name})")
    return _state_dict(self, name)


if "uploads" in variant:
    F._ShuffledBatches.__init__ = _batches_init_rec
    StepEngine.__init__ = _upl_init
    StepEngine.iteration = _upl_iteration
    StepEngine.state_dict = _upl_state_dict
first_upl = {}

# "keepgraphs": the HIP graphs of every engine stay alive until the process ends (is it their destruction that matters?)
kept = []
_enable_graphs = StepEngine.enable_graphs


def _enable_graphs_keep(self):
    _enable_graphs(self)
    kept.append(self._graphs)


if "keepgraphs" in variant:
    StepEngine.enable_graphs = _enable_graphs_keep

# every product issued twice (outside graph capture) and the two results compared on the device: a product that is
# not a pure function of its operands shows here, with its shape
from cycle_gan_rna_amd.ops import HipOps  # noqa: E402

twice = []
_gemm = HipOps.gemm


def _gemm_twice(self, Cm, A, B, transA, transB, **kw):
    if torch.cuda.is_current_stream_capturing():
        return _gemm(self, Cm, A, B, transA, transB, **kw)
    c0 = Cm.clone() if kw.get("accumulate") else None
    _gemm(self, Cm, A, B, transA, transB, **kw)
    r1 = Cm.clone()
    if c0 is not None:
        Cm.copy_(c0)
    _gemm(self, Cm, A, B, transA, transB, **kw)
    twice.append(((tuple(Cm.shape), tuple(A.shape), transA, transB, kw.get("accumulate", False), kw.get("epi", 0)),
                  (Cm != r1).sum()))


if "gemm2" in variant:
    HipOps.gemm = _gemm_twice


def poison_free_blocks():
    """Fill every block the caching allocator holds free with NaN: a later read of recycled memory that nobody has
    written since shows as NaN instead of as the (often identical) values of the previous run."""
    torch.cuda.synchronize()
    hold = []
    for seg in torch.cuda.memory_snapshot():
        for blk in seg["blocks"]:
            if blk["state"] == "inactive" and blk["size"] >= 512:
                try:
                    hold.append(torch.full((blk["size"] // 4,), float("nan"), device="cuda"))
                except RuntimeError:
                    pass
    torch.cuda.synchronize()
    del hold


first = {}
n_bad = 0
for rep in range(reps):
    for identity in (False, True):
        for mode in ("fast", "plain"):
            if "sync" in variant:
                import gc
                torch.cuda.synchronize()
                gc.collect()
                torch.cuda.synchronize()
            if "poison" in variant:
                poison_free_blocks()
            del uploads[:]
            del sums[:]
            del prints[:]
            del twice[:]
            m = train(identity, mode)
            if uploads:
                _checksum("after the run returned")
                cs = [(t, int(v)) for t, v in sums]
                if len(set(v for _, v in cs)) > 1:
                    print(f"rep {rep} identity={identity} {mode}: checksum of the copies over time:",
                          [(t, v - cs[0][1]) for t, v in cs], flush=True)
                ptrs = {n: d.data_ptr() for n, d, _ in uploads}
                got = [(n, d.cpu(), h) for n, d, h in uploads]
                ref_u = first_upl.setdefault(identity, got)
                for (n, d, h), (_, d0, _) in zip(got, ref_u):
                    want = h if h is not None else d0
                    if not torch.equal(d.to(want.dtype).reshape(-1), want.reshape(-1)):
                        w = (d.to(want.dtype).reshape(-1) != want.reshape(-1)).nonzero()
                        dd, ww = d.to(want.dtype).reshape(-1), want.reshape(-1)
                        print(f"rep {rep} identity={identity} {mode}: copy of {n} ({d.numel()} elements, device address "
                              f"{hex(ptrs[n])}) changed at", [(int(i), float(ww[i]), float(dd[i]),
                                                              hex(dd[i:i + 1].view(torch.int32).item() & 0xffffffff))
                                                             for i in w.reshape(-1)[:8]], flush=True)
            if twice:
                counts = torch.stack([c for _, c in twice]).cpu().tolist()
                for (desc, _), c in zip(twice, counts):
                    if c:
                        print(f"rep {rep} identity={identity} {mode}: product {desc} gave two results ({c} elements differ)", flush=True)
            fp = [(n, float(v)) for n, v in prints]
            ref_fp = first_fp.setdefault((identity, mode), fp)
            wrong = [(n, a, b) for (n, a), (_, b) in zip(fp, ref_fp) if a != b]
            if wrong:
                print(f"rep {rep} identity={identity} {mode}: {len(wrong)} fingerprints differ, first:", wrong[:6], flush=True)
            ref = first.setdefault(identity, m)
            bad = {k: [i for i, (a, b) in enumerate(zip(m[k], ref[k])) if a != b] for k in m}
            bad = {k: v for k, v in bad.items() if v}
            if bad:
                n_bad += 1
                print(f"rep {rep} identity={identity} {mode}: differs at", bad, flush=True)
                for k in bad:
                    print("   ", k, [f"{a:.9f}/{b:.9f}" for a, b in zip(m[k], ref[k])], flush=True)
    if rep % 10 == 9:
        ms = torch.cuda.memory_stats()
        print(f"... {rep + 1} repetitions, {n_bad} deviating runs; device allocations {ms.get('num_device_alloc')}, "
              f"frees {ms.get('num_device_free')}, reserved {ms.get('reserved_bytes.all.current', 0) >> 20} MiB", flush=True)
print(f"{reps} repetitions x 2 configurations x 2 modes: {n_bad} deviating runs", flush=True)
