"""Throughput of the CycleGAN training iteration on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the reference's inner loop body (GAN_functions.py:298-419):
a critic step, plus a generator step every nb_critic-th step (nb_critic = 4,
GAN_main.py:56), on one batch of synthetic log-normalised expression that is
already resident in HBM.  Workload (BASELINE.json configs[1]): 5000 genes per
species, batch 512 per GPU, default architecture of GAN_main.py:47-77, 12 cell
types.  N > 1: batch-sharded data parallel, 512 cells per rank (weak scaling).

Prints one JSON line (rank 0) with the whole-job cells/s, the GEMM roofline
measured with HIP events on the launch stream, and - at N = 1 - the CPU oracle
timed on the host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

G_GENES = 5000
BATCH = 512
N_TYPES = 12
NB_CRITIC = 4
F32_MFMA_PEAK_TFLOPS = 157.3      # dense float32 matrix peak of MI355X
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 matrix peak


def algorithmic_gflop_per_cell_iteration(G, nb_critic=4, E=12):
    """Necessary multiply-adds of one iteration per cell (2 FLOP each): generator
    and critic passes with the reference's wasted sweeps excluded."""
    Gf = G * 2048 + 2048 * 1024 + 1024 * 512 + 512 * 512 + 268 * 2048 + 2048 * 1024 + 1024 * 512 + 512 * G + 131072
    D1 = (G + E) * 1024
    Dr = 1024 * 512 + 512 * 256 + 256
    Df = D1 + Dr
    d_step = 2 * Gf + 6 * Df + 4 * (Df + Dr + E * 1024) + 2 * Df + 2 * (3 * D1 + 2 * Dr)
    g_step = 12 * Gf - 4096 * G + 10 * Df
    return 2.0 * (d_step + g_step / nb_critic) / 1e9


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(repeats=3, iters=12):
    """The CPU oracle (a float32 restatement of the reference step on torch CPU ops, including the backward sweeps
    the reference runs and discards) on the same shapes, timed on the host cores of this job: `iters` iterations
    (a multiple of nb_critic: each group is 4 critic steps + 1 generator step) per repeat, `repeats` repeats, at two
    thread counts - the cores this job may use and half of them; minimum and median per setting."""
    import statistics
    from oracle import cyclegan_oracle as O
    from oracle import init_params
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    # a one-GPU job on the test box owns 16 cores of a much larger machine (OMP_NUM_THREADS says how many)
    share = min(usable, int(os.environ.get("OMP_NUM_THREADS", usable)))
    cfg = O.default_config(preprocess_type="log_normalized", batch_size=BATCH)
    nets = init_params.build_nets(G_GENES, G_GENES, [N_TYPES], cfg, seed=0)
    xm, _ = O.synthetic_expression(BATCH, G_GENES, N_TYPES, seed=1234)
    xh, _ = O.synthetic_expression(BATCH, G_GENES, N_TYPES, seed=1235)
    lm, lh = O.balanced_labels(BATCH, N_TYPES, 1), O.balanced_labels(BATCH, N_TYPES, 2)
    cm, ch = O.centroid_table(xm, lm, N_TYPES), O.centroid_table(xh, lh, N_TYPES)
    lm, lh = lm[:, None], lh[:, None]
    settings = []
    for threads in sorted({share, max(1, share // 2)}, reverse=True):
        torch.set_num_threads(threads)
        tr = O.OracleTrainer(nets, cfg, cm, ch)
        torch.manual_seed(1)
        tr.iteration(1, xm, xh, lm, lh)            # warm-up: one critic step
        times = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            for i in range(iters):
                tr.iteration(i, xm, xh, lm, lh)
            times.append(time.perf_counter() - t0)
        settings.append(dict(threads=threads, cells_per_s_best=iters * BATCH / min(times),
                             cells_per_s_median=iters * BATCH / statistics.median(times),
                             seconds_min=min(times), seconds_median=statistics.median(times)))
    best = max(settings, key=lambda r: r["cells_per_s_best"])
    return dict(value=best["cells_per_s_best"], unit="cells/s", cores=best["threads"], kind="port",
                cpu_model=_cpu_model(), cpus_in_machine=os.cpu_count(), cpus_usable=usable, settings=settings,
                sample=f"{repeats} repeats of {iters} iterations (G={G_GENES}, B={BATCH}, nb_critic={NB_CRITIC}) of the "
                       f"float32 CPU oracle per thread setting, best repeat of the faster setting reported "
                       f"({best['seconds_min']:.1f} s); includes the reference's discarded backward sweeps")


def api_loop(dev, G, B, epochs=(1, 9), per_epoch=80):
    """The same workload through the public function, train_enhanced_cycle_gan (GAN_functions.py:220), in its default
    mode: 80 iterations per epoch (40 960 cells per species; a generator step every fifth iteration, as in the
    engine-only figure; the loop waits for the device once per epoch, when it averages the epoch's metrics), the nine scalars of :411-419 read back on every generator step, shuffled batches
    gathered on the device, per-epoch bookkeeping.  The function is called twice, for epochs[0] and for epochs[1]
    epochs; the difference of the two wall times over the extra iterations is the steady-state cost of an iteration
    (model upload, graph capture, the closing evaluation pass, figures and checkpoints are in both calls; they vary by
    tens of milliseconds from call to call, so each length is run twice and the shorter wall time counts)."""
    import tempfile
    import numpy as np
    import pandas as pd
    from cycle_gan_rna_amd import GAN_functions as F
    from cycle_gan_rna_amd import synthetic as S

    class Obs:                      # what the loop reads of an AnnData: .obs['myannotations']
        def __init__(self, names):
            self.obs = {"myannotations": pd.Series(list(names))}

    class TwoCol:                   # stands in for the fitted UMAP model of the progress figures
        def transform(self, x):
            return np.asarray(x)[:, :2]

    names = [f"type{i:02d}" for i in range(N_TYPES)]
    n_cells = per_epoch * B
    xm, lm = S.synthetic_expression(n_cells, G, N_TYPES, seed=1234)
    xh, lh = S.synthetic_expression(n_cells, G, N_TYPES, seed=1235)
    cat = np.concatenate([lm.numpy(), lh.numpy()])[:, None]
    walls = []
    cwd = os.getcwd()
    for n_epochs in (1,) + tuple(epochs) * 2:       # the first call pays the process's one-time costs and is not used
        cfg = S.default_config(preprocess_type="log_normalized", batch_size=B, epochs=n_epochs, progress=False,
                               use_lr_decay=False)
        torch.manual_seed(0)
        mods = (F.GeneExpressionGenerator(G, G, [N_TYPES], cfg), F.GeneExpressionGenerator(G, G, [N_TYPES], cfg),
                F.GeneExpressionDiscriminator(G, [N_TYPES], cfg), F.GeneExpressionDiscriminator(G, [N_TYPES], cfg))
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                metrics, _ = F.train_enhanced_cycle_gan(
                    *mods, Obs([names[i] for i in lm.tolist()]), Obs([names[i] for i in lh.tolist()]), xm, xh, cat,
                    xh.numpy()[:, :2], TwoCol(), dict(enumerate(names)), cfg, dev)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t0)
            finally:
                os.chdir(cwd)
        assert all(v == v for v in metrics["g_loss"]), metrics["g_loss"]
    walls = [min(walls[1], walls[3]), min(walls[2], walls[4])]
    iters = (epochs[1] - epochs[0]) * per_epoch
    ms = (walls[1] - walls[0]) / iters * 1e3
    return dict(value=B / (ms * 1e-3), unit="cells/s", ms_per_iteration=ms, iterations=iters,
                wall_seconds={str(e): w for e, w in zip(epochs, walls)},
                what="train_enhanced_cycle_gan on the same workload (default mode: HIP graphs, side streams, lookahead; "
                     "metrics of every generator step read back, one generator step late; batches shuffled and gathered on the device): wall time "
                     f"of {epochs[1]} epochs minus wall time of {epochs[0]} epochs over the {iters} extra iterations")


def big_shard(dev, genes=20000, batch=BATCH, steps=20):
    """One GPU's share of the largest named configuration (about 20 000 ortholog genes, batch 4096 over 8 GPUs = 512
    cells per GPU): the same step, float32, HIP graphs + side streams + lookahead, on one device."""
    from cycle_gan_rna_amd.engine import StepEngine
    from cycle_gan_rna_amd.noise import DeviceNoise
    from cycle_gan_rna_amd.ops import HipOps
    from cycle_gan_rna_amd import synthetic as S
    cfg = S.default_config(preprocess_type="log_normalized", batch_size=batch)
    nets = S.random_networks(genes, genes, [N_TYPES], cfg, seed=0)
    xm, lm = S.synthetic_expression(4 * batch, genes, N_TYPES, seed=1234, device=dev)
    xh, lh = S.synthetic_expression(4 * batch, genes, N_TYPES, seed=1235, device=dev)
    eng = StepEngine(nets, cfg, S.centroid_table(xm, lm, N_TYPES), S.centroid_table(xh, lh, N_TYPES), HipOps(dev), batch)
    lm, lh = lm[:, None].contiguous(), lh[:, None].contiguous()
    noise = DeviceNoise(dev, seed=1, default_generator=True)
    eng.enable_concurrency()
    eng.enable_graphs()
    eng.enable_lookahead()

    def b(i):
        s = (i % 4) * batch
        return xm[s:s + batch], xh[s:s + batch], lm[s:s + batch], lh[s:s + batch]
    def one(i):
        noise.begin_step()
        eng.iteration(i, *b(i), noise, nxt=b(i + 1), metrics=False)
    n0 = settle_graphs(eng, one)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n0, n0 + steps):
        one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = {**eng.d_losses(), **eng.g_losses()}
    assert all(v == v and abs(v) < 1e30 for v in losses.values() if isinstance(v, float)), losses
    eng.close()
    return dict(genes=genes, batch_per_gpu=batch, dtype="f32", steps=steps, ms_per_step=dt / steps * 1e3,
                value=batch * steps / dt, unit="cells/s",
                gflop_per_iteration=algorithmic_gflop_per_cell_iteration(genes) * batch)


def settle_graphs(eng, step, warmup=0):
    """Setup, untimed: every HIP graph of the schedule runs eagerly on its first call, is captured on its second and
    replayed from its third on, and the first replays of a graph are slower than the later ones.  The lookahead
    schedule uses six graphs, some of them once in two nb_critic cycles, so pairs of cycles are run until no new graph
    has appeared and each graph in use has been called four times.  Returns the index of the next iteration: the warm-up and the
    timed steps carry on from it (an index that jumps back makes the schedule issue a forward pass again that it had
    issued ahead, through graphs that the steady state never uses), padded so that the timed region starts with an
    iteration that holds a generator step, whatever `warmup` is.  step(i) runs iteration i."""
    n = 0
    if eng.use_graphs:
        for _ in range(16):
            before = {k: g["calls"] for k, g in eng._graphs.items()}
            for _ in range(2 * NB_CRITIC):      # two cycles: the workspace sets of the lookahead alternate
                step(n)
                n += 1
            # the graphs these cycles used (one that only the very first step of a run uses does not hold things up)
            used = [g["calls"] for k, g in eng._graphs.items() if g["calls"] != before.get(k)]
            if len(eng._graphs) == len(before) and used and min(used) >= 4:
                break
        for _ in range(-warmup % NB_CRITIC):
            step(n)
            n += 1
    return n


def _profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    return json.load(open(path)) if os.path.exists(path) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--genes", type=int, default=G_GENES)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: exact float32 MFMA everywhere; bf16: large dense-layer products on the bf16 MFMA")
    ap.add_argument("--no-bf16-line", action="store_true", help="skip the secondary bf16-MFMA measurement of an f32 run")
    ap.add_argument("--no-api-loop", action="store_true", help="skip the measurement through train_enhanced_cycle_gan")
    ap.add_argument("--no-big-shard", action="store_true", help="skip the 20 000-gene line (one GPU's share of the largest configuration)")
    ap.add_argument("--sequential", action="store_true", help="one stream instead of two")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="do not issue the generators' forward passes ahead of the previous critic work")
    ap.add_argument("--gemm-shapes", action="store_true", help="print the per-shape GEMM time table to stderr")
    ap.add_argument("--only", choices=["d", "g"], help="diagnostic: time only critic steps or only generator steps")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying HIP graphs")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as fresh child processes (this process has made no
        # GPU call yet and makes none), one per GPU, and relay their output; rank 0 prints the JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RCCL in production; "gloo" lets several ranks share one GPU to rehearse this code path (RCCL refuses that)
    backend = os.environ.get("CGRNA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("CGRNA_BENCH_WATCHDOG", "600")), exit=True)
    torch.cuda.set_device(local)
    # CGRNA_DDP_ONE_RANK=1: a process group of one rank and the several-rank schedule (RCCL calls between the parts of
    # a step), to exercise that path on a one-GPU machine
    one_rank_group = world == 1 and bool(os.environ.get("CGRNA_DDP_ONE_RANK"))
    # The collective library prints a version banner on stdout when it starts.  stdout carries the one JSON line and
    # nothing else: until that line is printed, whatever anything writes to file descriptor 1 goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if one_rank_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or one_rank_group:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)

    from cycle_gan_rna_amd.engine import StepEngine
    from cycle_gan_rna_amd.noise import DeviceNoise
    from cycle_gan_rna_amd.ops import HipOps
    from cycle_gan_rna_amd import synthetic as S

    G, B = a.genes, a.batch
    cfg = S.default_config(preprocess_type="log_normalized", batch_size=B)
    nets = S.random_networks(G, G, [N_TYPES], cfg, seed=0)               # same weights on every rank
    n_cells = 20 * B
    xm, lm = S.synthetic_expression(n_cells, G, N_TYPES, seed=1234 + 2 * rank, device=dev)
    xh, lh = S.synthetic_expression(n_cells, G, N_TYPES, seed=1235 + 2 * rank, device=dev)
    cm, ch = S.centroid_table(xm, lm, N_TYPES), S.centroid_table(xh, lh, N_TYPES)
    ops = HipOps(dev, precision="bf16" if a.dtype == "bf16" else "fp32")
    eng = StepEngine(nets, cfg, cm, ch, ops, B)
    lm, lh = lm[:, None].contiguous(), lh[:, None].contiguous()
    noise = DeviceNoise(dev, seed=1 + rank, default_generator=True)
    if not a.sequential:
        eng.enable_concurrency()
    if not a.eager:
        eng.enable_graphs()

    lookahead = not (a.sequential or a.only or a.no_lookahead or world > 1 or one_rank_group)
    if lookahead:
        eng.enable_lookahead()

    def batch(i):
        s = (i % 16) * B
        return xm[s:s + B], xh[s:s + B], lm[s:s + B], lh[s:s + B]

    def step(i):
        s = (i % 16) * B
        noise.begin_step()
        if not a.only:         # the loop body; with lookahead the generators' forward passes are issued ahead of the
            eng.iteration(i, *batch(i), noise, nxt=batch(i + 1) if eng.lookahead else None, metrics=False)   # critics' work
            return
        if a.only != "g":
            eng.d_step(xm[s:s + B], xh[s:s + B], lm[s:s + B], lh[s:s + B], noise)
        if (i % NB_CRITIC == 0 and a.only != "d") or a.only == "g":
            eng.g_step(xm[s:s + B], xh[s:s + B], lm[s:s + B], lh[s:s + B], noise)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    n0 = settle_graphs(eng, step, a.warmup)
    for i in range(n0, n0 + a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(n0 + a.warmup, n0 + a.warmup + a.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = {**eng.d_losses(), **eng.g_losses()}
    assert all(v == v and abs(v) < 1e30 for k, v in losses.items() if isinstance(v, float)), losses

    parts = None
    if world == 1 and not a.only and not a.no_roofline:
        # the two halves of an iteration on their own (each as its own HIP graph, no lookahead): a critic step runs
        # every iteration, a generator step every nb_critic-th
        def time_part(fn, n=20):
            for _ in range(3 if eng.use_graphs else 1):
                noise.begin_step()
                fn(*batch(0), noise)
            fence()
            t0 = time.perf_counter()
            for i in range(n):
                noise.begin_step()
                fn(*batch(i), noise)
            fence()
            return (time.perf_counter() - t0) / n * 1e3
        parts = dict(d_step_ms=time_part(eng.d_step), g_step_ms=time_part(eng.g_step))

    roof = None
    eng.flush()                        # several ranks: apply the critic update that is still in flight
    if not a.no_roofline and rank == 0:
        eng.sync.enabled = False       # only this rank runs the pass below, so it must not issue collectives
        # GEMM launches of NB_CRITIC iterations (4 critic + 1 generator step), launched one by one (no graph replay),
        # each call's kernels carrying a pair of HIP events attached to the dispatches themselves on the launch
        # stream.  A block of filler work is queued first so that the host runs ahead of the GPU.
        eng.use_graphs = False
        eng.lookahead = False
        eng.concurrent = False         # one stream: every launch is timed with the chip to itself
        filler_a = torch.rand(4096, 4096, device=dev)
        filler_c = torch.empty(4096, 4096, device=dev)
        # three repeats of NB_CRITIC iterations; the repeat with the smallest GEMM time is reported (every repeat's
        # total is listed in "gemm_ms_per_iteration_repeats"): one repeat is 342 launches in 10 ms of device time
        repeats = []
        for _ in range(3):
            ops.begin_gemm_timing()
            for i in range(NB_CRITIC):
                for _ in range(24):
                    ops.gemm_untimed(filler_c, filler_a, filler_a, False, True)
                step(i)
            torch.cuda.synchronize()
            repeats.append((ops.end_gemm_timing(), ops.last_gemm_detail))
        (n_launch, gemm_ms, exec_flop, gemm_bytes), ops.last_gemm_detail = min(repeats, key=lambda r: r[0][1])
        if a.gemm_shapes:
            agg = {}
            for shape, ms, fl in ops.last_gemm_detail:
                c = agg.setdefault(shape, [0, 0.0, 0.0])
                c[0] += 1; c[1] += ms; c[2] += fl
            print("   M     N     K tA tB acc dt |  n/it   us/launch   ms/it   TF/s", file=sys.stderr)
            for shape, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print("%5d %5d %5d %2d %2d %3d %2d | %5.1f %10.1f %7.3f %6.1f" % (*shape, n / NB_CRITIC, ms / n * 1e3,
                      ms / NB_CRITIC, fl / ms / 1e9), file=sys.stderr)
        # What the counters say about the same kernels (rocprofv3 --pmc passes of this workload, separate runs;
        # tools/profile_r03.sh writes the summaries): HBM bytes per launch and matrix-core busy share
        traffic, mfma = None, None
        tag = "bf16" if a.dtype == "bf16" else "f32"
        pmc = _profile_json("r03_pmc_%s.json" % tag) or _profile_json("r02_pmc_%s.json" % tag)
        if pmc and (G, B) == (G_GENES, BATCH):
            fam = [v for k, v in pmc["kernels"].items() if k.startswith("gemm")]
            if fam:     # bytes of every kernel of the GEMM family in an iteration, per cg_gemm call (as `achieved` is)
                traffic = sum(f["hbm_bytes_per_launch"] * f["launches_per_iteration"] for f in fam) / (n_launch / NB_CRITIC)
            mfma = pmc.get("mfma")
        alg_flop = algorithmic_gflop_per_cell_iteration(G) * 1e9 * B * NB_CRITIC
        # achieved: the FLOPs actually issued through cg_gemm (all of them algorithmically necessary; 2-3 % of the
        # iteration's algorithmic FLOPs run in the token kernels or are saved by a shared first layer) over the summed
        # durations of those launches - the MFMA kernels' own rate, split-K slab reduces included
        achieved = exec_flop / (gemm_ms * 1e-3) / 1e12
        peak = BF16_MFMA_PEAK_TFLOPS if a.dtype == "bf16" else F32_MFMA_PEAK_TFLOPS
        kname = ("gemm_bf16_kernel (v_mfma_f32_16x16x32_bf16; weights read from bf16 shadows) + the float32 kernels for "
                 "the small products" if a.dtype == "bf16"
                 else "gemm_f32_dma_kernel / gemm_f32_kernel (v_mfma_f32_32x32x2_f32)")
        roof = dict(bound="mfma", achieved=achieved, peak=peak, unit="TFLOP/s",
                    frac=achieved / peak, traffic=traffic, traffic_unit="HBM bytes per cg_gemm call (rocprofv3 PMC: all kernels of the GEMM family in an iteration / calls in an iteration)",
                    operand_bytes_per_launch=gemm_bytes / n_launch,
                    kernel=kname, launches_per_iteration=n_launch / NB_CRITIC,
                    avg_launch_us=gemm_ms * 1e3 / n_launch, gemm_ms_per_iteration=gemm_ms / NB_CRITIC,
                    gemm_ms_per_iteration_repeats=[r[0][1] / NB_CRITIC for r in repeats],
                    timing="HIP events attached to the dispatches (cg_gemm_timing_begin/_end): start of a call's "
                           "first kernel to the end of its last one, split-K slab reduce included",
                    algorithmic_tflops=alg_flop / (gemm_ms * 1e-3) / 1e12, mfma_counters=mfma,
                    hbm_kernels=(pmc or {}).get("hbm_kernels"))
    if world > 1:
        dist.barrier()
    other = None
    if a.dtype == "f32" and not a.no_bf16_line and not a.only and world == 1:      # a one-GPU comparison
        # the named configuration compares float32 with bf16 MFMA: time the same step with the large dense-layer
        # products on the bf16 matrix cores (float32 operands rounded on the fly, float32 accumulate)
        ops16 = HipOps(dev, precision="bf16")
        eng16 = StepEngine(nets, cfg, cm, ch, ops16, B)
        if not a.sequential:
            eng16.enable_concurrency()
        if not a.eager:
            eng16.enable_graphs()

        if lookahead:
            eng16.enable_lookahead()

        def step16(i):
            s = (i % 16) * B
            noise.begin_step()
            eng16.iteration(i, *batch(i), noise, nxt=batch(i + 1) if eng16.lookahead else None, metrics=False)
        n16 = settle_graphs(eng16, step16, a.warmup)
        for i in range(n16, n16 + a.warmup):
            step16(i)
        fence()
        t0 = time.perf_counter()
        for i in range(n16 + a.warmup, n16 + a.warmup + a.steps):
            step16(i)
        fence()
        dt16 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt16], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt16 = float(t.item())
        l16 = {**eng16.d_losses(), **eng16.g_losses()}
        assert all(v == v and abs(v) < 1e30 for v in l16.values() if isinstance(v, float)), l16
        eng16.close()
        other = dict(dtype="bf16", value=B * world * a.steps / dt16, unit="cells/s", ms_per_step=dt16 / a.steps * 1e3,
                     note="same step, dense-layer products with M,N>=128, K>=256 on v_mfma_f32_16x16x32_bf16")
    api = None
    if world == 1 and not one_rank_group and not a.only and not a.no_api_loop and a.dtype == "f32" and not a.eager \
            and not a.sequential:
        eng.close()
        api = api_loop(dev, G, B)
    big = None
    if world == 1 and not one_rank_group and not a.only and not a.no_big_shard and a.dtype == "f32" and not a.eager \
            and not a.sequential \
            and (G, B) == (G_GENES, BATCH):
        big = big_shard(dev)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        line = {
            "metric": "train cells/sec (G+D fwd/bwd/update)", "value": B * world * a.steps / dt, "unit": "cells/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"CycleGAN MLP, {G} genes per species, batch {B} per GPU, nb_critic {NB_CRITIC}, "
                                   f"hidden 2048/1024/512, latent 256, {N_TYPES} cell types, log-normalised synthetic "
                                   f"expression, {'bf16 MFMA for the large layers' if a.dtype == 'bf16' else 'float32 MFMA'}", "global_batch": B * world,
                       "parallelism": f"dp{world}"},
            "roofline": roof, "cpu_baseline": cpu, "bf16_mfma": other, "step_parts": parts,
            "api_loop": api, "shard_of_20k_gene_config": big,
        }
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if world > 1 or one_rank_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
