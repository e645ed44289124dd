"""Batch-sharded data parallelism on two CPU processes (gloo): every rank ends
a critic step and a generator step with the parameters that one process gets
by running both shards itself, averaging the gradients and applying one Adam
update - and both ranks hold identical parameters."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    from oracle import cyclegan_oracle as O
    from tests.helpers import make_problem, small_cfg
    cfg = small_cfg(preprocess_type="log_normalized", nb_critic=1)
    nets, xm, xh, lm, lh = make_problem(40, 36, 24, cfg)
    return cfg, nets, xm, xh, lm[:, None], lh[:, None], O.centroid_table(xm, lm, 12), O.centroid_table(xh, lh, 12)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from cycle_gan_rna_amd.engine import StepEngine
    from cycle_gan_rna_amd.noise import HostReplayNoise
    from oracle.ops_cpu import CpuOps
    cfg, nets, xm, xh, lm, lh, cm, ch = _problem()
    b = xm.shape[0] // world
    sl = slice(rank * b, (rank + 1) * b)
    eng = StepEngine(nets, cfg, cm, ch, CpuOps(), b)
    assert eng.sync.enabled and eng.sync.world == world
    torch.manual_seed(100 + rank)
    res = eng.iteration(0, xm[sl], xh[sl], lm[sl], lh[sl], HostReplayNoise("cpu"))
    eng.flush()      # apply the critic update that is still in flight
    torch.save(dict(d=eng.arena_d.p.clone(), g=eng.arena_g.p.clone(), coef=eng.coef.clone(), g_loss=res["g_loss"]),
               os.path.join(out, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_step_equals_gradient_averaging(tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(r0["d"], r1["d"]) and torch.equal(r0["g"], r1["g"]), "ranks diverged"
    assert torch.equal(r0["coef"], r1["coef"]), "ranks derived different loss scales"

    # single-process emulation: both shards, gradients averaged, one update
    sys.path.insert(0, ROOT)
    from cycle_gan_rna_amd.engine import StepEngine
    from cycle_gan_rna_amd.noise import HostReplayNoise
    from oracle.ops_cpu import CpuOps
    cfg, nets, xm, xh, lm, lh, cm, ch = _problem()
    b = xm.shape[0] // world
    engines = [StepEngine(nets, cfg, cm, ch, CpuOps(), b) for _ in range(world)]

    class NoAdam:
        """Runs the step of each shard but keeps the parameters frozen, collecting the gradients."""
    grads_d, grads_g = [], []
    for r, eng in enumerate(engines):
        eng._adam = lambda arena, lo=0, hi=None: None
        torch.manual_seed(100 + r)
        noise = HostReplayNoise("cpu")
        sl = slice(r * b, (r + 1) * b)
        eng.d_step(xm[sl], xh[sl], lm[sl], lh[sl], noise)
        grads_d.append(eng.arena_d.g.clone())
    # critic update with the averaged gradient, applied to every emulated replica
    gd = sum(grads_d) / world
    ops = CpuOps()
    for eng in engines:
        eng.arena_d.g.copy_(gd)
        eng.arena_d.step = 1
        eng.arena_d.lr_step.copy_(torch.tensor([cfg["lr"], 1.0], dtype=torch.float64))
        ops.adam(eng.arena_d.p, eng.arena_d.g, eng.arena_d.m, eng.arena_d.v, eng.arena_d.vmax, eng.arena_d.lr_step,
                 0.5, 0.9, 1e-8)
    assert float((engines[0].arena_d.p - r0["d"]).abs().max()) <= 1e-6


def test_two_rank_step_matches_replicated_oracle(tmp_path):
    """The same two-rank run against the oracle's emulation of data parallelism (oracle/ddp_emulation.py): replicas
    with their own noise, averaged gradients, averaged loss values behind the loss scales."""
    world, port = 2, 29400 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    sys.path.insert(0, ROOT)
    from oracle.ddp_emulation import ReplicatedTrainer
    from cycle_gan_rna_amd.engine import StepEngine
    from oracle.ops_cpu import CpuOps
    cfg, nets, xm, xh, lm, lh, cm, ch = _problem()
    b = xm.shape[0] // world
    rt = ReplicatedTrainer(nets, cfg, cm, ch, seeds=[100, 101])
    out = rt.iteration(0, [(xm[k * b:(k + 1) * b], xh[k * b:(k + 1) * b], lm[k * b:(k + 1) * b], lh[k * b:(k + 1) * b])
                           for k in range(world)])
    assert abs(out["g_loss"] - r0["g_loss"]) <= 1e-4 * max(abs(out["g_loss"]), 0.5)
    # parameters: lay the oracle's tensors out like the engine's arenas and compare the applied update
    eng = StepEngine(nets, cfg, cm, ch, CpuOps(), b)
    tr = rt.replicas[0]
    for arena, key in ((eng.arena_d, "d"), (eng.arena_g, "g")):
        ref, init = torch.zeros_like(arena.p), arena.p.clone()
        for (net, k), (off, n, shape) in arena.offsets.items():
            ref[off:off + n] = tr.nets[net][k].detach().reshape(-1)
        moved = (ref - init).abs() > 0.5 * cfg["lr"]
        d = (r0[key][moved].double() - ref[moved].double()).norm() / (ref[moved].double() - init[moved].double()).norm()
        assert float(d) <= 5e-2, (key, float(d))
