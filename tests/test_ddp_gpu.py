"""The device code path of the data-parallel step (side comm stream, events, scalar sync) with two ranks that
share the one GPU of the test box.  RCCL refuses two ranks on one device, so the ranks talk over gloo; the engine
code is the same as under RCCL (sum + scale on a side stream, then the fused Adam)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out, graphs):
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)      # a hung rank reports where, then exits
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from cycle_gan_rna_amd.engine import StepEngine
    from cycle_gan_rna_amd.noise import DeviceNoise
    from cycle_gan_rna_amd.ops import HipOps
    from oracle import cyclegan_oracle as O
    from tests.helpers import make_problem, small_cfg
    cfg = small_cfg(preprocess_type="log_normalized", nb_critic=1)
    nets, xm, xh, lm, lh = make_problem(64, 64, 32, cfg)
    cm, ch = O.centroid_table(xm, lm, 12), O.centroid_table(xh, lh, 12)
    b = 16
    sl = slice(rank * b, (rank + 1) * b)
    eng = StepEngine(nets, cfg, cm, ch, HipOps("cuda:0"), b)
    eng.enable_concurrency()
    if graphs:
        eng.enable_graphs()
    assert eng.sync.enabled
    noise = DeviceNoise("cuda:0", seed=5 + rank, default_generator=True)
    for i in range(5):
        noise.begin_step()
        res = eng.iteration(i, xm[sl].cuda(), xh[sl].cuda(), lm[sl, None].cuda(), lh[sl, None].cuda(), noise)
    assert not graphs or all(g["graph"] is not None for g in eng._graphs.values())
    torch.cuda.synchronize()
    eng.flush()      # apply the critic update that is still in flight
    torch.save(dict(d=eng.arena_d.p.cpu(), g=eng.arena_g.p.cpu(), coef=eng.coef.cpu(), g_loss=res["g_loss"]),
               os.path.join(out, f"rank{rank}_{int(graphs)}.pt"))
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_stay_identical(tmp_path):
    world, port = 2, 29600 + os.getpid() % 2000
    res = {}
    for graphs in (False, True):        # eager, then every device part of a step as a HIP graph with eager collectives
        mp.spawn(_worker, args=(world, port + int(graphs), str(tmp_path), graphs), nprocs=world, join=True)
        r0 = torch.load(tmp_path / f"rank0_{int(graphs)}.pt", weights_only=True)
        r1 = torch.load(tmp_path / f"rank1_{int(graphs)}.pt", weights_only=True)
        assert torch.isfinite(r0["d"]).all() and torch.isfinite(r0["g"]).all()
        assert torch.equal(r0["d"], r1["d"]) and torch.equal(r0["g"], r1["g"]), "ranks diverged"
        assert torch.equal(r0["coef"], r1["coef"])
        res[graphs] = r0
    assert torch.equal(res[False]["d"], res[True]["d"]) and torch.equal(res[False]["g"], res[True]["g"])


def test_bench_with_two_ranks(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (one process per rank under torch.distributed.run),
    with the two ranks sharing the test GPU over gloo: the timed loop, the max over ranks, the GEMM timing pass that
    only rank 0 runs (it must not issue collectives the other rank does not join) and the one JSON line."""
    import json
    import subprocess
    env = dict(os.environ, CGRNA_BENCH_BACKEND="gloo", CGRNA_BENCH_WATCHDOG="90")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(31600 + os.getpid() % 2000), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "4", "--warmup", "2", "--genes", "256", "--batch", "32"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["config"]["global_batch"] == 64
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0 and out["cpu_baseline"] is None
