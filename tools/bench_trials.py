"""Throughput of the hyper-parameter search workload: independent small trials, one worker process per GPU
(several per GPU on request), no collective.  Each trial draws its architecture and weights like
GAN_optimize_optuna.ipynb cell-1 (batch 16/32/64, hidden sizes from the three choices, latent 128-512, 3-6 blocks),
trains `--epochs` curriculum epochs on synthetic counts (`--genes` genes, `--cells` cells per species) and
evaluates.  Prints one JSON line: trials/min and cells/s over all workers.

    python tools/bench_trials.py --trials 64 --genes 2000 --cells 1024 --epochs 3 --workers-per-gpu 2
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ARGS = None


def make_trainer(device):
    import torch
    from cycle_gan_rna_amd import curriculum as C
    from cycle_gan_rna_amd.synthetic import synthetic_expression
    genes, cells = int(os.environ["CG_TRIAL_GENES"]), int(os.environ["CG_TRIAL_CELLS"])
    xm, lm = synthetic_expression(cells, genes, 12, seed=1)
    xh, lh = synthetic_expression(cells, genes, 12, seed=2)
    return C.CurriculumTrainer(torch.expm1(xm).round(), torch.expm1(xh).round(), lm, lh, 12, device=device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=64)
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--cells", type=int, default=1024)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--workers-per-gpu", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--phase", choices=["drawn", "full"], default="drawn",
                    help="full: every epoch in the last curriculum phase (critic step every iteration); drawn: the "
                         "trial's own schedule, whose first 10-50 epochs train the generators only")
    a = ap.parse_args()
    os.environ["CG_TRIAL_GENES"], os.environ["CG_TRIAL_CELLS"] = str(a.genes), str(a.cells)
    from cycle_gan_rna_amd import curriculum as C
    t0 = time.time()
    recs = C.run_trials(make_trainer, a.trials, out_dir=a.out, gpus=a.gpus, workers_per_gpu=a.workers_per_gpu,
                        seed=42, epochs=a.epochs,
                        curriculum=dict(warmup_epochs=0, cycle_focus_epochs=0, full_training_start=0)
                        if a.phase == "full" else None)
    wall = time.time() - t0
    done = [r for r in recs if r and r["status"] == "complete"]
    failed = [r for r in recs if r and r["status"] != "complete"]
    cells = sum(r["cells"] for r in done)
    train_s = sum(r["metrics"]["train_seconds"] for r in done)
    print(json.dumps({"workload": f"{a.trials} curriculum trials, {a.genes} genes, {a.cells} cells per species, "
                                  f"{a.epochs} epochs each, phase {a.phase}", "gpus": a.gpus, "workers_per_gpu": a.workers_per_gpu,
                      "wall_s": round(wall, 2), "completed": len(done), "not_completed": len(failed),
                      "trials_per_min": round(60.0 * len(done) / wall, 2), "cells_per_s": round(cells / wall, 1),
                      "cells_per_s_inside_training": round(cells / max(train_s, 1e-9) * max(1, len({r['worker'] for r in done})), 1),
                      "reasons": sorted({r.get("reason", "")[:80] for r in failed})}))


if __name__ == "__main__":
    main()
