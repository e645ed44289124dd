"""Batch-sharded data parallelism, emulated in one process with the CPU oracle.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/).  The reference has no
distributed code; this is the definition the data-parallel step of the product
is checked against: K replicas of the step of GAN_functions.py:298-419 start from
identical weights, each pushes its own shard of the batch through the step,
gradients are averaged over the replicas before each optimizer step
(:339-340, :409), and the five loss values that feed the dynamic loss scaling
(:380-399) are averaged over the replicas first, so every replica derives the
same scales.  BatchNorm batch statistics and the feature-matching batch means
stay local to a replica.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from . import cyclegan_oracle as O


def _average_grads(opts):
    """opts: the same optimizer of every replica; .grad of every parameter <- mean over replicas."""
    for ps in zip(*(o.params for o in opts)):
        if ps[0].grad is None:
            assert all(p.grad is None for p in ps)
            continue
        g = sum(p.grad for p in ps) / len(ps)
        for p in ps:
            p.grad = g.clone()


class ReplicatedTrainer:
    def __init__(self, nets, cfg, mouse_centroids, human_centroids, seeds: Sequence[int]):
        self.cfg = cfg
        self.replicas: List[O.OracleTrainer] = [
            O.OracleTrainer(nets, cfg, mouse_centroids, human_centroids, noise=O.GeneratorNoise(s)) for s in seeds]

    def iteration(self, i, shards):
        """shards: per replica (mouse, human, mouse_cat, human_cat).  Returns the metrics of replica 0 with the loss
        values replaced by their means over the replicas (what every rank reports)."""
        R = self.replicas
        outs = [tr.d_step(*sh, step=False) for tr, sh in zip(R, shards)]
        for name in ("opt_dm", "opt_dh"):
            _average_grads([getattr(tr, name) for tr in R])
            for tr in R:
                getattr(tr, name).step()
        out = dict(outs[0])
        if i % self.cfg["nb_critic"] == 0:
            Ls = [tr.g_losses(*sh) for tr, sh in zip(R, shards)]
            keys = ("g_adv", "g_cycle", "g_celltype", "g_fm", "kl")
            glob = {k: sum(L[k].item() for L in Ls) / len(Ls) for k in keys}
            for tr in R:
                tr.g_scales(dict(glob))
            gouts = [tr.g_finish(L, step=False) for tr, L in zip(R, Ls)]
            _average_grads([tr.opt_g for tr in R])
            for tr in R:
                tr.opt_g.step()
            out.update(gouts[0])
            out.update(g_adv=glob["g_adv"], cycle_loss=glob["g_cycle"], celltype_loss=glob["g_celltype"],
                       fm_loss=glob["g_fm"], kl_loss=glob["kl"],
                       g_loss=sum(g["g_loss"] for g in gouts) / len(gouts))
        return out
