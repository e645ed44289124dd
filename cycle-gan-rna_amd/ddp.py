"""Batch-sharded data parallelism for the step engine: one process per GPU, each
runs the whole step on its shard of the batch; gradients are averaged with RCCL
(torch.distributed backend "nccl") before the fused Adam, so every rank applies
the same update.

The reference has no distributed code; the conventions chosen here:
  * gradients: mean over ranks of each optimizer's flat gradient arena, sent in
    buckets on a side stream as soon as a network's gradients are complete, so
    the transfer overlaps the remaining backward work
  * the ten loss sums feeding the dynamic loss scaling (GAN_functions.py:380-406)
    are summed over ranks, so every rank derives identical scales
  * BatchNorm batch statistics and the feature-matching batch means stay local
    to a rank (the PyTorch DistributedDataParallel convention)
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, device, bucket_bytes=64 << 20):
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.enabled else 1
        self.device = torch.device(device)
        self.is_cuda = self.device.type == "cuda"
        self.bucket_elems = bucket_bytes // 4
        self.comm = torch.cuda.Stream(device=self.device) if (self.enabled and self.is_cuda) else None
        self.pending = []
        # The gloo backend stages device tensors through the host on a helper thread; issued right behind a HIP
        # graph replay that staging never completes on this stack, so gloo (CPU clusters and tests; RCCL is the
        # production backend) gets a host synchronisation before every collective.
        self.host_sync = self.enabled and self.is_cuda and dist.get_backend() == "gloo"

    def launch(self, flat_grad):
        """Start averaging `flat_grad` (a contiguous 1-D view whose producer kernels
        have been enqueued on the current stream)."""
        if not self.enabled:
            return
        if self.host_sync:
            torch.cuda.synchronize(self.device)
        if self.is_cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            self.comm.wait_event(ready)
            with torch.cuda.stream(self.comm):
                for s in range(0, flat_grad.numel(), self.bucket_elems):
                    chunk = flat_grad[s:s + self.bucket_elems]
                    # sum + scale rather than ReduceOp.AVG: identical on RCCL, and also valid on gloo, which lets
                    # the device code path be exercised by two ranks sharing one GPU
                    dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
                    chunk.mul_(1.0 / self.world)
            done = torch.cuda.Event()
            done.record(self.comm)
            self.pending.append(done)
        else:   # CPU tensors (gloo tests)
            for s in range(0, flat_grad.numel(), self.bucket_elems):
                chunk = flat_grad[s:s + self.bucket_elems]
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
                chunk.div_(self.world)

    def wait(self):
        """Make the current stream wait for every launched bucket."""
        for ev in self.pending:
            torch.cuda.current_stream(self.device).wait_event(ev)
        self.pending = []

    def sum_scalars(self, t):
        """Sum a small tensor over ranks, in place, on the current stream."""
        if self.host_sync:
            torch.cuda.synchronize(self.device)
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
