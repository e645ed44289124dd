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

import os

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, device, bucket_bytes=64 << 20, stream=None):
        # CGRNA_DDP_ONE_RANK: take the several-rank code path (collectives between the parts of a step) with a process
        # group of one rank - the way to run that path over RCCL on a machine with one GPU
        self.enabled = dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size() > 1 or bool(os.environ.get("CGRNA_DDP_ONE_RANK")))
        self.world = dist.get_world_size() if self.enabled else 1
        self.device = torch.device(device)
        self.is_cuda = self.device.type == "cuda"
        self.bucket_elems = bucket_bytes // 4
        # `stream`: a callable returning the communication stream to use (the engines pass the one their ops object
        # keeps for the life of the process); asked only when there is something to communicate
        self.comm = None
        if self.enabled and self.is_cuda:
            self.comm = stream() if stream is not None else torch.cuda.Stream(device=self.device)
        self.pending = []
        # The gloo backend (CPU clusters and tests; RCCL is the production backend) handed a device tensor stages it
        # through the host on a stream of its own, which it makes wait for the calling stream.  On this stack that wait
        # stalls (seconds to forever) when work is in flight on the device - reproduced without this package by
        # tools/diag_gloo_allreduce.py: all_reduce, a scaling kernel, all_reduce on one stream stalls in the second
        # call - and, with two processes sharing one GPU, now and then even behind a drained stream.  So for gloo the
        # staging is done here: device -> host copy, all_reduce of the host tensor, copy back.
        self.host_staged = self.enabled and self.is_cuda and dist.get_backend() == "gloo"
        # RCCL averages inside the collective (every rank's addend is scaled by 1/world before the sum).  For a power
        # of two that scaling is exact, so the result has the bits of sum-then-scale and the scaling launch per bucket
        # (a read and a write of every gradient byte) goes away; any other world size, and gloo, keep sum-then-scale.
        self.avg_in_collective = (self.enabled and self.is_cuda and dist.get_backend() == "nccl"
                                  and self.world & (self.world - 1) == 0
                                  and not os.environ.get("CGRNA_DDP_SUM_THEN_SCALE"))

    def launch(self, flat_grad, ranges=None, then=None):
        """Start averaging `flat_grad` (a contiguous 1-D tensor) over the ranks, or only its element ranges
        [(lo, hi), ...], whose producer kernels have been enqueued on the current stream.  Buckets of at most
        bucket_bytes go out on the side stream in the order given; wait() joins them.  `then` (a callable) is run
        behind the buckets with the side stream current - work that needs nothing but the averaged ranges, such
        as their optimizer step, which then runs beside whatever the calling stream goes on with; wait() covers it.
        Returns the event that marks the end of this call's work (None for host tensors, which are done on
        return)."""
        if not self.enabled:
            return None
        if ranges is None:
            ranges = [(0, flat_grad.numel())]
        chunks = [flat_grad[s:min(s + self.bucket_elems, hi)] for lo, hi in ranges if hi > lo
                  for s in range(lo, hi, self.bucket_elems)]
        if not self.is_cuda:   # CPU tensors (gloo tests)
            for chunk in chunks:
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
                chunk.div_(self.world)
            if then is not None:
                then()
            return None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        self.comm.wait_event(ready)
        with torch.cuda.stream(self.comm):
            for chunk in chunks:
                if self.host_staged:
                    host = chunk.to("cpu")                 # waits for the chunk's producers (the side stream is drained)
                    dist.all_reduce(host, op=dist.ReduceOp.SUM)
                    chunk.copy_(host.mul_(1.0 / self.world))
                    continue
                if self.avg_in_collective:
                    dist.all_reduce(chunk, op=dist.ReduceOp.AVG)
                    continue
                # sum + scale: also valid on gloo, which lets the device code path be exercised by two ranks
                # sharing one GPU
                dist.all_reduce(chunk, op=dist.ReduceOp.SUM)
                chunk.mul_(1.0 / self.world)
            if then is not None:
                then()
        done = torch.cuda.Event()
        done.record(self.comm)
        self.pending.append(done)
        return done

    def wait(self):
        """Make the current stream wait for every launched bucket."""
        for ev in self.pending:
            torch.cuda.current_stream(self.device).wait_event(ev)
        self.pending = []

    def sum_scalars(self, t):
        """Sum a small tensor over ranks, in place, on the current stream."""
        if not self.enabled:
            return t
        if self.is_cuda and self.host_staged:
            host = t.to("cpu")
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
