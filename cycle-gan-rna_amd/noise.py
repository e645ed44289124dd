"""Sources of the step's random inputs (reparameterisation eps, dropout keep
masks, gradient-penalty alpha).

The reference draws them implicitly from torch's global generator, in the
order its modules are called (eps then four attention masks per generator
call; two dropout masks per critic call; alpha before the penalty's critic
call).  The engine asks for them in that same order, so:

  HostReplayNoise  draws on the CPU with the reference's exact calls (shape,
                   dtype, order) and uploads - after the same torch.manual_seed
                   it sees the numbers a CPU run of the reference sees (parity
                   mode; costs a host->device copy per draw)
  DeviceNoise      draws in HBM from a torch CUDA generator (training mode)
"""
from __future__ import annotations

import torch


class HostReplayNoise:
    def __init__(self, device, dtype=torch.float32):
        self.device, self.dtype = device, dtype

    def randn(self, shape):
        return torch.randn(shape, dtype=self.dtype).to(self.device)

    def keep(self, shape, p_drop):
        return torch.empty(shape, dtype=self.dtype).bernoulli_(1.0 - p_drop).div_(1.0 - p_drop).to(self.device)

    def rand(self, shape):
        return torch.rand(shape).to(self.dtype).to(self.device)


class RecordedNoise:
    """Replays tensors recorded elsewhere (golden fixtures), in draw order."""

    def __init__(self, tensors, device, dtype=torch.float32):
        self.tensors = [t.to(dtype).to(device).contiguous() for t in tensors]
        self.pos = 0

    def _next(self, shape):
        t = self.tensors[self.pos]
        self.pos += 1
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t

    def randn(self, shape):
        return self._next(shape)

    def keep(self, shape, p_drop):
        return self._next(shape)

    def rand(self, shape):
        return self._next(shape)


class DeviceNoise:
    """Draws straight into HBM.  Buffers are reused per (kind, shape, call index
    within a step) so that a captured step replays with fresh numbers."""

    stacked_draws = True     # one draw serves several stacked calls of a module (no reference draw order to follow)

    def __init__(self, device, seed=0, default_generator=False):
        """default_generator=True draws from torch's default CUDA generator (seeded
        here), which HIP-graph capture supports; otherwise a private generator is used."""
        self.device = torch.device(device)
        if default_generator:
            self.gen = None
            with torch.cuda.device(self.device):
                torch.cuda.manual_seed(seed)
        else:
            self.gen = torch.Generator(device=self.device)
            self.gen.manual_seed(seed)
        self.bufs = {}
        self.count = {}
        self.space = ""      # buffer namespace: work issued on another lane draws into buffers of its own

    def begin_step(self):
        self.count = {}

    def _buf(self, kind, shape):
        key = (self.space, kind, tuple(shape))
        i = self.count.get(key, 0)
        self.count[key] = i + 1
        lst = self.bufs.setdefault(key, [])
        if i >= len(lst):
            lst.append(torch.empty(shape, dtype=torch.float32, device=self.device))
        return lst[i]

    def randn(self, shape):
        return self._buf("n", shape).normal_(generator=self.gen)

    def keep(self, shape, p_drop):
        return self._buf("k", shape).bernoulli_(1.0 - p_drop, generator=self.gen).div_(1.0 - p_drop)

    def rand(self, shape):
        return self._buf("u", shape).uniform_(generator=self.gen)
