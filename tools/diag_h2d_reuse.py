"""Pageable host-to-device copies into a block the caching allocator has just taken back, while the kernel that read
the block's previous contents may still be queued: tmp = cpu.to(device); dst.copy_(tmp); del tmp; repeat.  Every
destination is checked at the end.  (The pattern of loading a network parameter by parameter.)"""
import sys

import torch

dev = torch.device("cuda", 0)
sizes = [96 * 200, 96, 64 * 96, 64, 32 * 64, 32, 512 * 32, 512, 268 * 32, 200 * 96, 200, 1024 * 212, 1024, 16 * 16, 16]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
busy = torch.empty(4096, 4096, device=dev)
for r in range(rounds):
    src = [torch.full((n,), float(r * 100 + i)) for i, n in enumerate(sizes)]
    dst = [torch.zeros(n, device=dev) for n in sizes]
    busy.normal_()                                     # the stream is busy when the copies start
    (busy @ busy).sum()
    for s, d in zip(src, dst):
        d.copy_(s.to(dev))
    for i, (s, d) in enumerate(zip(src, dst)):
        if not torch.equal(d.cpu(), s):
            bad += 1
            w = (d.cpu() != s).nonzero()
            print(f"round {r} tensor {i} ({sizes[i]} floats): {len(w)} wrong elements, first {int(w[0])}: "
                  f"{float(d[int(w[0])])} instead of {float(s[0])}", flush=True)
print(f"{rounds} rounds x {len(sizes)} tensors: {bad} wrong copies", flush=True)
