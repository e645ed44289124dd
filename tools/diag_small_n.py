"""Launch the 3..16-column product (M x 12 x 1024, op(B) a column slice of a wide matrix) a number of times, with
contiguous B and with B = W[:, G:] of a (1024, G + 12) matrix, cold (a 256 MB fill between launches evicts the caches)
and warm.  Meant to be run under `rocprofv3 --kernel-trace --stats` and read from the kernel's durations there: the
HIP-event figures it prints include the host's launch latency whenever the GPU is idle when the launch arrives (about
30 us through ctypes), which for a kernel of a few microseconds is most of the number.   usage: diag_small_n.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cycle_gan_rna_amd.ops import HipOps  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ops = HipOps("cuda:0")
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.rand(M, 1024, device="cuda", generator=g)
W = torch.rand(1024, 5012, device="cuda", generator=g)
Bc = W[:, 5000:].contiguous()
C = torch.zeros(M, 12, device="cuda")
junk = torch.empty(64 << 20, device="cuda")


def timed(B, cold, reps=20):
    tot = 0.0
    for _ in range(reps):
        if cold:
            junk.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(C, A, B, False, False)
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


for name, B in (("contiguous", Bc), ("slice ld=5012", W[:, 5000:])):
    timed(B, False, 3)
    print(f"M={M} B {name}: warm {timed(B, False):.1f} us, cold {timed(B, True):.1f} us")
