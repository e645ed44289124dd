"""Micro-benchmark of cg_gemm on the shapes of the B=512, G=5000 step (HIP events, random data)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cycle_gan_rna_amd.ops import HipOps

SHAPES = [  # name, M, N, K, transA, transB
    ("enc0 fwd NT", 512, 2048, 5000, 0, 1), ("enc1 fwd NT", 512, 1024, 2048, 0, 1), ("enc2 fwd NT", 512, 512, 1024, 0, 1),
    ("dec3 fwd NT", 512, 5000, 512, 0, 1), ("D0 fwd NT", 512, 1024, 5012, 0, 1), ("D1 fwd NT", 512, 512, 1024, 0, 1),
    ("dec3 dgrad NN", 512, 512, 5000, 0, 0), ("enc0 dgrad NN", 512, 5000, 2048, 0, 0), ("D0 dgrad NN", 512, 5000, 1024, 0, 0),
    ("enc1 dgrad NN", 512, 2048, 1024, 0, 0),
    ("enc0 wgrad TN", 2048, 5000, 512, 1, 0), ("D0 wgrad TN", 1024, 5000, 512, 1, 0), ("enc1 wgrad TN", 1024, 2048, 512, 1, 0),
    ("dec3 wgrad TN", 5000, 512, 512, 1, 0), ("enc2 wgrad TN", 512, 1024, 512, 1, 0),
    ("square 4096", 4096, 4096, 4096, 0, 1),
]

def main():
    ops = HipOps("cuda:0", precision=sys.argv[1] if len(sys.argv) > 1 else "fp32")
    g = torch.Generator(device="cuda").manual_seed(0)
    tot_t = tot_f = 0.0
    for name, M, N, K, ta, tb in SHAPES:
        A = torch.rand((K, M) if ta else (M, K), device="cuda", generator=g) * 2 - 1
        B = torch.rand((N, K) if tb else (K, N), device="cuda", generator=g) * 2 - 1
        C = torch.zeros(M, N, device="cuda")
        for _ in range(3):
            ops.gemm(C, A, B, bool(ta), bool(tb))
        torch.cuda.synchronize()
        n = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.gemm(C, A, B, bool(ta), bool(tb))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        tf = 2.0 * M * N * K / us / 1e6
        if "square" not in name:
            tot_t += us; tot_f += 2.0 * M * N * K
        print(f"{name:16s} M={M:5d} N={N:5d} K={K:5d}  {us:8.1f} us  {tf:7.1f} TF/s")
    print(f"step shapes aggregate: {tot_f / tot_t / 1e6:.1f} TF/s")

if __name__ == "__main__":
    main()
