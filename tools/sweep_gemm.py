"""Sweep tile/split-K choices of cg_gemm (CGRNA_GEMM_FORCE) on the step's large shapes against the cost model's pick."""
import os
os.environ.setdefault("CGRNA_GEMM_TUNING", "1")   # cg_gemm looks at its tuning hooks only when this is set at its first call, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cycle_gan_rna_amd.ops import HipOps

SHAPES = [  # name, M, N, K, transA, transB, accumulate
    ("D0 fwd x3", 1536, 1024, 5000, 0, 1, 0), ("D0 fwd x2", 1024, 1024, 5000, 0, 1, 0), ("enc0 fwd", 512, 2048, 5000, 0, 1, 0),
    ("D0 wgrad K1024", 1024, 5000, 1024, 1, 0, 1), ("D0 wgrad K512", 1024, 5000, 512, 1, 0, 1), ("enc0 wgrad", 2048, 5000, 512, 1, 0, 1),
    ("D0 dgrad", 512, 5000, 1024, 0, 0, 0), ("gp w", 512, 1024, 5000, 0, 1, 0), ("enc0 dgrad", 512, 5000, 2048, 0, 0, 0),
    ("dec3 fwd", 512, 5000, 512, 0, 1, 0), ("enc1 fwd", 512, 1024, 2048, 0, 1, 0), ("D1 fwd x3", 1536, 512, 1024, 0, 1, 0),
    ("D2 fwd x3", 1536, 256, 512, 0, 1, 0), ("D1 wgrad", 512, 1024, 1024, 1, 0, 1), ("D1 dgrad", 1024, 1024, 512, 0, 0, 0),
    ("dec3 wgrad", 5000, 512, 512, 1, 0, 1), ("dec3 dgrad", 512, 512, 5000, 0, 0, 0),
    ("sq1024", 1024, 1024, 1024, 0, 1, 0), ("sq2048", 2048, 2048, 2048, 0, 1, 0), ("sq4096", 4096, 4096, 4096, 0, 1, 0),
    ("tall", 8192, 1024, 512, 0, 1, 0), ("G dec2", 512, 1024, 512, 0, 1, 0), ("G enc2", 512, 512, 1024, 0, 1, 0),
    ("D1 wgrad K1024", 512, 1024, 1024, 1, 0, 1), ("D2 wgrad", 256, 512, 1024, 1, 0, 1), ("mu", 512, 256, 512, 0, 1, 0),
]


_STREAM = None


def timeit(ops, C, A, B, ta, tb, acc, n=20):
    """Microseconds per launch with the launches replayed from a HIP graph (no host time between kernels)."""
    global _STREAM
    if _STREAM is None:
        _STREAM = torch.cuda.Stream()
        ops.prepare_stream(_STREAM)
    with torch.cuda.stream(_STREAM):
        for _ in range(2):
            ops.gemm(C, A, B, bool(ta), bool(tb), accumulate=bool(acc))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)      # see EngineBase._graphed
    with torch.cuda.graph(g, stream=_STREAM):
        for _ in range(n):
            ops.gemm(C, A, B, bool(ta), bool(tb), accumulate=bool(acc))
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def shapes_from_table(path):
    """Shapes printed by `bench.py --gemm-shapes` (M N K tA tB acc dt | ...)."""
    out = []
    for line in open(path):
        f = line.replace("|", " ").split()
        if len(f) >= 7 and f[0].isdigit():
            M, N, K, ta, tb, acc = (int(v) for v in f[:6])
            if min(M, N) > 2 and K > 16:
                out.append((f"{M}x{N}x{K}", M, N, K, ta, tb, acc))
    return out


def main():
    global SHAPES
    args = [a for a in sys.argv[1:] if not a.startswith("--shapes=")]
    for a in sys.argv[1:]:
        if a.startswith("--shapes="):
            SHAPES = shapes_from_table(a.split("=", 1)[1])
    ops = HipOps("cuda:0", precision=args[0] if args else "fp32")
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, M, N, K, ta, tb, acc in SHAPES:
        A = torch.rand((K, M) if ta else (M, K), device="cuda", generator=g) * 2 - 1
        B = torch.rand((N, K) if tb else (K, N), device="cuda", generator=g) * 2 - 1
        C = torch.zeros(M, N, device="cuda")
        os.environ.pop("CGRNA_GEMM_FORCE", None)
        base = timeit(ops, C, A, B, ta, tb, acc)
        res = []
        for t in range(3):
            for sp in (1, 2, 3, 4, 6, 8, 12, 16):
                os.environ["CGRNA_GEMM_FORCE"] = f"{t},{sp}"
                res.append((timeit(ops, C, A, B, ta, tb, acc, n=20), t, sp))
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "sweep_all%s.csv" % ("_bf16" if ops.precision == "bf16" else "")), "a") as f:
            for us, t, sp in res:
                f.write(f"{name},{M},{N},{K},{ta},{tb},{acc},{t},{sp},{us:.2f}\n")
        res.sort()
        fl = 2.0 * M * N * K / 1e6
        print(f"{name:16s} {M:5d}x{N:5d}x{K:5d} model {base:7.1f} us {fl / base:6.1f} TF | best " +
              "  ".join(f"t{t} s{sp}: {us:6.1f}us {fl / us:5.1f}TF" for us, t, sp in res[:4]), flush=True)


if __name__ == "__main__":
    main()
