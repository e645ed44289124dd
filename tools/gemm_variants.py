"""Build copies of libcgrna.so whose GEMM is compiled with different -D options and time them on fixed
(shape, tile, split-K) cases.  Tuning aid.

    python tools/gemm_variants.py build name=-DX=1,-DY=2 ...     (in the build container)
    python tools/gemm_variants.py run name ...                   (on the GPU box; one subprocess per variant)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cycle-gan-rna_amd")
VAR = os.path.join(PKG, "lib", "variants")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off"]

CASES = [  # M, N, K, tA, tB, acc, [(tile, splits)...]   tile 0=128x128 1=128x64 2=64x64
    (512, 2048, 5000, 0, 1, 0, [(2, 1), (1, 2), (0, 4)]),
    (1536, 1024, 5000, 0, 1, 0, [(2, 1), (1, 1), (0, 3)]),
    (1024, 5000, 1024, 1, 0, 1, [(2, 1), (1, 1), (0, 1)]),
    (512, 1024, 2048, 0, 1, 0, [(2, 1), (2, 2), (1, 4)]),
    (512, 512, 1024, 0, 1, 0, [(2, 1), (2, 4)]),
    (512, 5000, 1024, 0, 0, 0, [(2, 1), (1, 1)]),
    (512, 256, 512, 0, 1, 0, [(2, 1), (2, 4)]),
    (512, 1024, 512, 0, 0, 0, [(2, 1), (2, 2)]),
    (4096, 4096, 4096, 0, 1, 0, [(0, 1), (1, 1), (2, 1)]),
]


if os.environ.get("CGRNA_VARIANT_CASES"):      # "M,N,K,ta,tb,acc:tile.splits+tile.splits;..."
    CASES = []
    for item in os.environ["CGRNA_VARIANT_CASES"].split(";"):
        dims, _, forced = item.partition(":")
        CASES.append((*[int(v) for v in dims.split(",")], [tuple(int(x) for x in f.split(".")) for f in forced.split("+")]))


def build(specs):
    os.makedirs(VAR, exist_ok=True)
    objs = [os.path.join(PKG, "lib", f) for f in os.listdir(os.path.join(PKG, "lib")) if f.endswith(".o") and f != "gemm.o"]
    procs = []
    for spec in specs:
        name, _, defs = spec.partition("=")
        obj = os.path.join(VAR, f"gemm_{name}.o")
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *[d for d in defs.split(",") if d], "-c", os.path.join(PKG, "csrc", "gemm.hip"), "-o", obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        assert p.wait() == 0, name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o",
                               os.path.join(VAR, f"libcgrna_{name}.so"), obj, *objs])
        os.remove(obj)
        print("built", name)


def child(dtype):
    sys.path.insert(0, ROOT)
    import torch
    from cycle_gan_rna_amd.ops import HipOps
    ops = HipOps("cuda:0", precision=dtype)
    g = torch.Generator(device="cuda").manual_seed(0)
    for M, N, K, ta, tb, acc, forced in CASES:
        A = torch.rand((K, M) if ta else (M, K), device="cuda", generator=g) * 2 - 1
        B = torch.rand((N, K) if tb else (K, N), device="cuda", generator=g) * 2 - 1
        C = torch.zeros(M, N, device="cuda")
        out = []
        for t, sp in forced:
            os.environ["CGRNA_GEMM_TUNING"] = "1"; os.environ["CGRNA_GEMM_FORCE"] = f"{t},{sp}"
            for _ in range(3):
                ops.gemm(C, A, B, bool(ta), bool(tb), accumulate=bool(acc))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(C, A, B, bool(ta), bool(tb), accumulate=bool(acc))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50.0
            out.append(f"t{t}s{sp} {us:7.1f}us {2.0 * M * N * K / us / 1e6:6.1f}TF")
        print(f"{M:5d}x{N:5d}x{K:5d} {ta}{tb}{acc} | " + " | ".join(out), flush=True)


def run(names, dtype):
    for name in names:
        env = dict(os.environ)
        if name != "default":
            env["CGRNA_LIB"] = os.path.join(VAR, f"libcgrna_{name}.so")
        print(f"== {name} ({dtype})", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", dtype], env=env, check=True, timeout=300)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        dtype = "bf16" if "--bf16" in sys.argv else "fp32"
        run([a for a in sys.argv[2:] if not a.startswith("--")], dtype)
