"""Run one cg_gemm shape a number of times (for rocprofv3 passes).  usage: gemm_one.py M N K tA tB acc tile splits [reps] [dtype]
tile < 0: the library's own choice."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cycle_gan_rna_amd.ops import HipOps  # noqa: E402

M, N, K, ta, tb, acc, tile, splits = (int(v) for v in sys.argv[1:9])
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
ops = HipOps("cuda:0", precision=sys.argv[10] if len(sys.argv) > 10 else "fp32")
if tile >= 0:
    os.environ["CGRNA_GEMM_TUNING"] = "1"
    os.environ["CGRNA_GEMM_FORCE"] = f"{tile},{splits}"
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.rand((K, M) if ta else (M, K), device="cuda", generator=g) * 2 - 1
B = torch.rand((N, K) if tb else (K, N), device="cuda", generator=g) * 2 - 1
C = torch.zeros(M, N, device="cuda")
for _ in range(reps):
    ops.gemm(C, A, B, bool(ta), bool(tb), accumulate=bool(acc))
torch.cuda.synchronize()
print("done", float(C[0, 0]))
