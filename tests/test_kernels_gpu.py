"""Per-kernel parity on the MI355X: every C-ABI entry point against its CPU
restatement (oracle/ops_cpu.py) on the same seeded inputs.

Tolerances are for float32 arithmetic with different summation orders:
rtol 2e-5 relative to the largest magnitude of the expected tensor, unless a
test states otherwise.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.ops_cpu import CpuOps, EPI_LEAKY, EPI_MUL, EPI_NONE, ATTN_NPARAM  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    from cycle_gan_rna_amd.ops import HipOps
    return HipOps("cuda:0")


cpu = CpuOps()


def close(got, exp, rtol=2e-5, what=""):
    got = got.detach().cpu().double()
    exp = exp.detach().double()
    scale = max(float(exp.abs().max()), 1e-30)
    err = float((got - exp).abs().max()) / scale
    assert err <= rtol, f"{what}: rel err {err:.3e} > {rtol}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def dev(t):
    return t.cuda() if t is not None else None


GEMM_SHAPES = [
    # M, N, K, transA, transB
    (64, 64, 32, 0, 1), (128, 128, 64, 0, 1), (33, 70, 45, 0, 1), (512, 2048, 300, 0, 1), (512, 1024, 5012, 0, 1),
    (100, 5000, 512, 0, 1), (8, 1, 256, 0, 1), (257, 129, 31, 0, 0), (512, 268, 2048, 0, 0), (64, 200, 1, 0, 0),
    (1024, 212, 32, 1, 0), (96, 48, 515, 1, 0), (1, 256, 37, 1, 0), (16, 16, 8192, 1, 0), (70, 33, 19, 1, 1),
    (1536, 1024, 512, 0, 1),
]


@pytest.mark.parametrize("M,N,K,ta,tb", GEMM_SHAPES)
def test_gemm_plain(ops, M, N, K, ta, tb):
    A = rnd(*((K, M) if ta else (M, K)), seed=1)
    B = rnd(*((N, K) if tb else (K, N)), seed=2)
    bias = rnd(N, seed=3)
    exp = torch.empty(M, N)
    cpu.gemm(exp, A.double(), B.double(), ta, tb, bias=bias.double())
    got = torch.full((M, N), float("nan")).cuda()
    ops.gemm(got, dev(A), dev(B), ta, tb, bias=dev(bias))
    close(got, exp, 2e-5, "gemm")


@pytest.mark.parametrize("M,N,K,ta,tb", [(128, 128, 256, 0, 1), (512, 2048, 5000, 0, 1), (512, 5000, 1024, 0, 0),
                                         (1024, 5012, 512, 1, 0), (130, 300, 517, 0, 0), (257, 129, 263, 1, 0),
                                         (200, 136, 300, 1, 1), (512, 1024, 2048, 0, 1)])
def test_gemm_bf16(ops, M, N, K, ta, tb):
    """bf16 MFMA path: operands rounded to bf16 on the fly, float32 accumulation.  The expectation is the
    float64 product of the bf16-rounded operands, so only the accumulation order differs (rtol 2e-5)."""
    A = rnd(*((K, M) if ta else (M, K)), seed=1)
    B = rnd(*((N, K) if tb else (K, N)), seed=2)
    bias = rnd(N, seed=3)
    exp = torch.empty(M, N, dtype=torch.float64)
    cpu.gemm(exp, A.bfloat16().double(), B.bfloat16().double(), ta, tb, bias=bias.double())
    got = torch.full((M, N), float("nan")).cuda()
    ops.gemm(got, dev(A), dev(B), ta, tb, bias=dev(bias), dtype=1)
    close(got, exp, 2e-5, "gemm bf16")
    # and against the unrounded product: bf16 has 8 significand bits
    exact = torch.empty(M, N, dtype=torch.float64)
    cpu.gemm(exact, A.double(), B.double(), ta, tb, bias=bias.double())
    close(got, exact, 2e-2, "gemm bf16 vs float64")


@pytest.mark.parametrize("M,N,K,ta,tb,lda_pad,ldb_pad,flags", [
    (512, 1024, 5012, 0, 1, 0, 0, 2),      # weights [N][K] stored bf16, rows 8-byte aligned only (5012 * 2 bytes)
    (512, 1024, 5000, 0, 1, 0, 12, 2),     # a column slice of a wider bf16 matrix
    (512, 5000, 1024, 0, 0, 0, 12, 2),     # input gradient: weights [K][N] read along the rows, N not a multiple of 8
    (130, 268, 512, 0, 0, 0, 0, 2), (96, 160, 200, 1, 0, 0, 0, 2),
    (512, 2048, 5000, 0, 1, 0, 0, 6),      # both operands stored bf16
    (512, 1024, 300, 1, 0, 4, 0, 4),       # A stored bf16 and read along its rows
    (200, 136, 1030, 0, 1, 2, 6, 6)])
def test_gemm_bf16_operands_stored_as_bf16(ops, M, N, K, ta, tb, lda_pad, ldb_pad, flags):
    """CG_BF16 | CG_B_IS_BF16 / CG_A_IS_BF16 through the C ABI: an operand that is already bf16 in memory gives the
    same result, bit for bit, as the float32 copy of the same (bf16-representable) values rounded while staged."""
    import ctypes as C
    from cycle_gan_rna_amd import _lib
    lib = _lib.load_ctypes()
    ar, ac = (K, M) if ta else (M, K)
    br, bc = (N, K) if tb else (K, N)
    A = rnd(ar, ac + lda_pad, seed=1).bfloat16().float()
    B = rnd(br, bc + ldb_pad, seed=2).bfloat16().float()
    Ad, Bd = dev(A), dev(B)
    A16, B16 = Ad.bfloat16(), Bd.bfloat16()
    ref = torch.full((M, N), float("nan")).cuda()
    ops.gemm(ref, Ad[:, :ac], Bd[:, :bc], bool(ta), bool(tb), dtype=1)
    got = torch.full((M, N), float("nan")).cuda()
    ws = torch.empty(8 << 20, device="cuda")
    pa = (A16 if flags & 4 else Ad).data_ptr()
    pb = (B16 if flags & 2 else Bd).data_ptr()
    st = lib.cg_gemm(C.c_void_p(got.data_ptr()), N, C.c_void_p(pa), ac + lda_pad, C.c_void_p(pb), bc + ldb_pad, M, N, K,
                     ta, tb, 1 | flags, None, 0, None, 0, None, 0, 0.0, 0, C.c_void_p(ws.data_ptr()), ws.numel() * 4,
                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, st
    torch.cuda.synchronize()
    exp = torch.empty(M, N, dtype=torch.float64)
    cpu.gemm(exp, A[:, :ac].double(), B[:, :bc].double(), bool(ta), bool(tb))
    close(got, exp, 2e-5, "gemm with bf16-stored operands")
    assert torch.equal(got, ref)
    # an odd contiguous extent or a misaligned row start is refused, not mis-read
    st = lib.cg_gemm(C.c_void_p(got.data_ptr()), N, C.c_void_p(pa), ac + lda_pad, C.c_void_p(pb + 2), bc + ldb_pad, M, N, K,
                     ta, tb, 3, None, 0, None, 0, None, 0, 0.0, 0, None, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == -2


def test_gemm_bf16_exact_on_small_integers(ops):
    M, N, K = 96, 160, 200
    A = torch.randint(-3, 4, (K, M), generator=torch.Generator().manual_seed(0)).float()
    B = (torch.arange(K * N).reshape(K, N) % 7 - 3).float()
    got = torch.zeros(M, N).cuda()
    ops.gemm(got, dev(A), dev(B), True, False, dtype=1)
    assert torch.equal(got.cpu(), A.t() @ B)


def test_gemm_is_exact_fp32_fma_chain(ops):
    """Small-integer data: every product and sum is exact, so any slip in the k
    coverage or in the lane -> (row, col) map shows as an integer difference.
    B is asymmetric so a transposed write cannot hide."""
    M, N, K = 96, 160, 77
    A = torch.randint(-3, 4, (M, K), generator=torch.Generator().manual_seed(0)).float()
    B = (torch.arange(N * K).reshape(N, K) % 7 - 3).float()
    got = torch.zeros(M, N).cuda()
    ops.gemm(got, dev(A), dev(B), False, True)
    assert torch.equal(got.cpu(), A @ B.t())


def test_gemm_strided_views_and_offsets(ops):
    """Operands that are column slices of wider matrices (the critic's first
    layer reads W1[:, :G] and W1[:, G:], lda != K) and an unaligned base."""
    M, G, E, N = 40, 50, 12, 96
    W = rnd(N, G + E, seed=5)
    x = rnd(M, G, seed=6)
    e = rnd(M, E, seed=7)
    Wd = dev(W)
    got = torch.zeros(M, N).cuda()
    ops.gemm(got, dev(x), Wd[:, :G], False, True)
    ops.gemm(got, dev(e), Wd[:, G:], False, True, accumulate=True)
    close(got, torch.cat([x, e], 1).double() @ W.double().t(), 2e-5, "split-K-by-column gemm")
    # wgrad into a column slice of a wider gradient matrix
    dh = rnd(M, N, seed=8)
    dW = torch.zeros(N, G + E).cuda()
    ops.gemm(dW[:, :G], dev(dh), dev(x), True, False, accumulate=True)
    ops.gemm(dW[:, G:], dev(dh), dev(e), True, False, accumulate=True)
    close(dW, dh.double().t() @ torch.cat([x, e], 1).double(), 2e-5, "wgrad into slices")


@pytest.mark.parametrize("epi", [EPI_LEAKY, EPI_MUL])
def test_gemm_epilogues(ops, epi):
    M, N, K = 130, 200, 96
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3)
    E = (torch.rand(M, N, generator=torch.Generator().manual_seed(4)) < 0.7).float() / 0.7
    exp, aux = torch.empty(M, N, dtype=torch.float64), torch.empty(M, N, dtype=torch.float64)
    cpu.gemm(exp, A.double(), B.double(), False, True, bias=bias.double(), epi=epi, E=E.double(), aux=aux, slope=0.2)
    got, gaux = torch.zeros(M, N).cuda(), torch.zeros(M, N).cuda()
    ops.gemm(got, dev(A), dev(B), False, True, bias=dev(bias), epi=epi, E=dev(E), aux=gaux, slope=0.2)
    close(got, exp, 2e-5, "epilogue C")
    if epi == EPI_LEAKY:
        # pm can legitimately flip where |h| is at rounding level; require agreement elsewhere
        h = A.double() @ B.double().t() + bias.double()
        ok = h.abs() > 1e-4
        assert torch.allclose(gaux.cpu().double()[ok], aux[ok], rtol=1e-6, atol=0)


def test_gemm_rejects_bad_arguments(ops):
    from cycle_gan_rna_amd._lib import CgError
    a = torch.zeros(4, 8).cuda()
    c = torch.zeros(4, 4).cuda()
    with pytest.raises(AssertionError):
        ops.gemm(c, a, torch.zeros(5, 9).cuda(), False, True)
    with pytest.raises(CgError):
        ops.gemm(c, a, torch.zeros(4, 8).cuda(), False, True, epi=EPI_MUL)   # E missing


@pytest.mark.parametrize("M,N,slope,use_mask", [(512, 2048, 0.0, False), (37, 100, 0.2, True), (512, 512, 0.2, True), (300, 1000, 0.0, False), (129, 256, 1.0, True), (8192, 16, 1.0, False),
                                                (5, 33, 0.0, False), (128, 1024, 0.2, True)])
def test_bn_fwd_bwd(ops, M, N, slope, use_mask):
    X = rnd(M, N, seed=1) * 2 + 0.7
    gamma, beta = rnd(N, seed=2) + 1.5, rnd(N, seed=3)
    mask = ((torch.rand(M, N, generator=torch.Generator().manual_seed(4)) < 0.7).float() / 0.7) if use_mask else None
    rm, rv = rnd(N, seed=5), rnd(N, seed=6).abs() + 0.5
    # expected (float64 inside CpuOps)
    c64 = CpuOps(torch.float64)
    d = lambda t: None if t is None else t.double()  # noqa: E731
    eY, epm = torch.empty(M, N, dtype=torch.float64), torch.empty(M, N, dtype=torch.float64)
    emean, einv = torch.empty(N, dtype=torch.float64), torch.empty(N, dtype=torch.float64)
    erm, erv = rm.double().clone(), rv.double().clone()
    c64.bn_fwd(eY, d(X), d(gamma), d(beta), emean, einv, erm, erv, slope, d(mask), epm)
    gY, gpm = torch.empty(M, N).cuda(), torch.empty(M, N).cuda()
    gmean, ginv, grm, grv = torch.empty(N).cuda(), torch.empty(N).cuda(), dev(rm.clone()), dev(rv.clone())
    Xd = dev(X)
    ops.bn_fwd(gY, Xd, dev(gamma), dev(beta), gmean, ginv, grm, grv, slope, dev(mask), gpm)
    for g, e, w in [(gY, eY, "Y"), (gmean, emean, "mean"), (ginv, einv, "invstd"), (grm, erm, "rmean"), (grv, erv, "rvar")]:
        close(g, e, 3e-5, f"bn_fwd {w}")
    # backward, with and without parameter gradients / saved pm
    dY = rnd(M, N, seed=7)
    edX = torch.empty(M, N, dtype=torch.float64)
    edg, edb = torch.ones(N, dtype=torch.float64), torch.ones(N, dtype=torch.float64)
    egb, egx = torch.empty(N, dtype=torch.float64), torch.empty(N, dtype=torch.float64)
    c64.bn_bwd(edX, d(dY), d(X), eY if not use_mask else None, epm if use_mask else None, emean, einv, d(gamma), edg, edb,
               True, slope, egb, egx)
    gdX, gdg, gdb = torch.empty(M, N).cuda(), torch.ones(N).cuda(), torch.ones(N).cuda()
    ggb, ggx = torch.empty(N).cuda(), torch.empty(N).cuda()
    ops.bn_bwd(gdX, dev(dY), Xd, gY if not use_mask else None, gpm if use_mask else None, gmean, ginv, dev(gamma), gdg,
               gdb, True, slope, ggb, ggx)
    for g, e, w in [(gdX, edX, "dX"), (gdg, edg, "dgamma"), (gdb, edb, "dbeta"), (ggb, egb, "gbar"), (ggx, egx, "gxbar")]:
        close(g, e, 1e-4, f"bn_bwd {w}")


@pytest.mark.parametrize("b,nseg,order,N", [(512, 3, (0, 1, 2), 1024), (512, 3, (0, 2, 1), 1024), (37, 2, (1, 0), 100),
                                            (500, 3, (0, 2, 1), 1024)])
def test_bn_segments(ops, b, nseg, order, N):
    """Several batches normalised in one launch equal separate calls, running statistics updated in the given order."""
    M = b * nseg
    X = rnd(M, N, seed=1) * 2 + 0.7 + torch.arange(M)[:, None] / M            # segments have different statistics
    gamma, beta = rnd(N, seed=2) + 1.5, rnd(N, seed=3)
    mask = (torch.rand(M, N, generator=torch.Generator().manual_seed(4)) < 0.7).float() / 0.7
    rm, rv = rnd(N, seed=5), rnd(N, seed=6).abs() + 0.5
    c64 = CpuOps(torch.float64)
    d = lambda t: None if t is None else t.double()  # noqa: E731
    eY, epm = torch.empty(M, N, dtype=torch.float64), torch.empty(M, N, dtype=torch.float64)
    emean, einv = torch.empty(nseg, N, dtype=torch.float64), torch.empty(nseg, N, dtype=torch.float64)
    erm, erv = rm.double().clone(), rv.double().clone()
    c64.bn_fwd_seg(eY, d(X), nseg, order, d(gamma), d(beta), emean, einv, erm, erv, 0.2, d(mask), epm)
    gY, gpm = torch.empty(M, N).cuda(), torch.empty(M, N).cuda()
    gmean, ginv, grm, grv = torch.empty(nseg, N).cuda(), torch.empty(nseg, N).cuda(), dev(rm.clone()), dev(rv.clone())
    Xd = dev(X)
    ops.bn_fwd_seg(gY, Xd, nseg, order, dev(gamma), dev(beta), gmean, ginv, grm, grv, 0.2, dev(mask), gpm)
    for g, e, w in [(gY, eY, "Y"), (gmean, emean, "mean"), (ginv, einv, "invstd"), (grm, erm, "rmean"), (grv, erv, "rvar")]:
        close(g, e, 3e-5, f"bn_fwd_seg {w}")
    # one launch == separate launches, bit for bit (same kernels, same geometry per segment)
    sY, spm = torch.empty(M, N).cuda(), torch.empty(M, N).cuda()
    smean, sinv, srm, srv = torch.empty(nseg, N).cuda(), torch.empty(nseg, N).cuda(), dev(rm.clone()), dev(rv.clone())
    for sg in order:
        sl = slice(sg * b, (sg + 1) * b)
        ops.bn_fwd(sY[sl], Xd[sl], dev(gamma), dev(beta), smean[sg], sinv[sg], srm, srv, 0.2, dev(mask)[sl], spm[sl])
    assert torch.equal(sY, gY) and torch.equal(srm, grm) and torch.equal(srv, grv) and torch.equal(smean, gmean)
    dY = rnd(M, N, seed=7)
    edX = torch.empty(M, N, dtype=torch.float64)
    edg, edb = torch.ones(N, dtype=torch.float64), torch.ones(N, dtype=torch.float64)
    c64.bn_bwd_seg(edX, d(dY), d(X), None, epm, nseg, emean, einv, d(gamma), edg, edb, True, 0.2)
    gdX, gdg, gdb = torch.empty(M, N).cuda(), torch.ones(N).cuda(), torch.ones(N).cuda()
    ops.bn_bwd_seg(gdX, dev(dY), Xd, None, gpm, nseg, gmean, ginv, dev(gamma), gdg, gdb, True, 0.2)
    for g, e, w in [(gdX, edX, "dX"), (gdg, edg, "dgamma"), (gdb, edb, "dbeta")]:
        close(g, e, 1e-4, f"bn_bwd_seg {w}")


@pytest.mark.parametrize("n_tok", [8192, 16 * 37, 128, 48])
def test_token_residual_block(ops, n_tok):
    """Fused latent ResidualBlock kernels (forward and backward) against their float64 CPU restatement."""
    c64 = CpuOps(torch.float64)
    d = lambda t: t.double()  # noqa: E731
    X = rnd(n_tok, 16, seed=1)
    W1, W2 = rnd(16, 16, seed=2) * 0.4, rnd(16, 16, seed=3) * 0.4
    b1, b2 = rnd(16, seed=4), rnd(16, seed=5)
    g1, be1, g2, be2 = rnd(16, seed=6) + 1.5, rnd(16, seed=7), rnd(16, seed=8) + 1.5, rnd(16, seed=9)
    rm, rv = rnd(16, seed=10), rnd(16, seed=11).abs() + 0.5
    dY = rnd(n_tok, 16, seed=12)
    nb = ops.tok_blocks(n_tok)

    def run(o, cast, Z, Zd):
        T1, R, T2, Y, G1, dX = (Z(n_tok, 16) for _ in range(6))
        st = [Zd(nb * 32) for _ in range(4)]
        wp2, wp1 = Z(nb * 256), Z(nb * 256)
        m1, i1, m2, i2 = Z(16), Z(16), Z(16), Z(16)
        rm1, rv1, rm2, rv2 = cast(rm).clone(), cast(rv).clone(), cast(rm).clone(), cast(rv).clone()
        dW1, dW2 = Z(16, 16) + 1, Z(16, 16) + 1
        dg1, db1, dg2, db2 = Z(16) + 1, Z(16) + 1, Z(16) + 1, Z(16) + 1
        x, w1, w2 = cast(X), cast(W1), cast(W2)
        o.tok_lin_fwd(T1, st[0], x, w1, cast(b1))
        o.tok_bn_lin_fwd(R, T2, st[1], T1, st[0], cast(g1), cast(be1), m1, i1, rm1, rv1, w2, cast(b2))
        o.tok_bn_res_fwd(Y, T2, st[1], cast(g2), cast(be2), m2, i2, rm2, rv2, x)
        dy = cast(dY)
        o.tok_bn_bwd_stats(st[2], dy, T2, m2, i2)
        o.tok_bwd_mid(G1, st[3], wp2, dg2, db2, dy, T2, R, T1, st[2], m2, i2, cast(g2), w2, m1, i1)
        o.tok_bwd_in(dX, wp1, dg1, db1, G1, T1, x, dy, st[3], m1, i1, cast(g1), w1)
        o.tok_wgrad_reduce(dW2, dW1, wp2, wp1, n_tok)
        return dict(T1=T1, R=R, T2=T2, Y=Y, m1=m1, i1=i1, m2=m2, i2=i2, rm1=rm1, rv1=rv1, rm2=rm2, rv2=rv2, G1=G1, dX=dX,
                    dW1=dW1, dW2=dW2, dg1=dg1, db1=db1, dg2=dg2, db2=db2)
    exp = run(c64, d, lambda *s: torch.zeros(*s, dtype=torch.float64), lambda *s: torch.zeros(*s, dtype=torch.float64))
    got = run(ops, dev, lambda *s: torch.zeros(*s, device="cuda"), lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda"))
    for k in exp:
        close(got[k], exp[k], 2e-4 if k in ("G1", "dX", "dW1", "dW2", "dg1", "db1", "dg2", "db2") else 3e-5, f"tok {k}")


def test_bn_bwd_matches_autograd():
    """The CPU restatement of the BatchNorm backward itself against torch autograd (float64)."""
    M, N = 24, 10
    X = rnd(M, N, seed=1).double().requires_grad_(True)
    gamma, beta = (rnd(N, seed=2) + 1.5).double().requires_grad_(True), rnd(N, seed=3).double().requires_grad_(True)
    mu = X.mean(0)
    var = ((X - mu) ** 2).mean(0)
    h = (X - mu) / torch.sqrt(var + 1e-5) * gamma + beta
    y = torch.where(h > 0, h, 0.2 * h)
    dY = rnd(M, N, seed=4).double()
    gx, gg, gb = torch.autograd.grad(y, (X, gamma, beta), dY)
    c64 = CpuOps(torch.float64)
    Y, pm, mean, inv = (torch.empty(M, N, dtype=torch.float64), torch.empty(M, N, dtype=torch.float64),
                        torch.empty(N, dtype=torch.float64), torch.empty(N, dtype=torch.float64))
    c64.bn_fwd(Y, X.detach(), gamma.detach(), beta.detach(), mean, inv, None, None, 0.2, None, pm)
    dX, dg, db = torch.empty(M, N, dtype=torch.float64), torch.empty(N, dtype=torch.float64), torch.empty(N, dtype=torch.float64)
    c64.bn_bwd(dX, dY, X.detach(), None, pm, mean, inv, gamma.detach(), dg, db, False, 0.2)
    close(dX, gx, 1e-10), close(dg, gg, 1e-10), close(db, gb, 1e-10)


def test_bn_bwd_bwd(ops):
    M, N = 64, 100
    X, W, U1 = rnd(M, N, seed=1) + 0.3, rnd(M, N, seed=2), rnd(M, N, seed=3)
    gamma = rnd(N, seed=4) + 1.5
    c64 = CpuOps(torch.float64)
    mean, inv = torch.empty(N, dtype=torch.float64), torch.empty(N, dtype=torch.float64)
    c64.bn_fwd(torch.empty(M, N, dtype=torch.float64), X.double(), gamma.double(), torch.zeros(N, dtype=torch.float64),
               mean, inv, None, None, 1.0)
    u0, ub, mb = torch.empty(M, N, dtype=torch.float64), torch.empty(N, dtype=torch.float64), torch.empty(N, dtype=torch.float64)
    c64.bn_bwd(u0, U1.double(), X.double(), None, None, mean, inv, gamma.double(), None, None, False, 1.0, ub, mb)
    eVu, eVh, edg = (torch.empty(M, N, dtype=torch.float64), torch.empty(M, N, dtype=torch.float64),
                     torch.zeros(N, dtype=torch.float64))
    c64.bn_bwd_bwd(eVu, eVh, edg, False, W.double(), U1.double(), X.double(), mean, inv, gamma.double(), ub, mb)
    gVu, gVh, gdg = torch.empty(M, N).cuda(), torch.empty(M, N).cuda(), torch.zeros(N).cuda()
    ops.bn_bwd_bwd(gVu, gVh, gdg, False, dev(W), dev(U1), dev(X), dev(mean.float()), dev(inv.float()), dev(gamma),
                   dev(ub.float()), dev(mb.float()))
    close(gVu, eVu, 1e-4, "V_u1"), close(gVh, eVh, 1e-4, "V_h0"), close(gdg, edg, 1e-4, "dgamma")


def test_bn_bwd_bwd_matches_autograd():
    """CPU restatement of the backward-of-backward against torch double autograd (float64)."""
    M, N = 12, 7
    X = (rnd(M, N, seed=1) + 0.3).double().requires_grad_(True)
    gamma = (rnd(N, seed=4) + 1.5).double().requires_grad_(True)
    U1 = rnd(M, N, seed=3).double().requires_grad_(True)
    W = rnd(M, N, seed=2).double()
    mu = X.mean(0)
    inv = 1 / torch.sqrt(((X - mu) ** 2).mean(0) + 1e-5)
    xh = (X - mu) * inv
    u0 = inv * gamma * (U1 - U1.mean(0) - xh * (U1 * xh).mean(0))      # BatchNorm backward as a function
    gX, gG, gU = torch.autograd.grad(u0, (X, gamma, U1), W)
    c64 = CpuOps(torch.float64)
    Vu, Vh, dg = (torch.empty(M, N, dtype=torch.float64), torch.empty(M, N, dtype=torch.float64),
                  torch.zeros(N, dtype=torch.float64))
    c64.bn_bwd_bwd(Vu, Vh, dg, False, W, U1.detach(), X.detach(), mu.detach(), inv.detach(), gamma.detach(),
                   U1.detach().mean(0), (U1.detach() * xh.detach()).mean(0))
    close(Vu, gU, 1e-10, "V_u1"), close(Vh, gX, 1e-10, "V_h0"), close(dg, gG, 1e-10, "dgamma")


def test_elementwise_and_gathers(ops):
    M, N = 70, 333
    x, y = rnd(M, N, seed=1), rnd(M, N, seed=2)
    out = torch.empty(M, N).cuda()
    ops.axpby(out, 0.5, dev(x), -2.0, dev(y))
    close(out, 0.5 * x - 2 * y, 1e-6)
    ops.mul(out, dev(x), dev(y))
    close(out, x * y, 1e-6)
    coef = torch.tensor([0.0, 3.0, -1.5]).cuda()
    ops.fill_coef(out, coef, 2, 0.5)
    assert torch.equal(out.cpu(), torch.full((M, N), -0.75))
    table = rnd(12, 12, seed=3)
    labels = torch.randint(0, 12, (M, 1), generator=torch.Generator().manual_seed(4))
    wide = torch.zeros(M, 268).cuda()
    ops.embed_gather(wide[:, 256:], dev(table), dev(labels))
    assert torch.equal(wide[:, 256:].cpu(), table[labels[:, 0]])
    dout = rnd(M, 12, seed=5)
    dt = torch.ones(12, 12).cuda()
    ops.embed_scatter(dt, dev(dout), dev(labels), True)
    exp = torch.ones(12, 12, dtype=torch.float64)
    exp.index_add_(0, labels[:, 0], dout.double())
    close(dt, exp, 1e-5)


def test_reparam_libsize_gp(ops):
    M, L = 50, 256
    ml, eps, dz = rnd(M, 2 * L, seed=1) * 0.5, rnd(M, L, seed=2), rnd(M, L, seed=3)
    z, ez = torch.empty(M, L).cuda(), torch.empty(M, L)
    ops.reparam_fwd(z, dev(ml), dev(eps))
    cpu.reparam_fwd(ez, ml, eps)
    close(z, ez, 1e-5)
    coef = torch.tensor([0.1, 0.2, 0.3, 0.4, 2.5])
    dml, edml = torch.empty(M, 2 * L).cuda(), torch.empty(M, 2 * L)
    ops.reparam_bwd(dml, dev(dz), dev(ml), dev(eps), dev(coef), 4, 1.0)
    cpu.reparam_bwd(edml, dz, ml, eps, coef, 4, 1.0)
    close(dml, edml, 1e-5)
    G = 501
    X, dY = rnd(M, G, seed=4), rnd(M, G, seed=5)
    Y, rs, eY, ers = torch.empty(M, G).cuda(), torch.empty(M).cuda(), torch.empty(M, G), torch.empty(M)
    ops.libsize_fwd(Y, dev(X), rs, 1e4)
    cpu.libsize_fwd(eY, X, ers, 1e4)
    close(Y, eY, 1e-5), close(rs, ers, 1e-5)
    dX, edX = torch.empty(M, G).cuda(), torch.empty(M, G)
    ops.libsize_bwd(dX, dev(dY), dev(X), rs, 1e4)
    cpu.libsize_bwd(edX, dY, X, ers, 1e4)
    close(dX, edX, 1e-4)
    real, fake, alpha = rnd(M, G, seed=6), rnd(M, G, seed=7), torch.rand(M, 1, generator=torch.Generator().manual_seed(8))
    o, eo = torch.empty(M, G).cuda(), torch.empty(M, G)
    ops.gp_interp(o, dev(real), dev(fake), dev(alpha))
    cpu.gp_interp(eo, real, fake, alpha)
    close(o, eo, 1e-6)
    # the critic step's stacked input (real | generated | interpolated rows and their labels) in one launch, written
    # into a view of a wider buffer, with and without labels
    for n_cov in (0, 1, 2):
        wide = torch.full((3 * M, G + 12), -7.0).cuda()
        ewide = torch.full((3 * M, G + 12), -7.0)
        lab = elab = rl = fl = None
        if n_cov:
            gen = torch.Generator().manual_seed(10)
            rl, fl = torch.randint(0, 12, (M, n_cov), generator=gen), torch.randint(0, 12, (M, n_cov), generator=gen)
            lab, elab = torch.full((3 * M, n_cov), -1).cuda(), torch.full((3 * M, n_cov), -1)
        ops.critic_inputs(wide[:, :G], lab, dev(real), dev(fake), dev(alpha), dev(rl) if n_cov else None,
                          dev(fl) if n_cov else None)
        cpu.critic_inputs(ewide[:, :G], elab, real, fake, alpha, rl, fl)
        assert torch.equal(wide.cpu()[:2 * M], ewide[:2 * M]) and torch.equal(wide.cpu()[:, G:], ewide[:, G:])
        close(wide[2 * M:, :G], ewide[2 * M:, :G], 1e-6)
        if n_cov:
            assert torch.equal(lab.cpu(), elab)
    V, rp, eV, erp = torch.empty(M, G).cuda(), torch.empty(M).cuda(), torch.empty(M, G), torch.empty(M)
    ux = rnd(M, G, seed=9) * 0.05
    ops.gp_penalty(V, rp, dev(ux), 13.0)
    cpu.gp_penalty(eV, erp, ux, 13.0)
    close(V, eV, 2e-5), close(rp, erp, 2e-5)


def test_reductions_and_loss_bookkeeping(ops):
    M, N = 200, 777
    A, B = rnd(M, N, seed=1), rnd(M, N, seed=2)
    Cn = rnd(12, N, seed=3)
    labels = torch.randint(0, 12, (M, 1), generator=torch.Generator().manual_seed(4))
    ml = rnd(M, 512, seed=5) * 0.3
    red = (torch.zeros(8, 1024).cuda(), torch.zeros(8, dtype=torch.int32).cuda())
    ered = (torch.zeros(8, 1024, dtype=torch.float64), torch.zeros(8, dtype=torch.int32))
    c64 = CpuOps(torch.float64)
    ops.reduce_sum(red, 0, dev(A)), c64.reduce_sum(ered, 0, A.double())
    ops.reduce_l1(red, 1, dev(A), dev(B)), c64.reduce_l1(ered, 1, A.double(), B.double())
    ops.reduce_centroid_sq(red, 2, dev(A), dev(Cn), dev(labels)), c64.reduce_centroid_sq(ered, 2, A.double(), Cn.double(), labels)
    ops.reduce_kl(red, 3, dev(ml)), c64.reduce_kl(ered, 3, ml.double())
    ops.reduce_sqdiff(red, 4, dev(A[0]), dev(B[0])), c64.reduce_sqdiff(ered, 4, A[0].double(), B[0].double())
    ops.reduce_sum(red, 5, dev(A[:, :1])), c64.reduce_sum(ered, 5, A[:, :1].double())
    slots = torch.arange(6, dtype=torch.int32)
    scales = torch.tensor([1.0, 0.5, 2.0, -0.5, 1.0, 1.0], dtype=torch.float64)
    tot, etot = torch.zeros(6, dtype=torch.float64).cuda(), torch.zeros(6, dtype=torch.float64)
    ops.sum_parts(tot, red, dev(slots), dev(scales))
    c64.sum_parts(etot, ered, slots, scales)
    assert int(red[1][0]) <= 1024
    close(tot, etot, 1e-5, "sum_parts")     # float32 partial sums inside each block
    # loss bookkeeping: two consecutive calls (moving averages carry over)
    desc = dict(terms=[[(0, -1.0 / M), (5, -1.0 / M)], [(1, 1.0 / (M * N))], [(2, 1.0 / (M * N))], [(4, 0.5 / N)],
                       [(3, -0.5 / (M * 512))], [(1, 0.5 / (M * N)), (0, 0.25 / (M * N))]], ema_alpha=0.3)
    desc["lambda"] = [1.0, 1.0, 0.2, 0.3, 0.1, 0.7]      # the sixth term: fixed weight, outside the re-scaling
    ema, eema = torch.zeros(5, dtype=torch.float64).cuda(), torch.zeros(5, dtype=torch.float64)
    losses, elosses = torch.zeros(12, dtype=torch.float64).cuda(), torch.zeros(12, dtype=torch.float64)
    coef, ecoef = torch.zeros(6).cuda(), torch.zeros(6)
    for _ in range(2):
        ops.gloss_finalize(ema, losses, coef, red, desc)
        c64.gloss_finalize(eema, elosses, ecoef, ered, desc)
    close(losses, elosses, 1e-5, "losses"), close(coef, ecoef, 1e-5, "coef"), close(ema, eema, 1e-5, "ema")
    out, eout = torch.zeros(N).cuda(), torch.zeros(N, dtype=torch.float64)
    ops.colsum(out, dev(A), 1.0 / M), c64.colsum(eout, A.double(), 1.0 / M)
    close(out, eout, 1e-5, "colmean")
    v, ev = torch.zeros(1).cuda(), torch.zeros(1, dtype=torch.float64)
    ops.colvar_mean(v, torch.zeros(64, dtype=torch.float64).cuda(), dev(A)), c64.colvar_mean(ev, None, A.double())
    close(v, ev, 1e-5, "colvar")
    # gradients of the loss terms
    cf = torch.tensor([0.0, 2.0, 3.0, 4.0])
    for fn, args, eargs in [
        ("l1_bwd", (dev(A), dev(B), dev(cf), 1, 0.25), (A, B, cf, 1, 0.25)),
        ("centroid_bwd", (dev(A), dev(Cn), dev(labels), dev(cf), 2, 0.5), (A, Cn, labels, cf, 2, 0.5)),
    ]:
        g, e = torch.ones(M, N).cuda(), torch.ones(M, N)
        getattr(ops, fn)(g, *args, accumulate=True)
        getattr(cpu, fn)(e, *eargs, accumulate=True)
        close(g, e, 1e-6, fn)
    g, e = torch.zeros(M, 256).cuda(), torch.zeros(M, 256)
    mf, mr = rnd(256, seed=8), rnd(256, seed=9)
    ops.fm_bwd(g, dev(mf), dev(mr), dev(cf), 3, 0.1), cpu.fm_bwd(e, mf, mr, cf, 3, 0.1)
    close(g, e, 1e-6, "fm_bwd")


def test_adam_matches_torch_optimizer(ops):
    """Five steps of the fused kernel against torch.optim.Adam(amsgrad=True) on CPU."""
    n = 10007
    p0 = rnd(n, seed=1)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=3e-4, betas=(0.5, 0.9), amsgrad=True)
    p = dev(p0.clone())
    m, v, vm = torch.zeros(n).cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    lr_step = torch.zeros(2, dtype=torch.float64).cuda()
    for t in range(1, 6):
        g = rnd(n, seed=10 + t) * (10.0 ** (t - 3))
        ref.grad = g.clone()
        opt.step()
        lr_step.copy_(torch.tensor([3e-4, float(t)], dtype=torch.float64))
        ops.adam(p, dev(g), m, v, vm, lr_step, 0.5, 0.9, 1e-8)
    close(p, ref.detach(), 1e-6, "adam p")


def test_poisson_cycle_loss_weights_and_plain_adam(ops):
    """Kernels of the curriculum training (GAN_optimize_optuna.ipynb cell-1): poisson_loss (GAN_functions.py:216-217)
    and its gradient against autograd, loss weights taken as given, Adam without the running maximum."""
    M, N = 130, 333
    target = torch.poisson(torch.full((M, N), 0.4), generator=torch.Generator().manual_seed(1))
    wide = rnd(M, N + 6, seed=2).abs() * 2
    wide[:, 5] = 0.0                                  # pred = 0 where the library-size ReLU cut
    pred = wide[:, 3:3 + N]
    red = (torch.zeros(8, 1024).cuda(), torch.zeros(8, dtype=torch.int32).cuda())
    ops.reduce_poisson(red, 1, dev(wide)[:, 3:3 + N], dev(target))
    pr = pred.double().clone().requires_grad_(True)
    loss = torch.mean(pr - target.double() * torch.log(pr + 1e-10))
    loss.backward()
    tot = torch.zeros(1, dtype=torch.float64).cuda()
    ops.sum_parts(tot, red, dev(torch.tensor([1], dtype=torch.int32)), dev(torch.tensor([1.0 / (M * N)], dtype=torch.float64)))
    assert abs(float(tot[0]) - float(loss.detach())) <= 2e-5 * abs(float(loss.detach())) + 1e-6
    cf = torch.tensor([0.0, 3.0])
    g = torch.ones(M, N).cuda()
    ops.poisson_bwd(g, dev(wide)[:, 3:3 + N], dev(target), dev(cf), 1, 1.0 / (M * N), accumulate=True)
    exp = 1.0 + 3.0 * pr.grad
    big = exp.abs() > 1e3                             # target > 0 at pred = 0: 1e10-sized entries, compared relatively
    assert float(((g.cpu().double() - exp)[~big]).abs().max()) <= 2e-5 * float(exp[~big].abs().max())
    assert float(((g.cpu().double() - exp)[big] / exp[big]).abs().max()) <= 1e-5 if bool(big.any()) else True
    # loss weights handed in: no moving averages, coefficients = the weights
    desc = dict(terms=[[(1, 1.0)], [(1, 2.0)], [(1, 0.0)], [(1, -1.0)], [(1, 0.5)]], ema_alpha=0.3,
                **{"lambda": [9.0] * 5})
    w = torch.tensor([0.0, 7.5, 0.25, 0.0, 0.01])
    desc["weights"] = dev(w)
    ema = torch.full((5,), 3.0, dtype=torch.float64).cuda()
    losses, coef = torch.zeros(12, dtype=torch.float64).cuda(), torch.zeros(6).cuda()
    ops.gloss_finalize(ema, losses, coef, red, desc)
    assert torch.equal(coef.cpu()[:5], w) and torch.equal(ema.cpu(), torch.full((5,), 3.0, dtype=torch.float64))
    L = losses.cpu()
    assert abs(float(L[5]) - float((w.double() * L[:5]).sum())) <= 1e-9 * abs(float(L[5])) + 1e-12
    # plain Adam
    n = 5003
    p0 = rnd(n, seed=1)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=2e-4, betas=(0.35, 0.96))
    p, m, v = dev(p0.clone()), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    lr_step = torch.zeros(2, dtype=torch.float64).cuda()
    for t in range(1, 6):
        gr = rnd(n, seed=10 + t) * (10.0 ** (3 - t))   # shrinking gradients: a running maximum would differ
        ref.grad = gr.clone()
        opt.step()
        lr_step.copy_(torch.tensor([2e-4, float(t)], dtype=torch.float64))
        ops.adam(p, dev(gr), m, v, None, lr_step, 0.35, 0.96, 1e-8)
    close(p, ref.detach(), 1e-6, "plain adam p")


@pytest.mark.parametrize("n_cells,train", [(1, True), (7, True), (512, True), (33, False)])
def test_attention_layernorm(ops, n_cells, train):
    heads = 4
    X = rnd(n_cells * 16, 16, seed=1)
    params = rnd(ATTN_NPARAM, seed=2) * 0.3
    params[1088:1104] += 1.0
    keep = ((torch.rand(n_cells * heads, 16, 16, generator=torch.Generator().manual_seed(3)) < 0.9).float() / 0.9) if train else None
    c64 = CpuOps(torch.float64)
    eY = torch.empty(n_cells * 16, 16, dtype=torch.float64)
    c64.attn_fwd(eY, X.double(), None if keep is None else keep.double(), params.double(), heads)
    Y = torch.empty(n_cells * 16, 16).cuda()
    ops.attn_fwd(Y, dev(X), dev(keep), dev(params), heads)
    close(Y, eY, 2e-5, "attn fwd")
    dY = rnd(n_cells * 16, 16, seed=4)
    edX, edp = torch.empty(n_cells * 16, 16, dtype=torch.float64), torch.ones(ATTN_NPARAM, dtype=torch.float64)
    c64.attn_bwd(edX, edp, None, dY.double(), X.double(), None if keep is None else keep.double(), params.double(), heads, True)
    dX, dp = torch.empty(n_cells * 16, 16).cuda(), torch.ones(ATTN_NPARAM).cuda()
    ws = torch.empty(ops.attn_bwd_ws(n_cells)).cuda()
    ops.attn_bwd(dX, dp, ws, dev(dY), dev(X), dev(keep), dev(params), heads, True)
    close(dX, edX, 5e-5, "attn dX")
    close(dp, edp, 5e-5, "attn dparams")


@pytest.mark.parametrize("E,heads,n_cells,train", [
    (8, 2, 5, True), (8, 8, 64, True), (4, 1, 3, True), (4, 4, 700, True), (16, 8, 9, True), (16, 16, 33, False),
    (16, 1, 2, True), (32, 4, 130, True), (32, 8, 512, True), (32, 16, 1, True), (32, 32, 17, False), (16, 2, 600, True)])
def test_attention_layernorm_other_geometries(ops, E, heads, n_cells, train):
    """Token width latent_dim / 16 and the head count come from the config (GAN_functions.py:82,:103-113): the
    general kernel against the float64 restatement, more cells than workgroups included."""
    from cycle_gan_rna_amd.ops import attn_nparam
    npar = attn_nparam(E)
    X = rnd(n_cells * 16, E, seed=1)
    params = rnd(npar, seed=2) * 0.3
    params[4 * E * E + 4 * E:4 * E * E + 5 * E] += 1.0            # LayerNorm weight around 1
    g = torch.Generator().manual_seed(3)
    keep = ((torch.rand(n_cells * heads, 16, 16, generator=g) < 0.9).float() / 0.9) if train else None
    c64 = CpuOps(torch.float64)
    eY = torch.empty(n_cells * 16, E, dtype=torch.float64)
    c64.attn_fwd(eY, X.double(), None if keep is None else keep.double(), params.double(), heads)
    Y = torch.empty(n_cells * 16, E).cuda()
    ops.attn_fwd(Y, dev(X), dev(keep), dev(params), heads)
    close(Y, eY, 2e-5, "attn fwd")
    dY = rnd(n_cells * 16, E, seed=4)
    edX, edp = torch.empty(n_cells * 16, E, dtype=torch.float64), torch.ones(npar, dtype=torch.float64)
    c64.attn_bwd(edX, edp, None, dY.double(), X.double(), None if keep is None else keep.double(), params.double(),
                 heads, True)
    dX, dp = torch.empty(n_cells * 16, E).cuda(), torch.ones(npar).cuda()
    ws = torch.empty(ops.attn_bwd_ws(n_cells, E, heads)).cuda()
    ops.attn_bwd(dX, dp, ws, dev(dY), dev(X), dev(keep), dev(params), heads, True)
    close(dX, edX, 5e-5, "attn dX")
    close(dp, edp, 5e-5, "attn dparams")


def test_attention_rejects_geometry_without_kernel(ops):
    from cycle_gan_rna_amd._lib import CgError
    with pytest.raises(CgError, match="is not built"):
        ops.attn_bwd_ws(4, 12, 4)
    with pytest.raises(CgError, match="is not built"):
        ops.attn_bwd_ws(4, 16, 3)


@pytest.mark.parametrize("D", [4, 8, 32])
def test_token_residual_block_shapes_through_general_kernels(ops, D):
    """(cells * 16, D) token matrices with D x D weights: the shapes the ResidualBlock (GAN_functions.py:55-71) hands
    to cg_gemm / cg_bn_fwd / cg_bn_bwd when the token width is not 16."""
    T = 16 * 96
    X, W, b = rnd(T, D, seed=1), rnd(D, D, seed=2) * 0.4, rnd(D, seed=3)
    gamma, beta = 1 + 0.1 * rnd(D, seed=4), 0.1 * rnd(D, seed=5)
    c64 = CpuOps(torch.float64)
    for slope in (0.0, 1.0):
        eT, eY = torch.empty(T, D, dtype=torch.float64), torch.empty(T, D, dtype=torch.float64)
        em, ei, erm, erv = (torch.zeros(D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64),
                            torch.zeros(D, dtype=torch.float64), torch.ones(D, dtype=torch.float64))
        c64.gemm(eT, X.double(), W.double(), False, True, bias=b.double())
        c64.bn_fwd(eY, eT, gamma.double(), beta.double(), em, ei, erm, erv, slope)
        Tm, Y = torch.empty(T, D).cuda(), torch.empty(T, D).cuda()
        m, i, rm, rv = torch.zeros(D).cuda(), torch.zeros(D).cuda(), torch.zeros(D).cuda(), torch.ones(D).cuda()
        ops.gemm(Tm, dev(X), dev(W), False, True, bias=dev(b))
        ops.bn_fwd(Y, Tm, dev(gamma), dev(beta), m, i, rm, rv, slope)
        close(Tm, eT, 2e-5, "token linear")
        close(Y, eY, 5e-5, "token batchnorm")
        close(rv, erv, 1e-5, "running_var")
        dY = rnd(T, D, seed=6)
        edX, edg, edb = torch.empty(T, D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64), torch.zeros(D, dtype=torch.float64)
        c64.bn_bwd(edX, dY.double(), eT, eY, None, em, ei, gamma.double(), edg, edb, True, slope)
        dX, dg, db = torch.empty(T, D).cuda(), torch.zeros(D).cuda(), torch.zeros(D).cuda()
        ops.bn_bwd(dX, dev(dY), Tm, Y, None, m, i, dev(gamma), dg, db, True, slope)
        close(dX, edX, 1e-4, "token batchnorm dX")
        close(dg, edg, 1e-4, "token batchnorm dgamma")
        edW, dW = torch.zeros(D, D, dtype=torch.float64), torch.zeros(D, D).cuda()
        c64.gemm(edW, edX, X.double(), True, False, accumulate=True)
        ops.gemm(dW, dX, dev(X), True, False, accumulate=True)
        close(dW, edW, 1e-4, "token weight gradient")


@pytest.mark.parametrize("M,N,K,ta,tb,epi,acc", [
    (512, 12, 1024, 0, 0, EPI_NONE, 0), (1536, 1, 256, 0, 1, EPI_NONE, 0), (1, 256, 1024, 1, 0, EPI_NONE, 1),
    (512, 256, 1, 0, 0, EPI_MUL, 0), (300, 7, 100, 0, 0, EPI_LEAKY, 0), (130, 1024, 12, 0, 1, EPI_NONE, 1),
    (9, 200, 333, 1, 0, EPI_LEAKY, 0), (16, 130, 64, 0, 0, EPI_NONE, 1), (257, 16, 70, 0, 1, EPI_MUL, 0),
    (5, 3, 2, 1, 1, EPI_NONE, 0), (1024, 12, 1024, 0, 0, EPI_NONE, 0), (100, 3, 1300, 0, 1, EPI_NONE, 1),
    (7, 16, 600, 0, 0, EPI_MUL, 0), (515, 5, 64, 0, 0, EPI_NONE, 0), (12, 12, 1024, 0, 0, EPI_NONE, 1),
    (32, 15, 700, 0, 1, EPI_LEAKY, 0)])
def test_gemm_small_dimension_kernels(ops, M, N, K, ta, tb, epi, acc):
    """Products with M, N or K <= 16 take plain-FMA kernels (the critic's head, the embedding slice of its first
    layer, rank-one products): same results and epilogues as the tiled kernel, operands that are column slices."""
    wide_a = rnd(*((K, M + 5) if ta else (M, K + 5)), seed=11)
    wide_b = rnd(*((N, K + 3) if tb else (K, N + 3)), seed=12)
    A = wide_a[:, 2:2 + (M if ta else K)]
    B = wide_b[:, 1:1 + (K if tb else N)]
    bias = rnd(N, seed=13) if not acc else None
    E = (torch.rand(M, N, generator=torch.Generator().manual_seed(4)) < 0.7).float() / 0.7 if epi != EPI_NONE else None
    c0 = rnd(M, N, seed=14)
    exp = c0.double().clone()
    aux = torch.empty(M, N, dtype=torch.float64)
    cpu.gemm(exp, A.double(), B.double(), ta, tb, bias=None if bias is None else bias.double(), epi=epi,
             E=None if E is None else E.double(), aux=aux if epi == EPI_LEAKY else None, slope=0.2, accumulate=bool(acc))
    got, gaux = dev(c0.clone()), torch.zeros(M, N).cuda()
    ops.gemm(got, dev(wide_a)[:, 2:2 + (M if ta else K)], dev(wide_b)[:, 1:1 + (K if tb else N)], ta, tb, bias=dev(bias),
             epi=epi, E=dev(E), aux=gaux if epi == EPI_LEAKY else None, slope=0.2, accumulate=bool(acc))
    close(got, exp, 2e-5, "small-dimension gemm")
    if epi == EPI_LEAKY:
        h = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double()) + bias.double()
        ok = h.abs() > 1e-4
        assert torch.allclose(gaux.cpu().double()[ok], aux[ok], rtol=1e-6, atol=0)


@pytest.mark.parametrize("M,N,K,with_mask", [(512, 512, 1024, False), (512, 1024, 2048, True), (300, 256, 5000, False),
                                             (512, 2048, 268, False), (37, 260, 4096, True)])
def test_linear_batchnorm_with_slabs_left_to_the_consumer(ops, M, N, K, with_mask):
    """cg_gemm_slabs + cg_bn_fwd_slabs (the BatchNorm adds the split-K partial sums while it loads) against cg_gemm +
    cg_bn_fwd: the same additions in the same order, so pre-activation, activation, statistics and running statistics
    are equal bit for bit; and against the CPU restatement of the slab sum."""
    x, w, bias = dev(rnd(M, K, seed=1)), dev(rnd(N, K, seed=2) * 0.05), dev(rnd(N, seed=3))
    gamma, beta = dev(rnd(N, seed=4).abs() + 0.5), dev(rnd(N, seed=5))
    mask = dev((torch.rand(M, N, generator=torch.Generator().manual_seed(6)) > 0.3).float() / 0.7) if with_mask else None
    out = {}
    for mode in ("plain", "slabs"):
        h0, y = torch.full((M, N), -3.0).cuda(), torch.empty(M, N).cuda()
        pm = torch.empty(M, N).cuda() if with_mask else None
        mean, invstd = torch.empty(N).cuda(), torch.empty(N).cuda()
        rm, rv = torch.zeros(N).cuda(), torch.ones(N).cuda()
        if mode == "plain":
            ops.gemm(h0, x, w, False, True, bias=bias)
            ops.bn_fwd(y, h0, gamma, beta, mean, invstd, rm, rv, 0.2, mask, pm)
            n = None
        else:
            assert ops.bn_slabs_ok(M, N)
            n = ops.gemm(h0, x, w, False, True, bias=bias, keep_slabs=True)
            if n > 1:
                slabs = ops._ws()[0][:n * M * N].view(n, M, N).clone()
                assert float(h0.min()) == -3.0 and float(h0.max()) == -3.0          # untouched until the consumer runs
                ops.bn_fwd_slabs(y, h0, n, bias, gamma, beta, mean, invstd, rm, rv, 0.2, mask, pm)
                eh0, ey = torch.empty(M, N), torch.empty(M, N)
                cpu.bn_fwd_slabs(ey, eh0, slabs.cpu(), bias.cpu(), gamma.cpu(), beta.cpu(), torch.empty(N), torch.empty(N),
                                 None, None, 0.2, None if mask is None else mask.cpu())
                assert torch.equal(h0.cpu(), eh0)
                close(y, ey, 2e-5)
            else:
                ops.bn_fwd(y, h0, gamma, beta, mean, invstd, rm, rv, 0.2, mask, pm)
        out[mode] = (h0, y, mean, invstd, rm, rv, pm, n)
    for a, b in zip(out["plain"][:7], out["slabs"][:7]):
        assert (a is None and b is None) or torch.equal(a, b)
    if K >= 1024 and M * N <= 512 * 1024:
        assert out["slabs"][7] > 1, "expected this shape to split K (otherwise the test does not reach the slab path)"
    assert not ops.bn_slabs_ok(513, 512) and not ops.bn_slabs_ok(512, 128)


@pytest.mark.parametrize("M,N,K", [(512, 1024, 512), (512, 512, 5000), (300, 256, 2048), (512, 2048, 1024), (37, 260, 4096)])
def test_batchnorm_backward_with_slabs_left_to_the_consumer(ops, M, N, K):
    """An input-gradient product whose split-K partial sums stay in the scratch + cg_bn_bwd_slabs (the BatchNorm
    backward of the layer below adds them while it loads) against cg_gemm + cg_bn_bwd: the same additions in the same
    order, so the input gradient and the parameter gradients are equal bit for bit."""
    d, w = dev(rnd(M, K, seed=1)), dev(rnd(K, N, seed=2) * 0.05)
    h0, gamma = dev(rnd(M, N, seed=3)), dev(rnd(N, seed=4).abs() + 0.5)
    act = torch.relu(h0)
    mean, var = h0.mean(0), h0.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    out = {}
    for mode in ("plain", "slabs"):
        dact, dh0 = torch.full((M, N), -3.0).cuda(), torch.empty(M, N).cuda()
        dgamma, dbeta = dev(rnd(N, seed=5)), dev(rnd(N, seed=6))
        if mode == "plain":
            ops.gemm(dact, d, w, False, False)
            ops.bn_bwd(dh0, dact, h0, act, None, mean, invstd, gamma, dgamma, dbeta, True, 0.0)
            n = None
        else:
            assert ops.bn_bwd_slabs_ok(M, N)
            n = ops.gemm(dact, d, w, False, False, keep_slabs=True)
            if n > 1:
                assert float(dact.min()) == -3.0 and float(dact.max()) == -3.0      # never written: only the consumer reads it
                ops.bn_bwd_slabs(dh0, n, h0, act, None, mean, invstd, gamma, dgamma, dbeta, True, 0.0)
            else:
                ops.bn_bwd(dh0, dact, h0, act, None, mean, invstd, gamma, dgamma, dbeta, True, 0.0)
        out[mode] = (dh0, dgamma, dbeta, n)
    for a, b in zip(out["plain"][:3], out["slabs"][:3]):
        assert torch.equal(a, b)
    if K >= 1024 and M * N <= 512 * 1024:
        assert out["slabs"][3] > 1, "expected this shape to split K (otherwise the test does not reach the slab path)"
    assert not ops.bn_bwd_slabs_ok(513, 512) and not ops.bn_bwd_slabs_ok(512, 128)


@pytest.mark.parametrize("M,N,G,E", [(512, 1024, 5000, 12), (37, 1024, 200, 12), (64, 256, 96, 0)])
def test_first_layer_on_the_interpolates_by_linearity(ops, M, N, G, E):
    """cg_interp_hidden against the product it replaces: the first Linear on [alpha x_r + (1 - alpha) x_f | emb_f]
    from its outputs on [x_r | emb_r] and [x_f | emb_f].  Equal up to float32 rounding of two summation orders."""
    W = G + E
    xe = torch.zeros(3 * M, W)
    xe[:2 * M, :G] = rnd(2 * M, G, seed=1)
    if E:
        xe[:2 * M, G:] = rnd(2 * M, E, seed=2)
        xe[2 * M:, G:] = xe[M:2 * M, G:]
    alpha = torch.rand(M, 1, generator=torch.Generator().manual_seed(3))
    xe[2 * M:, :G] = alpha * xe[:M, :G] + (1 - alpha) * xe[M:2 * M, :G]
    w1, bias = rnd(N, W, seed=4) * 0.05, rnd(N, seed=5)
    direct = torch.empty(3 * M, N).cuda()
    ops.gemm(direct, dev(xe), dev(w1), False, True, bias=dev(bias))
    h = torch.full((3 * M, N), 9.0).cuda()
    h[:2 * M] = direct[:2 * M]
    dw1 = dev(w1)
    ops.interp_hidden(h, M, dev(alpha), dev(xe)[:, G:] if E else None, dw1[:, G:] if E else None)
    assert torch.equal(h[:2 * M], direct[:2 * M])
    close(h[2 * M:], direct[2 * M:].cpu(), 2e-6)
    eh = direct.cpu().clone()
    eh[2 * M:] = 9.0
    cpu.interp_hidden(eh, M, alpha, xe[:, G:] if E else None, w1[:, G:] if E else None)
    close(h[2 * M:], eh[2 * M:], 2e-6)


def _bf16_ops():
    from cycle_gan_rna_amd.ops import HipOps
    return HipOps("cuda:0", precision="bf16")


def _image(ops16, t):
    """The bf16 image the backend keeps of the shadowed float32 tensor `t` (whole tensor)."""
    p = t.data_ptr()
    for lo, hi, sh in ops16._shadows:
        if lo <= p < hi:
            off = (p - lo) // 4
            return sh[off:off + t.numel()].view(t.shape)
    raise AssertionError("tensor has no bf16 image")


@pytest.mark.parametrize("M,N,K,ta,tb,epi,acc", [
    (512, 1024, 2048, 0, 1, EPI_LEAKY, 0),      # tiled kernel, epilogue in the kernel
    (512, 512, 1024, 0, 1, EPI_NONE, 0),        # split K: the slab reduce writes the image
    (256, 512, 512, 1, 0, EPI_NONE, 1),         # accumulate
    (512, 512, 256, 0, 0, EPI_MUL, 0), (130, 70, 300, 0, 1, EPI_NONE, 0), (64, 12, 512, 0, 0, EPI_NONE, 0),
    (33, 2, 100, 0, 1, EPI_NONE, 0)])
@pytest.mark.parametrize("dtype", [0, 1])
def test_gemm_keeps_bf16_image_of_its_result(M, N, K, ta, tb, epi, acc, dtype):
    """cg_gemm_ex (through HipOps.gemm on a shadowed output): the bf16 image equals the stored float32 result rounded
    to nearest even, for every kernel that can take the product (tiled float32 / bf16, split-K + reduce, small-dimension
    kernels) and every epilogue; rows of a row slice outside the product keep their old image."""
    ops16 = _bf16_ops()
    full = ops16.shadowed(M + 3, N)
    assert ops16._shadows, "bf16 mode registers an image"
    out = full[:M]
    if acc:
        out.copy_(dev(rnd(M, N, seed=5)))
        ops16.refresh_shadow(full.view(-1), _image(ops16, full).view(-1))
    A = dev(rnd(*((K, M) if ta else (M, K)), seed=1))
    B = dev(rnd(*((N, K) if tb else (K, N)), seed=2))
    E = dev((torch.rand(M, N, generator=torch.Generator().manual_seed(4)) < 0.7).float() / 0.7) if epi != EPI_NONE else None
    aux = torch.empty(M, N).cuda() if epi == EPI_LEAKY else None
    ops16.gemm(out, A, B, ta, tb, bias=None if acc else dev(rnd(N, seed=3)), epi=epi, E=E, aux=aux, slope=0.2,
               accumulate=bool(acc), dtype=dtype)
    torch.cuda.synchronize()
    img = _image(ops16, full)
    assert torch.equal(img[:M], out.bfloat16())
    assert float(img[M:].float().abs().max()) == 0.0          # untouched rows
    assert float(out.abs().max()) > 0.0


@pytest.mark.parametrize("M,N,nseg", [(512, 1024, 1), (100, 300, 1), (600, 256, 1), (96, 1024, 3)])
def test_batchnorm_kernels_keep_bf16_images(M, N, nseg):
    """bn_fwd / bn_fwd_seg / bn_fwd_slabs / bn_eval write the bf16 image of the activation and bn_bwd / bn_bwd_seg
    that of the input gradient (one-launch and two-launch forms): image == float32 output rounded to nearest even."""
    ops16 = _bf16_ops()
    R = M * nseg
    X, dY = dev(rnd(R, N, seed=1)), dev(rnd(R, N, seed=2))
    gamma, beta = dev(rnd(N, seed=3)), dev(rnd(N, seed=4))
    mean, invstd = torch.empty(nseg, N).cuda(), torch.empty(nseg, N).cuda()
    rm, rv = torch.zeros(N).cuda(), torch.ones(N).cuda()
    Y, dX = ops16.shadowed(R, N), ops16.shadowed(R, N)
    if nseg == 1:
        ops16.bn_fwd(Y, X, gamma, beta, mean[0], invstd[0], rm, rv, 0.0)
    else:
        ops16.bn_fwd_seg(Y, X, nseg, None, gamma, beta, mean, invstd, rm, rv, 0.2)
    assert torch.equal(_image(ops16, Y), Y.bfloat16()) and float(Y.abs().max()) > 0
    dg, db = torch.zeros(N).cuda(), torch.zeros(N).cuda()
    if nseg == 1:
        ops16.bn_bwd(dX, dY, X, Y, None, mean[0], invstd[0], gamma, dg, db, True, 0.0)
    else:
        ops16.bn_bwd_seg(dX, dY, X, Y, None, nseg, mean, invstd, gamma, dg, db, True, 0.2)
    assert torch.equal(_image(ops16, dX), dX.bfloat16()) and float(dX.abs().max()) > 0
    if nseg == 1:
        Y.zero_()
        ops16.bn_eval(Y, X, gamma, beta, rm, rv, 0.0)
        assert torch.equal(_image(ops16, Y), Y.bfloat16()) and float(Y.abs().max()) > 0
    if nseg == 1 and ops16.bn_slabs_ok(M, N):
        x, w, bias = dev(rnd(M, 2048, seed=5)), dev(rnd(N, 2048, seed=6) * 0.05), dev(rnd(N, seed=7))
        H = torch.empty(M, N).cuda()
        n = ops16.gemm(H, x, w, False, True, bias=bias, keep_slabs=True, dtype=0)
        if n > 1:
            Y.zero_()
            ops16.bn_fwd_slabs(Y, H, n, bias, gamma, beta, mean[0], invstd[0], rm, rv, 0.0)
            assert torch.equal(_image(ops16, Y), Y.bfloat16()) and float(Y.abs().max()) > 0


@pytest.mark.parametrize("M,N,T", [(2048, 1024, 12), (512, 1024, 12), (37, 50, 3), (1000, 1024, 32)])
def test_label_colsum(ops, M, N, T):
    """cg_label_colsum: sums of rows by label (labels outside the table are skipped), with scale and accumulation,
    against the CPU restatement; twice the same bits (no atomics)."""
    x = rnd(M, N + 8, seed=1)[:, :N]
    lab = torch.randint(-1, T, (M, 1), generator=torch.Generator().manual_seed(2))
    base = rnd(T, N, seed=3)
    exp = base.clone()
    cpu.label_colsum(exp, x.double().float(), lab, scale=-0.5, accumulate=True)
    xd, ld = dev(x), lab.cuda()
    outs = []
    for _ in range(2):
        out = dev(base.clone())
        ops.label_colsum(out, xd, ld, scale=-0.5, accumulate=True)
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    close(outs[0], exp, 1e-5)
    plain = torch.full((T, N), 7.0).cuda()
    ops.label_colsum(plain, xd, ld)
    exp2 = torch.zeros(T, N)
    cpu.label_colsum(exp2, x, lab)
    close(plain, exp2, 1e-5)
    lab2 = torch.randint(-1, T, (M, 1), generator=torch.Generator().manual_seed(5))     # a second label: subtracted there
    diff, exp3 = torch.full((T, N), 7.0).cuda(), torch.zeros(T, N)
    ops.label_colsum(diff, xd, ld, minus=lab2.cuda())
    cpu.label_colsum(exp3, x, lab, minus=lab2)
    close(diff, exp3, 1e-5)


@pytest.mark.parametrize("M,N,K,ta,tb,epi,acc,dtype", [
    (512, 1024, 2048, 0, 1, EPI_LEAKY, 0, 0), (512, 2048, 5000, 0, 1, EPI_NONE, 0, 0), (512, 512, 1024, 0, 1, EPI_NONE, 0, 0),
    (256, 512, 1024, 1, 0, EPI_NONE, 1, 0), (512, 268, 2048, 0, 0, EPI_MUL, 0, 0), (1024, 1024, 5012, 0, 1, EPI_NONE, 0, 0),
    (130, 70, 3000, 0, 1, EPI_NONE, 0, 0), (131, 67, 2999, 1, 1, EPI_LEAKY, 0, 0), (512, 1024, 2048, 0, 1, EPI_LEAKY, 0, 1),
    (512, 512, 5000, 0, 0, EPI_NONE, 1, 1), (100, 36, 4000, 0, 1, EPI_NONE, 0, 1)])
def test_gemm_split_k_combined_in_the_kernel(M, N, K, ta, tb, epi, acc, dtype):
    """CG_WS_TILE_COUNTERS: a product that splits K and combines its slabs in the same launch (the last workgroup of
    each output tile adds them in split order) returns the bits of the two-launch form, whatever the arrival order
    (repeated calls), with every epilogue, with accumulation, on both MFMA paths, for sizes that are no multiple of
    a tile or of four; the counters are zero again afterwards."""
    import ctypes as C
    from cycle_gan_rna_amd import _lib
    lib = _lib.load_ctypes()
    A = dev(rnd(*((K, M) if ta else (M, K)), seed=1))
    B = dev(rnd(*((N, K) if tb else (K, N)), seed=2))
    bias = None if acc else dev(rnd(N, seed=3))
    E = dev((torch.rand(M, N, generator=torch.Generator().manual_seed(4)) < 0.7).float() / 0.7) if epi != EPI_NONE else None
    base = dev(rnd(M, N, seed=5))
    ws_floats = 8 * M * N + 4096
    ws = torch.full((ws_floats,), float("nan")).cuda()
    ws[-4096:].zero_()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731

    def run(flag):
        out = base.clone()
        aux = torch.zeros(M, N).cuda() if epi == EPI_LEAKY else None
        rc = lib.cg_gemm(p(out), N, p(A), A.shape[1], p(B), B.shape[1], M, N, K, ta, tb, dtype | flag, p(bias), epi, p(E), N,
                         p(aux), N, 0.2, acc, p(ws), (ws_floats if flag else ws_floats - 4096) * 4, C.c_void_p(st))
        assert rc == 0, rc
        return out, aux
    ref, ref_aux = run(0)
    for _ in range(10):
        got, got_aux = run(8)
        assert torch.equal(got, ref)
        assert ref_aux is None or torch.equal(got_aux, ref_aux)
    torch.cuda.synchronize()
    assert int(ws[-4096:].view(torch.int32).abs().max()) == 0
    exp = torch.empty(M, N)
    cpu.gemm(exp, A.cpu().double() if dtype == 0 else A.cpu().bfloat16().double(),
             B.cpu().double() if dtype == 0 else B.cpu().bfloat16().double(), ta, tb,
             bias=None if bias is None else bias.cpu().double())
    if epi == EPI_NONE:
        close(ref, exp + (base.cpu() if acc else 0), 2e-5, "split-K product")


def test_gemm_split_k_combined_in_the_kernel_under_load(ops):
    """The in-kernel combine hands slabs from workgroup to workgroup inside one launch (write-through stores, an
    arrival counter, loads that bypass the CU's cache): checked word for word against the two-launch results while a
    second stream keeps part of the chip busy with other products, with one workspace reused by products of
    different shapes back to back (the reader's caches have seen those addresses before), 60 rounds."""
    shapes = [(512, 1024, 2048, 0, 1), (512, 268, 2048, 0, 0), (1024, 1024, 5012, 0, 1), (512, 512, 1024, 1, 0),
              (512, 2048, 5000, 0, 1), (256, 512, 1024, 1, 0)]
    import os
    cases = []
    for i, (M, N, K, ta, tb) in enumerate(shapes):
        A = dev(rnd(*((K, M) if ta else (M, K)), seed=10 + i))
        B = dev(rnd(*((N, K) if tb else (K, N)), seed=20 + i))
        cases.append((A, B, ta, tb, torch.empty(M, N).cuda(), torch.empty(M, N).cuda()))
    side = ops.stream("side0")
    X, Y, Z = dev(rnd(2048, 2048, seed=1)), dev(rnd(2048, 2048, seed=2)), torch.empty(2048, 2048).cuda()
    # reference: the same products without the flag (two launches), through the C ABI
    import ctypes as C
    from cycle_gan_rna_amd import _lib
    lib = _lib.load_ctypes()
    ws = ops._ws()[0]
    st = torch.cuda.current_stream().cuda_stream
    for A, B, ta, tb, ref, _ in cases:
        M, N = ref.shape
        K = A.shape[0] if ta else A.shape[1]
        rc = lib.cg_gemm(C.c_void_p(ref.data_ptr()), N, C.c_void_p(A.data_ptr()), A.shape[1], C.c_void_p(B.data_ptr()),
                         B.shape[1], M, N, K, ta, tb, 0, None, 0, None, 0, None, 0, 0.0, 0, C.c_void_p(ws.data_ptr()),
                         (ws.numel() - ops.WS_COUNTER_FLOATS) * 4, C.c_void_p(st))
        assert rc == 0
    torch.cuda.synchronize()
    bad = 0
    for rnd_i in range(60):
        with torch.cuda.stream(side):
            for _ in range(1 + rnd_i % 3):
                ops.gemm(Z, X, Y)
        for A, B, ta, tb, ref, out in cases:
            out.fill_(float("nan"))
            ops.gemm(out, A, B, ta, tb)
            bad += int((out != ref).sum())
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} words differ from the two-launch product"
    assert int(ws[-ops.WS_COUNTER_FLOATS:].view(torch.int32).abs().max()) == 0
    assert not os.environ.get("CGRNA_GEMM_SEPARATE_REDUCE"), "the test compared the two-launch form with itself"


@pytest.mark.parametrize("n_cells,with_next,train", [(512, True, True), (512, False, True), (37, True, True), (3, True, False),
                                                     (4100, True, True)])
def test_attention_block_in_one_launch(ops, n_cells, with_next, train):
    """cg_attn_block_fwd (+ cg_tok_bn_lin_fwd_from) against the launches it replaces - cg_tok_bn_res_fwd, cg_attn_ln_fwd
    and the next block's cg_tok_lin_fwd: the residual sum, the block's output, the BatchNorm statistics it saves and
    updates and the next block's pre-activation are the same bits; the column sums of that pre-activation are added in
    another order (relative 1e-12 in float64), and the BatchNorm + Linear that consumes them agrees to float32
    rounding."""
    heads, n = 4, n_cells * 16
    Z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    Zd = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")  # noqa: E731
    Xin, T1 = dev(rnd(n, 16, seed=1)), dev(rnd(n, 16, seed=2))
    W2, b2 = dev(rnd(16, 16, seed=3) * 0.4), dev(rnd(16, seed=5))
    W1n, b1n = dev(rnd(16, 16, seed=13) * 0.4), dev(rnd(16, seed=14))
    g1, be1, g2, be2 = dev(rnd(16, seed=6) + 1.5), dev(rnd(16, seed=7)), dev(rnd(16, seed=8) + 1.5), dev(rnd(16, seed=9))
    rm, rv = dev(rnd(16, seed=10)), dev(rnd(16, seed=11).abs() + 0.5)
    params = rnd(ATTN_NPARAM, seed=2) * 0.3
    params[1088:1104] += 1.0
    params = dev(params)
    keep = dev((torch.rand(n_cells * heads, 16, 16, generator=torch.Generator().manual_seed(3)) < 0.9).float() / 0.9) if train else None
    nb_tok, nb_att = ops.tok_blocks(n), ops.attn_block_blocks(n_cells)
    # the block's first two launches are common to both forms
    s0, s1 = Zd(nb_tok * 32), Zd(nb_tok * 32)
    R, T2 = Z(n, 16), Z(n, 16)
    ops.tok_lin_fwd(T1, s0, Xin, W2, b2)        # any pre-activation with its column sums
    ops.tok_bn_lin_fwd(R, T2, s1, T1, s0, g1, be1, Z(16), Z(16), rm.clone(), rv.clone(), W2, b2)
    # separate launches
    Ys, Xs, T1s, m2s, i2s, rm2s, rv2s = Z(n, 16), Z(n, 16), Z(n, 16), Z(16), Z(16), rm.clone(), rv.clone()
    sn_s = Zd(nb_tok * 32)
    ops.tok_bn_res_fwd(Xs, T2, s1, g2, be2, m2s, i2s, rm2s, rv2s, Xin)
    ops.attn_fwd(Ys, Xs, keep, params, heads)
    ops.tok_lin_fwd(T1s, sn_s, Ys, W1n, b1n)
    # one launch
    Yf, Xf, T1f, m2f, i2f, rm2f, rv2f = Z(n, 16), Z(n, 16), Z(n, 16), Z(16), Z(16), rm.clone(), rv.clone()
    sn_f = Zd(max(nb_att, nb_tok) * 32)
    ops.attn_block_fwd(Yf, Xf, T2, Xin, s1, g2, be2, m2f, i2f, rm2f, rv2f, keep, params, heads,
                       nxt=(T1f, sn_f, W1n, b1n) if with_next else None)
    torch.cuda.synchronize()
    for a, b_, what in ((Xf, Xs, "residual sum"), (Yf, Ys, "block output"), (m2f, m2s, "mean"), (i2f, i2s, "invstd"),
                        (rm2f, rm2s, "running mean"), (rv2f, rv2s, "running var")):
        assert torch.equal(a, b_), what
    if not with_next:
        assert float(T1f.abs().max()) == 0.0
        return
    assert torch.equal(T1f, T1s)
    tot_s = sn_s.view(nb_tok, 32).sum(0)
    tot_f = sn_f[:nb_att * 32].view(nb_att, 32).sum(0)
    assert float(((tot_f - tot_s).abs() / tot_s.abs().clamp_min(1e-30)).max()) < 1e-11
    Rs, T2s, Rf, T2f = Z(n, 16), Z(n, 16), Z(n, 16), Z(n, 16)
    so_s, so_f = Zd(nb_tok * 32), Zd(nb_tok * 32)
    ops.tok_bn_lin_fwd(Rs, T2s, so_s, T1s, sn_s, g1, be1, Z(16), Z(16), rm.clone(), rv.clone(), W2, b2)
    ops.tok_bn_lin_fwd_from(Rf, T2f, so_f, T1f, sn_f, nb_att, g1, be1, Z(16), Z(16), rm.clone(), rv.clone(), W2, b2)
    close(Rf, Rs.cpu(), 1e-6, "consumer of the fused statistics: R")
    close(T2f, T2s.cpu(), 1e-6, "consumer of the fused statistics: T2")


@pytest.mark.parametrize("n_cells", [512, 37, 2000])
def test_attention_block_backward_in_one_launch(ops, n_cells):
    """cg_attn_block_bwd + cg_tok_bwd_mid_from against cg_attn_ln_bwd, cg_tok_bn_bwd_stats and cg_tok_bwd_mid: the
    attention's input gradient and parameter gradients are the same bits, the sums for the BatchNorm backward agree
    to float64 rounding (another order of additions), what that backward makes of them to float32 rounding."""
    heads, n = 4, n_cells * 16
    Z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    Zd = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")  # noqa: E731
    X, dY = dev(rnd(n, 16, seed=1)), dev(rnd(n, 16, seed=4))
    T2, R, T1 = dev(rnd(n, 16, seed=5)), dev(rnd(n, 16, seed=6).abs()), dev(rnd(n, 16, seed=7))
    m2, i2, m1, i1 = dev(rnd(16, seed=8)), dev(rnd(16, seed=9).abs() + 0.5), dev(rnd(16, seed=10)), dev(rnd(16, seed=11).abs() + 0.5)
    g2, W2 = dev(rnd(16, seed=12) + 1.5), dev(rnd(16, 16, seed=13) * 0.4)
    params = rnd(ATTN_NPARAM, seed=2) * 0.3
    params[1088:1104] += 1.0
    params = dev(params)
    keep = dev((torch.rand(n_cells * heads, 16, 16, generator=torch.Generator().manual_seed(3)) < 0.9).float() / 0.9)
    nb_tok, nb_att = ops.tok_blocks(n), ops.attn_bwd_blocks(n_cells)
    ws = torch.empty(ops.attn_bwd_ws(n_cells)).cuda()
    out = {}
    for form in ("separate", "fused"):
        dX, dp = Z(n, 16), torch.ones(ATTN_NPARAM).cuda()
        s3, s4 = Zd(max(nb_tok, nb_att) * 32), Zd(nb_tok * 32)
        G1, wp2, dg2, db2 = Z(n, 16), Z(nb_tok * 256), Z(16), Z(16)
        if form == "separate":
            ops.attn_bwd(dX, dp, ws, dY, X, keep, params, heads, True)
            ops.tok_bn_bwd_stats(s3, dX, T2, m2, i2)
            ops.tok_bwd_mid(G1, s4, wp2, dg2, db2, dX, T2, R, T1, s3, m2, i2, g2, W2, m1, i1)
            tot = s3[:nb_tok * 32].view(nb_tok, 32).sum(0)
        else:
            ops.attn_block_bwd(dX, dp, ws, dY, X, keep, params, heads, True, T2, m2, i2, s3)
            ops.tok_bwd_mid_from(G1, s4, wp2, dg2, db2, dX, T2, R, T1, s3, nb_att, m2, i2, g2, W2, m1, i1)
            tot = s3[:nb_att * 32].view(nb_att, 32).sum(0)
        torch.cuda.synchronize()
        out[form] = dict(dX=dX, dp=dp, tot=tot, G1=G1, wp2=wp2.view(nb_tok, 256).sum(0), dg2=dg2, db2=db2,
                         s4=s4.view(nb_tok, 32).sum(0))
    a, b_ = out["fused"], out["separate"]
    assert torch.equal(a["dX"], b_["dX"]) and torch.equal(a["dp"], b_["dp"])
    assert float(((a["tot"] - b_["tot"]).abs() / b_["tot"].abs().clamp_min(1e-30)).max()) < 1e-10
    for k in ("G1", "wp2", "dg2", "db2", "s4"):
        close(a[k], b_[k].cpu(), 2e-6, f"consumer of the fused sums: {k}")


def test_merged_leaf_launches(ops):
    """cg_colsum3 and cg_reduce_sum_seg give the bits of the separate calls they stand for; cg_embed_gather writes
    zeros for a negative label."""
    X = [dev(rnd(1024, 512, seed=1)), dev(rnd(1024, 1024, seed=2)), dev(rnd(512, 256, seed=3)), dev(rnd(37, 50, seed=4))]
    base = [dev(rnd(x.shape[1], seed=9)) for x in X]
    sep, mul = [b_.clone() for b_ in base], [b_.clone() for b_ in base]
    for o, x, k in zip(sep, X, range(4)):
        ops.colsum(o, x, 1.0 + k, k != 2)
    ops.colsum_multi([(o, x, 1.0 + k, k != 2) for o, x, k in zip(mul, X, range(4))])
    for a, b_ in zip(mul, sep):
        assert torch.equal(a, b_)
    from cycle_gan_rna_amd.ops import MAX_PARTS
    val = dev(rnd(3 * 512, 1, seed=5))
    reds = []
    for form in range(2):
        red = (torch.zeros(4 * MAX_PARTS).cuda(), torch.zeros(4, dtype=torch.int32).cuda())
        if form == 0:
            for k in range(3):
                ops.reduce_sum(red, 1 + k, val[k * 512:(k + 1) * 512])
        else:
            ops.reduce_sum_seg(red, 1, val, 3)
        reds.append(red)
    assert torch.equal(reds[0][0], reds[1][0]) and torch.equal(reds[0][1], reds[1][1]) and int(reds[0][1][1]) > 0
    table = dev(rnd(12, 12, seed=6))
    lab = torch.tensor([[3], [-1], [11], [0], [-1]]).cuda()
    out = torch.full((5, 12), 7.0).cuda()
    ops.embed_gather(out, table, lab)
    exp = torch.empty(5, 12)
    cpu.embed_gather(exp, table.cpu(), lab.cpu())
    assert torch.equal(out.cpu(), exp) and float(out[1].abs().max()) == 0.0 and float(out[4].abs().max()) == 0.0


def test_penalty_row_terms_as_partial_sums(ops):
    """cg_gp_penalty_parts: the row terms land in the reduction slot (one partial per row, the count set), V is the V
    of cg_gp_penalty."""
    from cycle_gan_rna_amd.ops import MAX_PARTS
    M, G = 512, 5000
    ux = dev(rnd(M, G, seed=9) * 0.05)
    V0, rp = torch.empty(M, G).cuda(), torch.empty(M).cuda()
    ops.gp_penalty(V0, rp, ux, 13.0)
    red = (torch.zeros(3 * MAX_PARTS).cuda(), torch.zeros(3, dtype=torch.int32).cuda())
    V1 = torch.empty(M, G).cuda()
    ops.gp_penalty_parts(V1, red, 2, ux, 13.0)
    assert torch.equal(V0, V1) and torch.equal(red[0][2 * MAX_PARTS:2 * MAX_PARTS + M], rp)
    assert red[1].tolist() == [0, 0, M] and float(red[0][:2 * MAX_PARTS].abs().max()) == 0.0


def test_copy_many_in_one_launch(ops):
    """cg_copy4: contiguous copies of different sizes and element types in one launch, 16-byte pieces or 4-byte words
    by alignment; five pairs take two launches."""
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.rand(3 * 512 * 500 + 8, device="cuda", generator=g)
    srcs = [base[:512 * 500].view(512, 500), base[512 * 500 + 1:512 * 500 + 1 + 300 * 7].view(300, 7),      # odd offset
            torch.randint(0, 12, (512, 1), device="cuda", generator=g), base[2 * 512 * 500:2 * 512 * 500 + 4],
            torch.rand(1, 1, device="cuda", generator=g)]
    dsts = [torch.zeros_like(s_) for s_ in srcs]
    ops.copy_many(list(zip(dsts, srcs)))
    for d, s_ in zip(dsts, srcs):
        assert torch.equal(d, s_)
