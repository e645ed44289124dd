"""cg_bn_fwd_static (the entry point that allocates nothing: its column-sum scratch is an array inside the
library's device image) against cg_bn_fwd with a caller workspace, through the C ABI: two calls in a row, the
second on a smaller input, bit for bit (same kernels); plus the size limit."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _run(lib, static, M, N, seed, check_alloc=True):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = (torch.randn(M, N, generator=g) * 2 + 0.5).to(dev)
    gamma, beta = torch.rand(N, generator=g).add(0.5).to(dev), torch.randn(N, generator=g).to(dev)
    mask = (torch.rand(M, N, generator=g) < 0.7).float().div(0.7).to(dev)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    Y, pm = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    mean, invstd = torch.empty(N, device=dev), torch.empty(N, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    common = (_p(Y), N, _p(X), N, M, N, _p(gamma), _p(beta), 1e-5, _p(mean), _p(invstd), _p(rm), _p(rv), 0.1, 0.2,
              _p(mask), N, _p(pm), N, None, 0)
    if static:
        torch.cuda.synchronize()
        before = (torch.cuda.memory_reserved(), torch.cuda.mem_get_info()[0])
        rc = lib.cg_bn_fwd_static(*common, st)
        torch.cuda.synchronize()
        # nothing was allocated, neither through torch nor behind its back (free device memory as the driver sees it)
        assert not check_alloc or (torch.cuda.memory_reserved(), torch.cuda.mem_get_info()[0]) == before
    else:
        need = lib.cg_bn_workspace_bytes(M, N)
        ws = torch.empty(need // 8 + 1, dtype=torch.float64, device=dev)
        rc = lib.cg_bn_fwd(*common, _p(ws), ws.numel() * 8, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return [t.cpu() for t in (Y, pm, mean, invstd, rm, rv)], X.cpu(), gamma.cpu(), beta.cpu(), mask.cpu()


def test_static_scratch_entry_point_matches_workspace_entry_point():
    from cycle_gan_rna_amd import _lib
    lib = _lib.load_ctypes()
    # the first call on a device looks up the array's address and loads the kernels (the runtime's own set-up, made
    # outside any capture); every later call is checked for allocating nothing
    _run(lib, True, 640, 200, 0, check_alloc=False)
    # both shapes take the two-launch form that uses the scratch (more than 512 rows, or fewer than 256 columns);
    # the second call is smaller, so stale sums of the first stay in the array beside the fresh ones
    # (the third shape takes the one-launch form, which keeps its sums in LDS and does not touch the array)
    for M, N, seed in ((2048, 1000, 1), (640, 200, 2), (512, 512, 3)):
        got, X, gamma, beta, mask = _run(lib, True, M, N, seed)
        exp, *_ = _run(lib, False, M, N, seed)
        for a, b in zip(got, exp):
            assert torch.equal(a, b)
        # and against torch on the CPU (float64), to the float32 bar of the other BatchNorm tests
        xd = X.double()
        mu, var = xd.mean(0), xd.var(0, unbiased=False)
        h = (xd - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
        ref = h * torch.where(h > 0, 1.0, 0.2) * mask.double()
        assert float((got[0].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6
    # too many columns for the array: refused, nothing launched
    N = 50000
    dev = torch.device("cuda:0")
    X = torch.zeros(8, N, device=dev)
    v = torch.ones(N, device=dev)
    out = torch.empty(N, device=dev)
    rc = lib.cg_bn_fwd_static(_p(X), N, _p(X), N, 8, N, _p(v), _p(v), 1e-5, _p(out), _p(out), None, None, 0.1, 0.0,
                              None, 0, None, 0, None, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -3
