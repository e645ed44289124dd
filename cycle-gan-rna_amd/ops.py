"""Tensor-level wrappers over the C ABI of libcgrna.so.

Each method takes torch tensors that already live in HBM, checks shapes on the
host (a kernel is never launched on operands that do not match what its grid
assumes), and enqueues one kernel on torch's current HIP stream.  torch is only
the owner of the memory and the stream here; no torch arithmetic runs in this
file, and there is no fallback when the library or the GPU is missing.
"""
from __future__ import annotations

import bisect
import ctypes as C
import os
import weakref

import torch

from . import _lib

EPI_NONE, EPI_LEAKY, EPI_MUL = 0, 1, 2
MAX_PARTS = 1024
ATTN_NPARAM = 48 * 16 + 48 + 16 * 16 + 16 + 16 + 16


def attn_nparam(E):
    """Floats in one attention parameter block for token width E: in_proj (3E x E + 3E), out_proj (E x E + E),
    LayerNorm (2E)."""
    return 4 * E * E + 6 * E


def _ptr(t):
    """Device address as an integer (0 for None)."""
    return t.data_ptr() if t is not None else 0


def _ld(t):
    """Leading dimension of a 2-D row-major view (or of a vector seen as one row)."""
    if t is None:
        return 0
    if t.dim() == 1:
        return t.shape[0]
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _chk(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype, (t.device, t.dtype)


class HipOps:
    """The product backend: every method is one or two HIP kernel launches."""

    name = "hip"
    WS_COUNTER_FLOATS = 16384 // 4
    # cg_gemm `dtype` flag: the workspace ends with tile arrival counters, split-K products combine their slabs in the
    # same launch (CGRNA_GEMM_SEPARATE_REDUCE in the environment keeps the second launch)
    WS_TILE_COUNTERS = 8

    def __init__(self, device=None, precision="fp32"):
        """precision: 'fp32' - every product on the exact float32 MFMA;
        'bf16' - the large dense-layer products (forward, input gradient, weight gradient with
        M, N >= 128 and K >= 256) round their operands to bf16 and use the bf16 MFMA with float32
        accumulation; the 16-wide token layers, critic head and embedding products stay float32."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.precision = precision
        if not torch.cuda.is_available():
            raise RuntimeError("cycle-gan-rna_amd needs a HIP device (torch.cuda.is_available() is False)")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:0")
        self._stream_token = -((self.device.index or 0) + 1)
        self._ext = not isinstance(self.lib, C.CDLL)
        self._gemm_events = None
        # scratch shared by the kernels that need one (split-K partials, BatchNorm chunk sums); kernels on
        # one stream run in order, so one buffer per use serves every call
        # (one pair per stream, because two streams may run kernels at the same time)
        self._scratch = {}
        # float32 arenas that have a bf16 shadow: (first byte, one past the last byte, shadow tensor)
        self._shadows = []
        self._shadow_los = []
        self._retired_scratch = []

    def _ws(self):
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._scratch.get(key)
        if ws is None:
            ws = [torch.empty((32 << 20) + self.WS_COUNTER_FLOATS, dtype=torch.float32, device=self.device),     # 128 MiB split-K slabs + counters
                  torch.empty(1 << 20, dtype=torch.float64, device=self.device)]      # 8 MiB BatchNorm chunk sums
            if os.environ.get("CGRNA_POISON_SCRATCH"):      # debugging aid: a kernel that reads scratch it did not write shows as NaN
                ws[0].fill_(float("nan"))
                ws[1].fill_(float("nan"))
            # the slab scratch ends with the tile arrival counters of the split-K products (CG_WS_TILE_COUNTERS): zero
            # once, every product leaves them zero
            ws[0][-self.WS_COUNTER_FLOATS:].zero_()
            self._scratch[key] = ws
        return ws

    def stream(self, role, scratch=True):
        """The side stream of this device with the given role ("side0", "ahead", "capture-main", "comm" ...), made at
        first use (with scratch for the kernels launched on it, unless scratch=False) and kept for the life of the
        process: every engine built on these ops runs its branches on the same few streams.  (A fresh
        torch.cuda.Stream per engine walks through torch's pool of 32 streams, and each new handle would get scratch
        of its own and allocator pools of its own.)"""
        if not hasattr(self, "_streams"):
            self._streams = {}
        st = self._streams.get(role)
        if st is None:
            st = self._streams[role] = torch.cuda.Stream(device=self.device)
            if scratch:
                self.prepare_stream(st)
        return st

    def prepare_stream(self, stream):
        """Allocate this stream's scratch now (no allocation may happen during graph capture)."""
        with torch.cuda.stream(stream):
            self._ws()

    def _bnws(self, M, N, nseg=1):
        ws = self._ws()
        need = self.lib.cg_bn_workspace_bytes(M, N) * nseg
        if need > ws[1].numel() * 8:
            # A larger buffer replaces this stream's chunk-sum scratch.  A HIP graph captured earlier on this stream
            # holds the old buffer's address, so the old buffer is kept for the life of the process instead of
            # going back to the allocator; and nothing may be allocated while a capture is running.
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("BatchNorm scratch of %d bytes is needed inside a graph capture: run the step "
                                   "once eagerly at this size first" % need)
            self._retired_scratch.append(ws[1])
            ws[1] = torch.empty(need // 8 + 1, dtype=torch.float64, device=self.device)
        return ws[1].data_ptr(), ws[1].numel() * 8

    # -- hardware queues and forked HIP graphs --------------------------------------
    @staticmethod
    def forked_graphs_supported():
        """(ok, reason): False when the process's HIP settings give more hardware queues than four (slower, see
        below) or leave a forked graph fewer hardware queues than internal streams (GPU_MAX_HW_QUEUES below 4, or more
        graph streams than queues asked for with
        DEBUG_HIP_FORCE_GRAPH_QUEUES): two of its streams then always share the launch stream's queue, which the
        runtime's launch code does not survive: at every launch it uses all but one of the graph's (up to four)
        internal streams, skipping those that share the launch stream's hardware queue, without checking that enough
        are left, and reads past the end of its stream list."""
        def env_int(name, default):
            try:
                return int(os.environ.get(name, default))
            except ValueError:
                return default
        queues = env_int("GPU_MAX_HW_QUEUES", 4)
        graph_streams = env_int("DEBUG_HIP_FORCE_GRAPH_QUEUES", 4)
        if queues < 4 or graph_streams > queues:
            return False, ("GPU_MAX_HW_QUEUES=%d with %d graph streams: a HIP graph with parallel branches cannot be "
                           "launched safely" % (queues, graph_streams))
        if queues > 4:
            # the branches of a replayed graph then land on queues that the hardware time-slices against each other
            # instead of running their kernels back to back: 3.8 / 4.7 / 5.3 ms per iteration with 6 / 8 / 16 queues
            # (5000 genes, batch 512) against 3.5 ms for the same step captured on one stream and 2.55 ms with four
            return False, ("GPU_MAX_HW_QUEUES=%d: with more than four hardware queues the branches of a replayed graph "
                           "are time-sliced against each other and run slower than one stream" % queues)
        return True, ""

    # -- measurement: HIP events around every GEMM launch, on the launch stream ----------
    def begin_gemm_timing(self, max_calls=1 << 14):
        """From here to end_gemm_timing() every GEMM call's kernels carry a pair of HIP events attached to the
        dispatches themselves (cg_gemm_timing_begin): start of the call's first kernel, end of its last one."""
        _lib.check(self.lib.cg_gemm_timing_begin(max_calls), "cg_gemm_timing_begin")
        self._gemm_events = []
        self._gemm_capacity = max_calls

    def end_gemm_timing(self):
        """Returns (launches, total milliseconds, executed FLOP, operand + result bytes) since begin_gemm_timing()."""
        ev, self._gemm_events = self._gemm_events, None
        ms = (C.c_float * self._gemm_capacity)()
        n = self.lib.cg_gemm_timing_end(C.addressof(ms), self._gemm_capacity)
        if n != len(ev):
            raise RuntimeError("cg_gemm_timing_end: %d calls recorded by the library, %d by the wrapper" % (n, len(ev)))
        keep = [i for i, e in enumerate(ev) if ms[i] >= 0 and not e[3]]
        self.last_gemm_detail = [(ev[i][2], float(ms[i]), ev[i][0]) for i in keep]
        return (len(keep), sum(float(ms[i]) for i in keep), sum(ev[i][0] for i in keep), sum(ev[i][1] for i in keep))

    # -- plumbing -------------------------------------------------------------
    def _stream(self):
        # "the current HIP stream of this device", resolved inside the extension (c10::hip::getCurrentHIPStream);
        # the ctypes binding (a differently built library, CGRNA_LIB) is handed the handle itself
        if self._ext:
            return self._stream_token
        return torch.cuda.current_stream(self.device).cuda_stream

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def to_device(self, t):
        return t.to(self.device)

    # -- bf16 shadows -----------------------------------------------------------
    def register_shadow(self, f32_flat, bf16_flat):
        """`bf16_flat` holds (and its owner keeps up to date) the bf16 image of the contiguous float32 tensor
        `f32_flat`: products of the bf16 mode read an operand that is a view into `f32_flat` from the shadow instead
        (half the bytes, no conversion while staging).  The shadow is filled here."""
        assert f32_flat.is_contiguous() and bf16_flat.is_contiguous() and f32_flat.numel() == bf16_flat.numel()
        assert bf16_flat.dtype == torch.bfloat16 and bf16_flat.data_ptr() % 16 == 0 and f32_flat.data_ptr() % 16 == 0
        self._shadows.append((f32_flat.data_ptr(), f32_flat.data_ptr() + 4 * f32_flat.numel(), bf16_flat))
        self._shadows.sort(key=lambda e: e[0])          # address ranges are disjoint: looked up by bisection
        self._shadow_los = [e[0] for e in self._shadows]
        self.refresh_shadow(f32_flat, bf16_flat)
        return f32_flat.data_ptr()

    def unregister_shadow(self, key):
        """Forget the shadow registered for the arena at address `key` (its owner is gone; the address range may be
        handed to another tensor)."""
        self._shadows = [e for e in self._shadows if e[0] != key]
        self._shadow_los = [e[0] for e in self._shadows]

    def refresh_shadow(self, f32_flat, bf16_flat):
        n = f32_flat.numel()
        st = self.lib.cg_to_bf16(_ptr(bf16_flat), n, _ptr(f32_flat), n, 1, n, self._stream())
        _lib.check(st, "cg_to_bf16")

    def _shadow_ptr(self, t):
        """Device address of the bf16 image of the float32 view `t` (same leading dimension, in elements), or None."""
        p = t.data_ptr()
        i = bisect.bisect_right(self._shadow_los, p) - 1
        if i >= 0:
            lo, hi, sh = self._shadows[i]
            if p < hi:
                return sh.data_ptr() + (p - lo) // 2
        return None

    def shadowed(self, *shape):
        """zeros(*shape) whose bf16 image the kernels of this object keep up to date (bf16 mode; plain zeros
        otherwise): the wrappers below that write such a tensor - gemm, bn_fwd*, bn_eval, bn_bwd* - hand the kernel the
        image's address, and gemm reads an operand that has an image from it.  For workspaces that ONLY those
        wrappers write (an activation or a gradient that the next product reads); a tensor written in any other way
        must not be made here, its image would go stale."""
        t = self.zeros(*shape)
        if self.precision == "bf16" and not os.environ.get("CGRNA_NO_ACT_SHADOWS"):
            t16 = self.zeros(*shape, dtype=torch.bfloat16)
            weakref.finalize(t, self.unregister_shadow, self.register_shadow(t.view(-1), t16.view(-1)))
        return t

    def copy(self, out, x):
        """out <- x (2-D), keeping the bf16 image of `out` up to date when it has one."""
        c16, _ = self._twin(out)
        if not c16:
            out.copy_(x)
            return
        M, N = out.shape
        assert x.shape == out.shape
        _lib.check(self.lib.cg_copy_bf16(_ptr(out), c16, _ld(out), _ptr(x), _ld(x), M, N, self._stream()), "cg_copy_bf16")

    def zero(self, out):
        """out <- 0 (2-D), keeping the bf16 image of `out` up to date when it has one."""
        c16, _ = self._twin(out)
        if not c16:
            out.zero_()
            return
        M, N = out.shape
        z = getattr(self, "_zero_src", None)
        if z is None or z.numel() < M * N:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("zero(): run the step once eagerly at this size before capturing it")
            if z is not None:
                self._retired_scratch.append(z)         # a captured graph may hold its address
            z = self._zero_src = torch.zeros(M * N, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.cg_copy_bf16(_ptr(out), c16, _ld(out), _ptr(z), N, M, N, self._stream()), "cg_copy_bf16")

    def image_of(self, x, buf16):
        """A bf16 image of the float32 matrix `x` for gemm(..., a16= / b16=): the one the backend already keeps if `x`
        lives in a shadowed workspace, else `x` rounded into the caller's bf16 buffer `buf16` (one launch).  Returns
        the device address (same leading dimension as `x`, in elements), or None outside bf16 mode."""
        if self.precision != "bf16" or buf16 is None:
            return None
        sp = self._shadow_ptr(x) if self._shadows else None
        if sp is not None:
            return sp
        M, N = x.shape
        assert buf16.dtype == torch.bfloat16 and buf16.is_contiguous() and buf16.numel() >= M * _ld(x)
        _lib.check(self.lib.cg_to_bf16(_ptr(buf16), _ld(x), _ptr(x), _ld(x), M, N, self._stream()), "cg_to_bf16")
        return buf16.data_ptr()

    def _twin(self, t):
        """(address of the bf16 image of the output view `t`, its leading dimension) for a kernel's bf16-output
        argument; (0, 0) when `t` has no image."""
        if not self._shadows:
            return 0, 0
        sp = self._shadow_ptr(t)
        return (sp, _ld(t)) if sp is not None else (0, 0)

    # -- GEMM -----------------------------------------------------------------
    def gemm(self, Cm, A, B, transA=False, transB=False, bias=None, epi=EPI_NONE, E=None, aux=None, slope=0.0,
             accumulate=False, dtype=None, keep_slabs=False, a16=None, b16=None):
        """keep_slabs (no epilogue, no accumulation): if the product splits K, leave the partial sums in this
        stream's scratch for the consumer to add (bn_fwd_slabs) and return their number; 1 = Cm holds the result.
        a16 / b16: device address of a bf16 image of A / B (image_of), used when the product runs on the bf16 MFMA."""
        _chk(Cm), _chk(A), _chk(B)
        M, N = Cm.shape
        K = A.shape[0] if transA else A.shape[1]
        assert (A.shape[1] if transA else A.shape[0]) == M, (A.shape, Cm.shape, transA)
        assert (B.shape[0] if transB else B.shape[1]) == N and (B.shape[1] if transB else B.shape[0]) == K, \
            (A.shape, B.shape, Cm.shape, transA, transB)
        if bias is not None:
            assert bias.numel() == N
        if E is not None:
            assert tuple(E.shape) == (M, N)
        if aux is not None:
            assert tuple(aux.shape) == (M, N)
        if dtype is None:
            dtype = 1 if (self.precision == "bf16" and M >= 128 and N >= 128 and K >= 256) else 0
        pA, pB = _ptr(A), _ptr(B)
        if dtype == 1 and (self._shadows or a16 is not None or b16 is not None):
            # an operand that lives in an arena with a bf16 shadow (the weights) is read from the shadow; the library's
            # conditions: 8-byte aligned rows, even contiguous extent
            sa = a16 if a16 is not None else self._shadow_ptr(A)
            sb = b16 if b16 is not None else self._shadow_ptr(B)
            if sb is not None and sb % 8 == 0 and _ld(B) % 4 == 0 and B.shape[1] % 2 == 0 and \
                    (A.data_ptr() % 16 == 0 and _ld(A) % 4 == 0):
                pB, dtype = sb, dtype | 2
            if sa is not None and sa % 8 == 0 and _ld(A) % 4 == 0 and A.shape[1] % 2 == 0 and \
                    (dtype & 2 or (B.data_ptr() % 16 == 0 and _ld(B) % 4 == 0)):
                pA, dtype = sa, dtype | 4
        n_slabs = None
        if keep_slabs:
            assert epi == EPI_NONE and E is None and aux is None and not accumulate
            n_slabs = C.c_int(1)
            st = self.lib.cg_gemm_slabs(_ptr(Cm), _ld(Cm), pA, _ld(A), pB, _ld(B), M, N, K, int(transA), int(transB),
                                        dtype, _ptr(bias), _ptr(self._ws()[0]),
                                        (self._ws()[0].numel() - self.WS_COUNTER_FLOATS) * 4,
                                        C.addressof(n_slabs), self._stream())
        else:
            c16, _ = self._twin(Cm)
            st = self.lib.cg_gemm_ex(_ptr(Cm), _ld(Cm), pA, _ld(A), pB, _ld(B), M, N, K, int(transA), int(transB),
                                     dtype | self.WS_TILE_COUNTERS, _ptr(bias), epi, _ptr(E), _ld(E), _ptr(aux), _ld(aux), float(slope),
                                     int(accumulate), c16, _ptr(self._ws()[0]), self._ws()[0].numel() * 4,
                                     self._stream())
        if self._gemm_events is not None:       # the library has attached an event pair to this call's kernels
            elem = lambda flag: 2.0 if dtype & flag else 4.0  # noqa: E731
            self._gemm_events.append((2.0 * M * N * K,
                                      elem(4) * M * K + elem(2) * K * N + 4.0 * M * N * (2 if accumulate else 1),
                                      (M, N, K, int(transA), int(transB), int(accumulate), dtype & 7), self._gemm_skip))
        _lib.check(st, "cg_gemm")
        return n_slabs.value if n_slabs is not None else None

    _gemm_skip = False

    def gemm_untimed(self, *args, **kw):
        """gemm() that end_gemm_timing() leaves out of its totals (filler work)."""
        self._gemm_skip = True
        try:
            self.gemm(*args, **kw)
        finally:
            self._gemm_skip = False

    # -- BatchNorm --------------------------------------------------------------
    def bn_fwd(self, Y, X, gamma, beta, mean, invstd, rmean, rvar, slope, mask=None, pm=None, eps=1e-5,
               momentum=0.1):
        M, N = X.shape
        assert Y.shape == X.shape and gamma.numel() == N and mean.numel() == N and invstd.numel() == N
        st = self.lib.cg_bn_fwd(_ptr(Y), _ld(Y), _ptr(X), _ld(X), M, N, _ptr(gamma), _ptr(beta), eps, _ptr(mean),
                                _ptr(invstd), _ptr(rmean), _ptr(rvar), momentum, float(slope), _ptr(mask), _ld(mask),
                                _ptr(pm), _ld(pm), *self._twin(Y), *self._bnws(M, N), self._stream())
        _lib.check(st, "cg_bn_fwd")

    def bn_slabs_ok(self, M, N):
        """Can bn_fwd_slabs take an M x N input?  (Only the one-launch BatchNorm reads slabs.)"""
        return bool(self.lib.cg_bn_fwd_slabs_ok(M, N))

    def bn_fwd_slabs(self, Y, X, n_slabs, x_bias, gamma, beta, mean, invstd, rmean, rvar, slope, mask=None, pm=None,
                     eps=1e-5, momentum=0.1):
        """bn_fwd whose input is still in the n_slabs partial sums that gemm(..., keep_slabs=True) just left in this
        stream's scratch: X receives bias + slabs, Y the normalised activation."""
        M, N = X.shape
        assert Y.shape == X.shape and gamma.numel() == N and mean.numel() == N and invstd.numel() == N
        ws = self._ws()[0]
        assert n_slabs >= 1 and n_slabs * M * N <= ws.numel() - self.WS_COUNTER_FLOATS
        st = self.lib.cg_bn_fwd_slabs(_ptr(Y), _ld(Y), _ptr(X), _ld(X), _ptr(ws), n_slabs, _ptr(x_bias), M, N,
                                      _ptr(gamma), _ptr(beta), eps, _ptr(mean), _ptr(invstd), _ptr(rmean), _ptr(rvar),
                                      momentum, float(slope), _ptr(mask), _ld(mask), _ptr(pm), _ld(pm), *self._twin(Y),
                                      self._stream())
        _lib.check(st, "cg_bn_fwd_slabs")

    @staticmethod
    def _order_code(order, nseg):
        order = tuple(range(nseg)) if order is None else tuple(order)
        assert sorted(order) == list(range(nseg)), order
        return sum(s << (4 * k) for k, s in enumerate(order))

    def bn_fwd_seg(self, Y, X, nseg, order, gamma, beta, mean, invstd, rmean, rvar, slope, mask=None, pm=None,
                   eps=1e-5, momentum=0.1):
        """bn_fwd on nseg equal row segments, each with its own batch statistics; mean/invstd: (nseg, N)."""
        M, N = X.shape
        assert M % nseg == 0 and Y.shape == X.shape and gamma.numel() == N
        assert mean.numel() >= nseg * N and invstd.numel() >= nseg * N and mean.is_contiguous() and invstd.is_contiguous()
        st = self.lib.cg_bn_fwd_seg(_ptr(Y), _ld(Y), _ptr(X), _ld(X), M // nseg, nseg, self._order_code(order, nseg), N,
                                    _ptr(gamma), _ptr(beta), eps, _ptr(mean), _ptr(invstd), _ptr(rmean), _ptr(rvar),
                                    momentum, float(slope), _ptr(mask), _ld(mask), _ptr(pm), _ld(pm), *self._twin(Y),
                                    *self._bnws(M // nseg, N, nseg), self._stream())
        _lib.check(st, "cg_bn_fwd_seg")

    def bn_bwd_seg(self, dX, dY, X, Y, PM, nseg, mean, invstd, gamma, dgamma, dbeta, accumulate, slope):
        M, N = X.shape
        assert M % nseg == 0 and dX.shape == X.shape and dY.shape == X.shape
        assert mean.numel() >= nseg * N and invstd.numel() >= nseg * N
        st = self.lib.cg_bn_bwd_seg(_ptr(dX), _ld(dX), _ptr(dY), _ld(dY), _ptr(X), _ld(X), _ptr(Y), _ld(Y), _ptr(PM),
                                    _ld(PM), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(dgamma), _ptr(dbeta),
                                    int(accumulate), 0, 0, float(slope), M // nseg, nseg, N, *self._twin(dX),
                                    *self._bnws(M // nseg, N, nseg), self._stream())
        _lib.check(st, "cg_bn_bwd_seg")

    def bn_eval(self, Y, X, gamma, beta, rmean, rvar, slope, eps=1e-5):
        M, N = X.shape
        st = self.lib.cg_bn_eval(_ptr(Y), _ld(Y), _ptr(X), _ld(X), M, N, _ptr(gamma), _ptr(beta), _ptr(rmean),
                                 _ptr(rvar), eps, float(slope), *self._twin(Y), self._stream())
        _lib.check(st, "cg_bn_eval")

    def bn_running_update(self, rmean, rvar, mean, invstd, M, eps=1e-5, momentum=0.1):
        """The running-statistics update of one more train-mode call on the batch whose statistics bn_fwd stored."""
        N = mean.numel()
        assert rmean.numel() == N and rvar.numel() == N and invstd.numel() == N
        st = self.lib.cg_bn_running_update(_ptr(rmean), _ptr(rvar), _ptr(mean), _ptr(invstd), M, N, eps, momentum,
                                           self._stream())
        _lib.check(st, "cg_bn_running_update")

    def bn_bwd(self, dX, dY, X, Y, PM, mean, invstd, gamma, dgamma, dbeta, accumulate, slope, gbar=None, gxbar=None):
        M, N = X.shape
        assert dX.shape == X.shape and dY.shape == X.shape
        st = self.lib.cg_bn_bwd(_ptr(dX), _ld(dX), _ptr(dY), _ld(dY), _ptr(X), _ld(X), _ptr(Y), _ld(Y), _ptr(PM),
                                _ld(PM), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(dgamma), _ptr(dbeta),
                                int(accumulate), _ptr(gbar), _ptr(gxbar), float(slope), M, N, *self._twin(dX),
                                *self._bnws(M, N), self._stream())
        _lib.check(st, "cg_bn_bwd")

    def bn_bwd_slabs_ok(self, M, N):
        """Can bn_bwd_slabs take an M x N gradient?  (Only the one-launch BatchNorm backward reads slabs.)"""
        return bool(self.lib.cg_bn_bwd_slabs_ok(M, N))

    def bn_bwd_slabs(self, dX, n_slabs, X, Y, PM, mean, invstd, gamma, dgamma, dbeta, accumulate, slope):
        """bn_bwd whose incoming gradient is still in the n_slabs partial sums that gemm(..., keep_slabs=True) just
        left in this stream's scratch."""
        M, N = X.shape
        assert dX.shape == X.shape
        ws = self._ws()[0]
        assert n_slabs >= 1 and n_slabs * M * N <= ws.numel() - self.WS_COUNTER_FLOATS
        st = self.lib.cg_bn_bwd_slabs(_ptr(dX), _ld(dX), _ptr(ws), n_slabs, _ptr(X), _ld(X), _ptr(Y), _ld(Y), _ptr(PM),
                                      _ld(PM), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(dgamma), _ptr(dbeta),
                                      int(accumulate), float(slope), M, N, *self._twin(dX), self._stream())
        _lib.check(st, "cg_bn_bwd_slabs")

    def bn_bwd_bwd(self, Vu, Vh, dgamma, accumulate, W, U1, X, mean, invstd, gamma, ubar, mbar, PM=None):
        """PM: Vu is stored multiplied by it (see cg_bn_bwd_bwd)."""
        M, N = X.shape
        assert PM is None or PM.shape == X.shape
        assert Vu.shape == X.shape and Vh.shape == X.shape and W.shape == X.shape and U1.shape == X.shape
        st = self.lib.cg_bn_bwd_bwd(_ptr(Vu), _ld(Vu), _ptr(Vh), _ld(Vh), _ptr(dgamma), int(accumulate), _ptr(W),
                                    _ld(W), _ptr(U1), _ld(U1), _ptr(X), _ld(X), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                    _ptr(ubar), _ptr(mbar), M, N, self._twin(Vh)[0], _ptr(PM), _ld(PM),
                                    *self._bnws(M, N), self._stream())
        _lib.check(st, "cg_bn_bwd_bwd")

    # -- elementwise ------------------------------------------------------------
    def axpby(self, out, a, x, b=0.0, y=None):
        M, N = out.shape
        assert x.shape == out.shape and (y is None or y.shape == out.shape)
        st = self.lib.cg_axpby(_ptr(out), _ld(out), float(a), _ptr(x), _ld(x), float(b), _ptr(y), _ld(y), M, N,
                               self._stream())
        _lib.check(st, "cg_axpby")

    def mul(self, out, x, y):
        M, N = out.shape
        assert x.shape == out.shape and y.shape == out.shape
        st = self.lib.cg_mul(_ptr(out), _ld(out), _ptr(x), _ld(x), _ptr(y), _ld(y), M, N, self._stream())
        _lib.check(st, "cg_mul")

    def fill_coef(self, out, coef, idx, scale):
        M, N = out.shape
        st = self.lib.cg_fill_coef(_ptr(out), _ld(out), M, N, _ptr(coef), idx, float(scale), self._stream())
        _lib.check(st, "cg_fill_coef")

    def embed_gather(self, out, table, labels):
        M = out.shape[0]
        n_types, E = table.shape
        _chk(labels, torch.int64)
        assert out.shape[1] == E and labels.shape[0] == M and table.is_contiguous()
        st = self.lib.cg_embed_gather(_ptr(out), _ld(out), _ptr(table), E, _ptr(labels), labels.stride(0), M, n_types,
                                      self._twin(out)[0], self._stream())
        _lib.check(st, "cg_embed_gather")

    def embed_scatter(self, dtable, dout, labels, accumulate):
        n_types, E = dtable.shape
        M = dout.shape[0]
        assert dout.shape[1] == E and labels.shape[0] == M and dtable.is_contiguous()
        st = self.lib.cg_embed_scatter(_ptr(dtable), E, _ptr(dout), _ld(dout), _ptr(labels), labels.stride(0), M,
                                       n_types, int(accumulate), self._stream())
        _lib.check(st, "cg_embed_scatter")

    LABEL_COLSUM_MAX_TYPES = 32

    def label_colsum(self, out, X, labels, scale=1.0, accumulate=False, minus=None):
        """out[t] (+)= scale * sum of the rows of X whose label is t (labels (M, 1) or (M,) int64; others skipped);
        minus: a second label per row at which the row is subtracted."""
        n_types, N = out.shape
        M = X.shape[0]
        assert X.shape[1] == N and labels.shape[0] == M and out.is_contiguous() and labels.dtype == torch.int64
        assert minus is None or (minus.shape[0] == M and minus.dtype == torch.int64)
        st = self.lib.cg_label_colsum(_ptr(out), _ptr(X), _ld(X), _ptr(labels), labels.stride(0), M, N, n_types,
                                      float(scale), int(accumulate), _ptr(minus), minus.stride(0) if minus is not None else 0,
                                      self._stream())
        _lib.check(st, "cg_label_colsum")

    def reparam_fwd(self, z, ml, eps):
        M, L = z.shape
        assert ml.shape == (M, 2 * L) and eps.shape == (M, L) and eps.is_contiguous()
        st = self.lib.cg_reparam_fwd(_ptr(z), _ld(z), _ptr(ml), _ld(ml), _ptr(eps), M, L, self._stream())
        _lib.check(st, "cg_reparam_fwd")

    def reparam_bwd(self, dml, dz, ml, eps, coef, kl_idx, kl_weight):
        M, L = dz.shape
        assert dml.shape == (M, 2 * L) and ml.shape == (M, 2 * L) and eps.shape == (M, L)
        st = self.lib.cg_reparam_bwd(_ptr(dml), _ld(dml), _ptr(dz), _ld(dz), _ptr(ml), _ld(ml), _ptr(eps), _ptr(coef),
                                     kl_idx, float(kl_weight), M, L, self._stream())
        _lib.check(st, "cg_reparam_bwd")

    def libsize_fwd(self, Y, X, rowsum, lib):
        M, N = X.shape
        assert Y.shape == X.shape and rowsum.numel() == M
        st = self.lib.cg_libsize_fwd(_ptr(Y), _ld(Y), _ptr(X), _ld(X), _ptr(rowsum), float(lib), M, N, self._stream())
        _lib.check(st, "cg_libsize_fwd")

    def libsize_bwd(self, dX, dY, X, rowsum, lib):
        M, N = X.shape
        st = self.lib.cg_libsize_bwd(_ptr(dX), _ld(dX), _ptr(dY), _ld(dY), _ptr(X), _ld(X), _ptr(rowsum), float(lib),
                                     M, N, self._stream())
        _lib.check(st, "cg_libsize_bwd")

    def gp_interp(self, out, real, fake, alpha):
        M, G = out.shape
        assert real.shape == out.shape and fake.shape == out.shape and alpha.numel() == M
        st = self.lib.cg_gp_interp(_ptr(out), _ld(out), _ptr(real), _ld(real), _ptr(fake), _ld(fake), _ptr(alpha), M,
                                   G, self._stream())
        _lib.check(st, "cg_gp_interp")

    def critic_inputs(self, out, lab_out, real, fake, alpha, real_lab, fake_lab, interp_block=2):
        """out <- [real; fake; (interp_block - 2 row blocks left alone); interpolates],
        lab_out, the same row blocks <- [real_lab; fake_lab; (-1 ...); fake_lab]."""
        M, G = real.shape
        assert out.shape == ((interp_block + 1) * M, G) and fake.shape == (M, G) and alpha.numel() == M
        n_cov = 0
        if lab_out is not None:
            n_cov = real_lab.shape[1]
            _chk(lab_out, torch.int64), _chk(real_lab, torch.int64), _chk(fake_lab, torch.int64)
            assert lab_out.shape == ((interp_block + 1) * M, n_cov) and real_lab.shape == (M, n_cov) and \
                fake_lab.shape == (M, n_cov)
            assert lab_out.is_contiguous() and real_lab.is_contiguous() and fake_lab.is_contiguous()
        st = self.lib.cg_critic_inputs(_ptr(out), _ld(out), _ptr(real), _ld(real), _ptr(fake), _ld(fake), _ptr(alpha), M,
                                       G, _ptr(lab_out), _ptr(real_lab), _ptr(fake_lab), n_cov, int(interp_block),
                                       self._twin(out)[0], self._stream())
        _lib.check(st, "cg_critic_inputs")

    def interp_hidden(self, H, M, alpha, emb, We):
        """H[2M:3M] <- alpha H[:M] + (1 - alpha) H[M:2M] + alpha (emb[M:2M] - emb[:M]) We^T: the critic's first layer
        on the interpolates from its outputs on the real and the generated batch (emb, We None: no covariates)."""
        N = H.shape[1]
        assert H.shape[0] >= 3 * M and alpha.numel() == M
        E = 0
        if emb is not None:
            E = emb.shape[1]
            assert emb.shape[0] >= 2 * M and tuple(We.shape) == (N, E) and emb.stride(1) == 1 and We.stride(1) == 1
        st = self.lib.cg_interp_hidden(_ptr(H), _ld(H), _ptr(alpha), _ptr(emb), _ld(emb), E, _ptr(We), _ld(We), M, N,
                                       self._stream())
        _lib.check(st, "cg_interp_hidden")

    def gp_combine(self, D, VH, alpha, AV, emb, dE):
        """D[:M] += alpha VH, D[M:] += (1 - alpha) VH (in place), AV <- alpha VH, dE <- emb[M:] - emb[:M] (emb, dE may
        be None): see cg_gp_combine."""
        M, N = VH.shape
        assert D.shape == (2 * M, N) and AV.shape == (M, N) and alpha.numel() == M
        E = 0
        if emb is not None:
            E = emb.shape[1]
            assert emb.shape[0] >= 2 * M and tuple(dE.shape) == (M, E) and emb.stride(1) == 1 and dE.stride(1) == 1
        st = self.lib.cg_gp_combine(_ptr(D), _ld(D), _ptr(VH), _ld(VH), _ptr(alpha), _ptr(AV), _ld(AV), _ptr(emb),
                                    _ld(emb), E, _ptr(dE), _ld(dE), M, N, self._twin(D)[0], self._stream())
        _lib.check(st, "cg_gp_combine")

    def gp_penalty(self, V, rowpen, UX, lambda_gp):
        M, G = UX.shape
        assert V.shape == UX.shape and rowpen.numel() == M
        st = self.lib.cg_gp_penalty(_ptr(V), _ld(V), _ptr(rowpen), _ptr(UX), _ld(UX), float(lambda_gp), M, G,
                                    self._twin(V)[0], self._stream())
        _lib.check(st, "cg_gp_penalty")

    def gp_penalty_parts(self, V, red, slot, UX, lambda_gp):
        """gp_penalty whose row terms go straight into reduction `slot` of `red` (one partial sum per row; at most
        MAX_PARTS rows): no reduce_sum launch for the penalty's mean."""
        M, G = UX.shape
        assert V.shape == UX.shape and M <= MAX_PARTS
        p, n = self._slot(red, slot)
        st = self.lib.cg_gp_penalty_parts(_ptr(V), _ld(V), p, n, _ptr(UX), _ld(UX), float(lambda_gp), M, G,
                                          self._twin(V)[0], self._stream())
        _lib.check(st, "cg_gp_penalty_parts")

    # -- reductions ---------------------------------------------------------------
    # `red` = (parts [n_slots, MAX_PARTS] float32, nparts [n_slots] int32); slot picks the row.
    def _slot(self, red, slot):
        parts, nparts = red
        return (parts.data_ptr() + slot * MAX_PARTS * 4, nparts.data_ptr() + slot * 4)

    def reduce_sum(self, red, slot, X):
        X2 = X if X.dim() == 2 else X.view(1, -1)
        p, n = self._slot(red, slot)
        st = self.lib.cg_reduce_sum(p, n, _ptr(X2), _ld(X2), X2.shape[0], X2.shape[1], self._stream())
        _lib.check(st, "cg_reduce_sum")

    def reduce_l1(self, red, slot, A, B):
        assert A.shape == B.shape
        p, n = self._slot(red, slot)
        st = self.lib.cg_reduce_l1(p, n, _ptr(A), _ld(A), _ptr(B), _ld(B), A.shape[0], A.shape[1], self._stream())
        _lib.check(st, "cg_reduce_l1")

    def reduce_sqdiff(self, red, slot, A, B):
        A2, B2 = (A, B) if A.dim() == 2 else (A.view(1, -1), B.view(1, -1))
        p, n = self._slot(red, slot)
        st = self.lib.cg_reduce_sqdiff(p, n, _ptr(A2), _ld(A2), _ptr(B2), _ld(B2), A2.shape[0], A2.shape[1],
                                       self._stream())
        _lib.check(st, "cg_reduce_sqdiff")

    def reduce_centroid_sq(self, red, slot, A, Cn, labels):
        assert Cn.shape[1] == A.shape[1] and labels.shape[0] == A.shape[0]
        p, n = self._slot(red, slot)
        st = self.lib.cg_reduce_centroid_sq(p, n, _ptr(A), _ld(A), _ptr(Cn), _ld(Cn), _ptr(labels), labels.stride(0),
                                            A.shape[0], A.shape[1], Cn.shape[0], self._stream())
        _lib.check(st, "cg_reduce_centroid_sq")

    def reduce_kl(self, red, slot, ml):
        M, L2 = ml.shape
        p, n = self._slot(red, slot)
        st = self.lib.cg_reduce_kl(p, n, _ptr(ml), _ld(ml), M, L2 // 2, self._stream())
        _lib.check(st, "cg_reduce_kl")

    def colsum(self, out, X, scale=1.0, accumulate=False):
        M, N = X.shape
        assert out.numel() == N
        st = self.lib.cg_colsum(_ptr(out), _ptr(X), _ld(X), M, N, float(scale), int(accumulate), self._stream())
        _lib.check(st, "cg_colsum")

    def copy_many(self, pairs):
        """[(dst, src), ...]: dst <- src for contiguous device tensors of equal size and dtype, four per launch."""
        for i in range(0, len(pairs), 4):
            args = []
            for k in range(4):
                if i + k < len(pairs):
                    dst, src = pairs[i + k]
                    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and \
                        dst.numel() == src.numel() and dst.device == src.device == self.device
                    args += [dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size()]
                else:
                    args += [0, 0, 0]
            _lib.check(self.lib.cg_copy4(*args, self._stream()), "cg_copy4")

    def colsum_multi(self, jobs):
        """[(out, X, scale, accumulate), ...]: colsum of every job, three per launch."""
        for i in range(0, len(jobs), 3):
            args = []
            for k in range(3):
                if i + k < len(jobs):
                    out, X, scale, acc = jobs[i + k]
                    M, N = X.shape
                    assert out.numel() == N
                    args += [_ptr(out), _ptr(X), _ld(X), M, N, float(scale), int(acc)]
                else:
                    args += [0, 0, 0, 0, 0, 0.0, 0]
            _lib.check(self.lib.cg_colsum3(*args, self._stream()), "cg_colsum3")

    def reduce_sum_seg(self, red, slot, X, nseg):
        """reduce_sum of the nseg equal row segments of X into the slots slot .. slot + nseg - 1."""
        assert X.dim() == 2 and X.shape[0] % nseg == 0
        p, n = self._slot(red, slot)
        st = self.lib.cg_reduce_sum_seg(p, n, _ptr(X), _ld(X), X.shape[0] // nseg, nseg, X.shape[1], self._stream())
        _lib.check(st, "cg_reduce_sum_seg")

    def colvar_mean(self, out, ws, X):
        M, N = X.shape
        assert ws.dtype == torch.float64 and ws.numel() >= (N + 31) // 32
        st = self.lib.cg_colvar_mean(_ptr(out), _ptr(ws), _ptr(X), _ld(X), M, N, self._stream())
        _lib.check(st, "cg_colvar_mean")

    def reduce_poisson(self, red, slot, pred, target):
        M, N = pred.shape
        p, n = self._slot(red, slot)
        st = self.lib.cg_reduce_poisson(p, n, _ptr(pred), _ld(pred), _ptr(target), _ld(target), M, N, self._stream())
        _lib.check(st, "cg_reduce_poisson")

    def poisson_bwd(self, dA, pred, target, coef, idx, scale, accumulate=False):
        M, N = pred.shape
        st = self.lib.cg_poisson_bwd(_ptr(dA), _ld(dA), _ptr(pred), _ld(pred), _ptr(target), _ld(target), _ptr(coef),
                                     idx, float(scale), M, N, int(accumulate), self._stream())
        _lib.check(st, "cg_poisson_bwd")

    def l1_bwd(self, dA, A, B, coef, idx, scale, accumulate=False):
        M, N = A.shape
        st = self.lib.cg_l1_bwd(_ptr(dA), _ld(dA), _ptr(A), _ld(A), _ptr(B), _ld(B), _ptr(coef), idx, float(scale), M,
                                N, int(accumulate), self._stream())
        _lib.check(st, "cg_l1_bwd")

    def centroid_bwd(self, dA, A, Cn, labels, coef, idx, scale, accumulate=False):
        M, N = A.shape
        st = self.lib.cg_centroid_bwd(_ptr(dA), _ld(dA), _ptr(A), _ld(A), _ptr(Cn), _ld(Cn), _ptr(labels),
                                      labels.stride(0), _ptr(coef), idx, float(scale), M, N, Cn.shape[0],
                                      int(accumulate), self._stream())
        _lib.check(st, "cg_centroid_bwd")

    def fm_bwd(self, dF, mean_fake, mean_real, coef, idx, scale, accumulate=False):
        M, N = dF.shape
        st = self.lib.cg_fm_bwd(_ptr(dF), _ld(dF), _ptr(mean_fake), _ptr(mean_real), _ptr(coef), idx, float(scale), M,
                                N, int(accumulate), self._stream())
        _lib.check(st, "cg_fm_bwd")

    def gloss_finalize(self, ema, losses, coef, red, desc):
        parts, nparts = red
        d = _lib.GlossDesc()
        for k in range(len(desc["terms"])):          # five terms, or six with the identity term
            terms = desc["terms"][k]
            d.n_terms[k] = len(terms)
            for j, (slot, w) in enumerate(terms):
                d.slot[k][j] = slot
                d.weight[k][j] = w
            d.lambda_[k] = desc["lambda"][k]
        d.ema_alpha = desc["ema_alpha"]
        w = desc.get("weights")
        assert w is None or (w.dtype == torch.float32 and w.numel() == 5 and w.is_contiguous())
        d.weights = _ptr(w)
        assert ema.dtype == torch.float64 and losses.dtype == torch.float64 and losses.numel() >= 12 and coef.numel() >= 6
        st = self.lib.cg_gloss_finalize(_ptr(ema), _ptr(losses), _ptr(coef), _ptr(parts), _ptr(nparts), C.addressof(d),
                                        self._stream())
        _lib.check(st, "cg_gloss_finalize")

    def sum_parts(self, totals, red, slots, scales):
        parts, nparts = red
        assert totals.dtype == torch.float64 and slots.dtype == torch.int32 and scales.dtype == torch.float64
        st = self.lib.cg_sum_parts(_ptr(totals), _ptr(parts), _ptr(nparts), _ptr(slots), _ptr(scales), slots.numel(),
                                   self._stream())
        _lib.check(st, "cg_sum_parts")

    # -- optimizer ------------------------------------------------------------------
    def adam(self, p, g, m, v, vmax, lr_step, beta1, beta2, eps, p16=None):
        """vmax None: plain Adam (no running maximum of v).  p16: bf16 shadow of p, rewritten with the new values."""
        assert lr_step.dtype == torch.float64 and lr_step.numel() == 2
        n = p.numel()
        assert g.numel() == n and m.numel() == n and v.numel() == n and (vmax is None or vmax.numel() == n)
        assert p16 is None or (p16.dtype == torch.bfloat16 and p16.numel() == n and p16.is_contiguous())
        st = self.lib.cg_adam_amsgrad(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vmax), n, _ptr(lr_step), beta1, beta2,
                                      eps, _ptr(p16), self._stream())
        _lib.check(st, "cg_adam_amsgrad")

    # -- latent ResidualBlock on tokens (fused) ------------------------------------------
    def tok_blocks(self, n_tok):
        return self.lib.cg_tok_blocks(n_tok)

    def tok_lin_fwd(self, T1, stat_out, X, W, b):
        n = X.shape[0]
        assert X.shape == (n, 16) and T1.shape == (n, 16) and X.is_contiguous() and T1.is_contiguous()
        assert stat_out.dtype == torch.float64 and stat_out.numel() >= self.tok_blocks(n) * 32
        _lib.check(self.lib.cg_tok_lin_fwd(_ptr(T1), _ptr(stat_out), _ptr(X), _ptr(W), _ptr(b), n, self._stream()),
                   "cg_tok_lin_fwd")

    def tok_bn_lin_fwd(self, R, T2, stat_out, T1, stat_in, gamma, beta, mean, invstd, rmean, rvar, W2, b2, eps=1e-5,
                       momentum=0.1):
        n = T1.shape[0]
        assert all(t.shape == (n, 16) and t.is_contiguous() for t in (R, T2, T1))
        assert stat_out.numel() >= self.tok_blocks(n) * 32 and stat_in.numel() >= self.tok_blocks(n) * 32
        _lib.check(self.lib.cg_tok_bn_lin_fwd(_ptr(R), _ptr(T2), _ptr(stat_out), _ptr(T1), _ptr(stat_in), _ptr(gamma),
                                              _ptr(beta), eps, _ptr(mean), _ptr(invstd), _ptr(rmean), _ptr(rvar),
                                              momentum, _ptr(W2), _ptr(b2), n, self._stream()), "cg_tok_bn_lin_fwd")

    def attn_block_blocks(self, n_cells):
        return self.lib.cg_attn_block_blocks(n_cells)

    def tok_bn_lin_fwd_from(self, R, T2, stat_out, T1, stat_in, n_stat_in, gamma, beta, mean, invstd, rmean, rvar, W2, b2,
                            eps=1e-5, momentum=0.1):
        """tok_bn_lin_fwd whose input statistics are n_stat_in partials (written by attn_block_fwd)."""
        n = T1.shape[0]
        assert all(t.shape == (n, 16) and t.is_contiguous() for t in (R, T2, T1))
        assert stat_out.numel() >= self.tok_blocks(n) * 32 and stat_in.numel() >= n_stat_in * 32
        _lib.check(self.lib.cg_tok_bn_lin_fwd_from(_ptr(R), _ptr(T2), _ptr(stat_out), _ptr(T1), _ptr(stat_in), n_stat_in,
                                                   _ptr(gamma), _ptr(beta), eps, _ptr(mean), _ptr(invstd), _ptr(rmean),
                                                   _ptr(rvar), momentum, _ptr(W2), _ptr(b2), n, self._stream()),
                   "cg_tok_bn_lin_fwd_from")

    def attn_block_ok(self, E, heads):
        """Is there a one-launch form of (BatchNorm + residual, attention + LayerNorm, next block's first Linear)?"""
        return E == 16 and heads == 4 and not os.environ.get("CGRNA_NO_BLOCK_FUSION")

    def attn_block_fwd(self, Y, Xres, T2, Xin, stat_in, gamma, beta, mean, invstd, rmean, rvar, keep, params, heads,
                       nxt=None, eps=1e-5, momentum=0.1):
        """Xres = Xin + bn2(T2); Y = LayerNorm(Xres + attention(Xres)); nxt = (T1, stat_out, W1, b1) of the next block:
        T1 = Y W1^T + b1 and attn_block_blocks(n_cells) partial sums of T1, T1^2 into stat_out."""
        n, E = T2.shape
        n_cells = n // 16
        assert all(t.shape == (n, E) and t.is_contiguous() for t in (Y, Xres, T2, Xin))
        assert params.numel() == attn_nparam(E) and stat_in.numel() >= self.tok_blocks(n) * 32
        assert keep is None or (keep.is_contiguous() and keep.numel() == n_cells * heads * 256)
        T1, so, W1, b1 = nxt if nxt is not None else (None, None, None, None)
        assert nxt is None or (T1.shape == (n, E) and T1.is_contiguous() and so.numel() >= self.attn_block_blocks(n_cells) * 32)
        st = self.lib.cg_attn_block_fwd(_ptr(Y), _ptr(Xres), _ptr(T2), _ptr(Xin), _ptr(stat_in), self.tok_blocks(n),
                                        _ptr(gamma), _ptr(beta), eps, _ptr(mean), _ptr(invstd), _ptr(rmean), _ptr(rvar),
                                        momentum, _ptr(keep), _ptr(params), n_cells, E, heads, _ptr(T1), _ptr(so),
                                        _ptr(W1), _ptr(b1), self._stream())
        _lib.check(st, "cg_attn_block_fwd")

    def tok_bn_res_fwd(self, Y, T2, stat_in, gamma, beta, mean, invstd, rmean, rvar, X, eps=1e-5, momentum=0.1):
        n = T2.shape[0]
        assert all(t.shape == (n, 16) and t.is_contiguous() for t in (Y, T2, X))
        assert stat_in.numel() >= self.tok_blocks(n) * 32
        _lib.check(self.lib.cg_tok_bn_res_fwd(_ptr(Y), _ptr(T2), _ptr(stat_in), _ptr(gamma), _ptr(beta), eps,
                                              _ptr(mean), _ptr(invstd), _ptr(rmean), _ptr(rvar), momentum, _ptr(X), n,
                                              self._stream()), "cg_tok_bn_res_fwd")

    def tok_bn_bwd_stats(self, stat_out, dY, T2, mean, invstd):
        n = dY.shape[0]
        assert all(t.shape == (n, 16) and t.is_contiguous() for t in (dY, T2))
        assert stat_out.numel() >= self.tok_blocks(n) * 32
        _lib.check(self.lib.cg_tok_bn_bwd_stats(_ptr(stat_out), _ptr(dY), _ptr(T2), _ptr(mean), _ptr(invstd), n,
                                                self._stream()), "cg_tok_bn_bwd_stats")

    def tok_bwd_mid(self, G1, stat_out, wpart, dgamma2, dbeta2, dY, T2, R, T1, stat_in, mean2, invstd2, gamma2, W2,
                    mean1, invstd1):
        n = dY.shape[0]
        assert all(t.shape == (n, 16) and t.is_contiguous() for t in (G1, dY, T2, R, T1))
        nb = self.tok_blocks(n)
        assert stat_out.numel() >= nb * 32 and stat_in.numel() >= nb * 32 and wpart.numel() >= nb * 256
        _lib.check(self.lib.cg_tok_bwd_mid(_ptr(G1), _ptr(stat_out), _ptr(wpart), _ptr(dgamma2), _ptr(dbeta2), _ptr(dY),
                                           _ptr(T2), _ptr(R), _ptr(T1), _ptr(stat_in), _ptr(mean2), _ptr(invstd2),
                                           _ptr(gamma2), _ptr(W2), _ptr(mean1), _ptr(invstd1), n, self._stream()),
                   "cg_tok_bwd_mid")

    def tok_bwd_in(self, dX, wpart, dgamma1, dbeta1, G1, T1, X, dY, stat_in, mean1, invstd1, gamma1, W1):
        n = dY.shape[0]
        assert all(t.shape == (n, 16) and t.is_contiguous() for t in (dX, G1, T1, X, dY))
        nb = self.tok_blocks(n)
        assert stat_in.numel() >= nb * 32 and wpart.numel() >= nb * 256
        _lib.check(self.lib.cg_tok_bwd_in(_ptr(dX), _ptr(wpart), _ptr(dgamma1), _ptr(dbeta1), _ptr(G1), _ptr(T1),
                                          _ptr(X), _ptr(dY), _ptr(stat_in), _ptr(mean1), _ptr(invstd1), _ptr(gamma1),
                                          _ptr(W1), n, self._stream()), "cg_tok_bwd_in")

    def tok_wgrad_reduce(self, dW2, dW1, wpart2, wpart1, n_tok):
        assert dW2.numel() == 256 and dW1.numel() == 256 and dW2.is_contiguous() and dW1.is_contiguous()
        _lib.check(self.lib.cg_tok_wgrad_reduce(_ptr(dW2), _ptr(dW1), _ptr(wpart2), _ptr(wpart1), n_tok,
                                                self._stream()), "cg_tok_wgrad_reduce")

    # -- attention ------------------------------------------------------------------
    # X, Y, dX, dY: (cells * 16, E) token rows; E = latent_dim / 16
    def attn_fwd(self, Y, X, keep, params, heads):
        E = X.shape[-1]
        n_cells = X.numel() // (16 * E)
        assert X.is_contiguous() and Y.is_contiguous() and params.numel() == attn_nparam(E)
        assert keep is None or (keep.is_contiguous() and keep.numel() == n_cells * heads * 256)
        st = self.lib.cg_attn_ln_fwd(_ptr(Y), _ptr(X), _ptr(keep), _ptr(params), n_cells, E, heads, self._stream())
        _lib.check(st, "cg_attn_ln_fwd")

    def attn_block_bwd(self, dX, dparams, ws, dY, X, keep, params, heads, accumulate, T2, mean2, invstd2, stat_out,
                       reduce_now=True):
        """attn_bwd + tok_bn_bwd_stats of the ResidualBlock below: attn_bwd_blocks(n_cells) partials into stat_out.
        reduce_now=False: the parameter gradients stay as per-workgroup partials in ws (latent_param_reduce)."""
        E = X.shape[-1]
        n_cells = X.numel() // (16 * E)
        assert ws.numel() >= self.attn_bwd_ws(n_cells, E, heads) and dparams.numel() == attn_nparam(E)
        assert X.is_contiguous() and dY.is_contiguous() and dX.is_contiguous() and T2.is_contiguous()
        assert stat_out.numel() >= self.attn_bwd_blocks(n_cells) * 32
        st = self.lib.cg_attn_block_bwd(_ptr(dX), _ptr(dparams), _ptr(ws), _ptr(dY), _ptr(X), _ptr(keep), _ptr(params),
                                        n_cells, E, heads, int(accumulate), _ptr(T2), _ptr(mean2), _ptr(invstd2),
                                        _ptr(stat_out), int(reduce_now), self._stream())
        _lib.check(st, "cg_attn_block_bwd")

    def latent_param_reduce(self, attn_grads, attn_ws, tok_dW2, tok_dW1, wp2, wp1, n_cells):
        """The parameter-gradient reduces of all residual-attention blocks of a backward pass in two launches.
        attn_grads / tok_dW2 / tok_dW1: per block, the gradient tensors (views of one arena, at equal distances);
        attn_ws, wp2, wp1: (blocks, partials) workspaces the backward kernels filled.  Returns False (nothing
        launched) when the blocks' tensors are not at equal distances."""
        nb = len(attn_grads)

        def stride(ts):
            d = [(ts[k + 1].data_ptr() - ts[k].data_ptr()) for k in range(nb - 1)]
            return (d[0] // 4 if d else 0) if all(x == d[0] and x % 4 == 0 for x in d) else None
        sa, s2, s1 = stride(attn_grads), stride(tok_dW2), stride(tok_dW1)
        if sa is None or s2 is None or s1 is None or s2 != s1:
            return False
        assert attn_ws.shape[0] == nb and wp2.shape[0] == nb and wp1.shape[0] == nb and attn_ws.is_contiguous()
        st = self.lib.cg_attn_param_reduce_blocks(_ptr(attn_grads[0]), sa, _ptr(attn_ws), attn_ws.stride(0), nb, n_cells,
                                                  1, self._stream())
        _lib.check(st, "cg_attn_param_reduce_blocks")
        st = self.lib.cg_tok_wgrad_reduce_blocks(_ptr(tok_dW2[0]), _ptr(tok_dW1[0]), s2, _ptr(wp2), _ptr(wp1),
                                                 wp2.stride(0), nb, n_cells * 16, self._stream())
        _lib.check(st, "cg_tok_wgrad_reduce_blocks")
        return True

    def attn_bwd_blocks(self, n_cells):
        return self.lib.cg_attn_bwd_blocks(n_cells)

    def tok_bwd_mid_from(self, G1, stat_out, wpart, dgamma2, dbeta2, dY, T2, R, T1, stat_in, n_stat_in, mean2, invstd2,
                         gamma2, W2, mean1, invstd1):
        n = dY.shape[0]
        assert all(t.shape == (n, 16) and t.is_contiguous() for t in (G1, dY, T2, R, T1))
        nb = self.tok_blocks(n)
        assert stat_out.numel() >= nb * 32 and stat_in.numel() >= n_stat_in * 32 and wpart.numel() >= nb * 256
        _lib.check(self.lib.cg_tok_bwd_mid_from(_ptr(G1), _ptr(stat_out), _ptr(wpart), _ptr(dgamma2), _ptr(dbeta2),
                                                _ptr(dY), _ptr(T2), _ptr(R), _ptr(T1), _ptr(stat_in), n_stat_in,
                                                _ptr(mean2), _ptr(invstd2), _ptr(gamma2), _ptr(W2), _ptr(mean1),
                                                _ptr(invstd1), n, self._stream()), "cg_tok_bwd_mid_from")

    def attn_bwd_ws(self, n_cells, E=16, heads=4):
        n = self.lib.cg_attn_bwd_ws(n_cells, E, heads)
        if n < 0:
            raise _lib.CgError(f"cg_attn_bwd_ws: token width {E} with {heads} heads is not built "
                               "(widths 4, 8, 16, 32; heads dividing the width)")
        return n

    def attn_bwd(self, dX, dparams, ws, dY, X, keep, params, heads, accumulate):
        E = X.shape[-1]
        n_cells = X.numel() // (16 * E)
        assert ws.numel() >= self.attn_bwd_ws(n_cells, E, heads) and dparams.numel() == attn_nparam(E)
        assert X.is_contiguous() and dY.is_contiguous() and dX.is_contiguous()
        st = self.lib.cg_attn_ln_bwd(_ptr(dX), _ptr(dparams), _ptr(ws), _ptr(dY), _ptr(X), _ptr(keep), _ptr(params),
                                     n_cells, E, heads, int(accumulate), self._stream())
        _lib.check(st, "cg_attn_ln_bwd")
