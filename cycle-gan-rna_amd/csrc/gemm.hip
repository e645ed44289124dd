// MFMA GEMM for gfx950: C[M,N] (+)= op(A)[M,K] * op(B)[K,N], float32 accumulate.
//
// float32 path: v_mfma_f32_32x32x2_f32 (exact float32 products, k-ordered fmaf
// chain).  256 threads = 4 waves laid out 2x2 over a BM x BN block tile, each
// wave owning (BM/2) x (BN/2) as 32x32 MFMA tiles.  Tiles are staged
// global -> registers -> LDS, double-buffered over K in slabs of BK = 32.
//
// An operand whose reduction index is contiguous in memory ("k-major": A with
// transA=0, B with transB=1) is kept in LDS as [row][BK+4] and read with one
// ds_read_b128 per four MFMA k-steps; an operand whose reduction index is the
// slow one ("m-major") is kept as [k][rows+4] and read with ds_read_b32.  The
// 32x32x2 MFMA consumes two k values per step (lane half h = lane>>5 supplies
// one of them); both operands use k = 8c + 4h + s for step s of group c, so the
// sum covers every k exactly once whatever the storage.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <hip/hip_ext.h>
#include "common.h"
#include "gemm_tuned.h"

// ------------------------------------------------------------------------------------------------
// Measurement aid (cg_gemm_timing_begin / _end): while a recording is open on the calling thread, the first kernel
// of every cg_gemm* call carries a start event and its last kernel (the split-K slab reduce, when there is one) a
// stop event, attached to the dispatches themselves (hipExtLaunchKernelGGL), so that the elapsed time of a pair is
// the device time from the first kernel's start to the last kernel's end.
// ------------------------------------------------------------------------------------------------
namespace {
struct GemmTiming {
  std::vector<hipEvent_t> ev;     // start/stop pairs, one per call
  std::vector<char> state;        // per call: bit 0 start attached, bit 1 stop attached
  size_t calls = 0;
  bool open = false;
  hipEvent_t start = nullptr, stop = nullptr;     // of the call in progress, until a launch takes them
};
thread_local GemmTiming tl_timing;

inline void timing_begin_call() {
  GemmTiming& t = tl_timing;
  t.start = t.stop = nullptr;
  if (!t.open || 2 * (t.calls + 1) > t.ev.size()) return;
  t.start = t.ev[2 * t.calls];
  t.stop = t.ev[2 * t.calls + 1];
  t.calls += 1;
}
inline hipEvent_t timing_take_start() {
  GemmTiming& t = tl_timing;
  hipEvent_t e = t.start;
  if (e) { t.start = nullptr; t.state[t.calls - 1] |= 1; }
  return e;
}
inline hipEvent_t timing_take_stop(bool last) {
  GemmTiming& t = tl_timing;
  if (!last) return nullptr;
  hipEvent_t e = t.stop;
  if (e) { t.stop = nullptr; t.state[t.calls - 1] |= 2; }
  return e;
}
}  // namespace
// A kernel launch of cg_gemm*: `last` says whether the call launches nothing after it.
// (the plain launch whenever no recording is open, so that captured steps hold ordinary kernel nodes)
#define CG_GEMM_LAUNCH(last, kernel, grid, block, st, ...)                                                            \
  do {                                                                                                               \
    if (tl_timing.open)                                                                                              \
      hipExtLaunchKernelGGL(kernel, grid, block, 0, st, timing_take_start(), timing_take_stop(last), 0, __VA_ARGS__); \
    else                                                                                                             \
      hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                                   \
  } while (0)

#ifndef CG_PF_SMALL
#define CG_PF_SMALL 2
#define CG_PF_MID 2
#define CG_PF_BIG 2
#endif
#ifndef CG_PFH_SMALL
#define CG_PFH_SMALL 2
#define CG_PFH_MID 2
#define CG_PFH_BIG 1
#endif
#ifndef CG_KWH_SMALL
#define CG_KWH_SMALL 2
#define CG_KWH_MID 2
#define CG_KWH_BIG 1
#endif
#ifndef CG_KW_SMALL
#define CG_KW_SMALL 2
#define CG_KW_MID 2
#define CG_KW_BIG 1
#endif

namespace {

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* E;
  float* aux;
  int M, N, K, lda, ldb, ldc, lde, ldaux;
  int epi, accumulate;
  float slope;
  int tiles_n;
  int kt_per_split;   // K slabs handled by one split (split-K; see xcd_tile_split); all of them when gridDim.y == 1
  float* ws;          // split-K: partial results [gridDim.y][M][N]
  unsigned a_bytes, b_bytes;   // footprints of A and B in memory (hardware range check of the buffer loads)
  u16* C16;           // optional bf16 image of C (same leading dimension): every value stored to C is also stored here
  unsigned* counters; // split-K combined inside the kernel (see split_tail): one arrival counter per output tile, else nullptr
  int vec4;           // the combining pass may use 16-byte accesses (see reduce_vec4)
};

#ifndef CG_BK
#define CG_BK 32
#endif
constexpr int BK = CG_BK;

// Slabs of global loads kept in flight ahead of the MFMAs, per block tile (register stages).
constexpr int prefetch_depth(int bm, int bn) { return bm * bn <= 64 * 64 ? CG_PF_SMALL : (bm * bn <= 128 * 64 ? CG_PF_MID : CG_PF_BIG); }
// Wave groups per workgroup of the float32 kernel (see gemm_f32_kernel).
constexpr int kwaves(int bm, int bn) { return bm * bn <= 64 * 64 ? CG_KW_SMALL : (bm * bn <= 128 * 64 ? CG_KW_MID : CG_KW_BIG); }
constexpr int prefetch_depth_bf16(int bm, int bn) { return bm * bn <= 64 * 64 ? CG_PFH_SMALL : (bm * bn <= 128 * 64 ? CG_PFH_MID : CG_PFH_BIG); }
constexpr int kwaves_bf16(int bm, int bn) { return bm * bn <= 64 * 64 ? CG_KWH_SMALL : (bm * bn <= 128 * 64 ? CG_KWH_MID : CG_KWH_BIG); }

// Operand tiles are read with buffer loads: the hardware range check returns 0 for an offset at or beyond the
// operand's footprint, so rows and k values outside the matrix are requested at OOB_OFF and need no branch; what a
// 16-byte load picks up past the end of a row (the next row's elements) is cleared when the tile goes to LDS.
// Branch-free loads let the compiler count them (s_waitcnt vmcnt(n)) and keep several slabs in flight.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB_OFF = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Four elements along the contiguous dimension starting at byte offset `off` (element index c0 of climit).
template <bool VEC>
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t rs, unsigned off, bool ok, int c0, int climit) {
  if (VEC) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB_OFF, 0, 0));
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (ok && c0 + e < climit) ? off + 4 * e : OOB_OFF, 0, 0));
  return v;
}

template <bool VEC>
__device__ __forceinline__ f32x4 clip4(f32x4 v, int c0, int climit) {
  if (VEC) {
#pragma unroll
    for (int e = 1; e < 4; ++e) v[e] = c0 + e < climit ? v[e] : 0.f;
  }
  return v;
}

// Stage registers for one operand tile of ROWS x BK.
//  k-major: element (r,k) = ptr[(row0+r)*ld + k0+k]
//  m-major: element (r,k) = ptr[(k0+k)*ld + row0+r]
template <int ROWS, bool MMAJ, bool VEC, int NT>
struct Stager {
  static constexpr int NV = ROWS * BK / 4 / NT;  // float4 per thread
  static_assert(NV >= 1 && NV * NT * 4 == ROWS * BK, "tile does not divide over the workgroup");
  f32x4 v[NV];
  int k0s;   // first k of the slab held in v

  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0, int nrows, int k0, int K) {
    const int t = threadIdx.x;
    k0s = k0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + NT * j;
      if (!MMAJ) {
        const int r = idx / (BK / 4), kq = idx % (BK / 4);
        const int row = row0 + r, k = k0 + 4 * kq;
        v[j] = bload4<VEC>(rs, ((unsigned)row * ld + k) * 4u, row < nrows && k < K, k, K);
      } else {
        const int k = idx / (ROWS / 4), rq = idx % (ROWS / 4);
        const int row = row0 + 4 * rq, kk = k0 + k;
        v[j] = bload4<VEC>(rs, ((unsigned)kk * ld + row) * 4u, kk < K && row < nrows, row, nrows);
      }
    }
  }
  __device__ __forceinline__ void store(float* s, int row0, int nrows, int K) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + NT * j;
      if (!MMAJ) {
        const int r = idx / (BK / 4), kq = idx % (BK / 4);
        *reinterpret_cast<f32x4*>(s + r * (BK + 4) + 4 * kq) = clip4<VEC>(v[j], k0s + 4 * kq, K);
      } else {
        const int k = idx / (ROWS / 4), rq = idx % (ROWS / 4);
        *reinterpret_cast<f32x4*>(s + k * (ROWS + 4) + 4 * rq) = clip4<VEC>(v[j], row0 + 4 * rq, nrows);
      }
    }
  }
};

// One output element through the epilogue (bias already added by the caller where it applies).
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int row, int col, float h) {
  float* cp = g.C + (size_t)row * g.ldc + col;
  float out;
  if (g.epi == CG_EPI_NONE) {
    out = g.accumulate ? h + *cp : h;
  } else if (g.epi == CG_EPI_LEAKY) {
    float pm = h > 0.f ? 1.f : g.slope;
    if (g.E) pm *= g.E[(size_t)row * g.lde + col];
    out = h * pm;
    if (g.aux) g.aux[(size_t)row * g.ldaux + col] = pm;
  } else {
    out = h * g.E[(size_t)row * g.lde + col];
  }
  *cp = out;
  if (g.C16) g.C16[(size_t)row * g.ldc + col] = f2bf(out);
}

// Workgroup -> (output tile, K split).  Workgroups are handed to the eight XCDs round-robin in dispatch order
// (x fastest, then y).  Each XCD gets a contiguous run of the (split, tile) pairs, split-major: with S splits an XCD
// works on 1/8 of the pairs, i.e. on one split's K range (or a few) for a compact patch of tiles, so what its L2
// requests from the fabric is that K range of the operand panels only - not the whole K extent of every panel its
// tiles touch, as when all splits of a tile share an XCD.  Which split a partial sum belongs to (its slab) does not
// depend on where it was computed, so results are unchanged.
__device__ __forceinline__ void xcd_tile_split(int& tile, int& split) {
  const int tiles = gridDim.x, nwg = tiles * gridDim.y;
  const int lin = blockIdx.x + tiles * blockIdx.y;
  const int q = nwg >> 3, rem = nwg & 7, xcd = lin & 7, idx = lin >> 3;
  const int p = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
  split = p / tiles;
  tile = p - split * tiles;
}

// Tile index -> (tile row, tile column).  Within an XCD's contiguous run the tiles are walked column-first inside
// bands of GROUP_M tile rows, so the workgroups that are resident together (32 CUs x up to 4) cover a compact
// patch of the output: the A row panels and B column panels they share stay within that XCD's 4 MB L2 instead of
// sweeping a whole row of tiles (all of B) at a time.
constexpr int GROUP_M = 8;
__device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int band = tile / (GROUP_M * tiles_n);
  const int first = band * GROUP_M;
  const int rows = min(GROUP_M, tiles_m - first);
  const int r = tile - band * GROUP_M * tiles_n;
  tm = first + r % rows;
  tn = r / rows;
}

// ------------------------------------------------------------------------------------------------
// bf16 path: float32 operands in memory, rounded to bf16 while they are staged into LDS, multiplied
// with v_mfma_f32_16x16x32_bf16 (float32 accumulate).  BKH = 64 per slab, double-buffered; the LDS
// images of the two operand kinds are described at lds_elems() below.
// ------------------------------------------------------------------------------------------------
constexpr int BKH = 64;
constexpr int LDH = BKH + 16;   // LDS row stride in bf16 elements

__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  uint2 r;
  r.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  r.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  return r;
}

// LDS images: a k-major operand is [row][k] with LDH elements per row (one ds_read_b128 = the 8 consecutive k of a
// lane's MFMA fragment).  An m-major operand stays as it lies in memory, [k][row] with ROWS + 16 elements per k row -
// its 16-byte loads run along the rows and go to LDS untransposed, lane after lane contiguous - and the fragment is
// fetched with two transposing reads (ds_read_b64_tr_b16: 4 k rows x 16 rows per 16-lane group, delivered
// row-major).  Transposing in registers on the way in instead made 32 lanes of a store hit one bank.
template <int ROWS>
constexpr int mmaj_stride() { return ROWS + 16; }
template <int ROWS, bool MMAJ>
constexpr int lds_elems() { return MMAJ ? BKH * mmaj_stride<ROWS>() : ROWS * LDH; }

template <int ROWS, bool MMAJ, bool VEC, int NT>
struct StagerH {
  static constexpr int NV = ROWS * BKH / 4 / NT;   // float4 per thread
  static_assert(NV >= 1 && NV * NT * 4 == ROWS * BKH, "tile does not divide over the workgroup");
  f32x4 v[NV];
  int k0s;

  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0, int nrows, int k0, int K) {
    const int t = threadIdx.x;
    k0s = k0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + NT * j;
      if (!MMAJ) {
        const int r = idx / (BKH / 4), kq = idx % (BKH / 4);
        const int row = row0 + r, k = k0 + 4 * kq;
        v[j] = bload4<VEC>(rs, ((unsigned)row * ld + k) * 4u, row < nrows && k < K, k, K);
      } else {
        const int k = idx / (ROWS / 4), rq = idx % (ROWS / 4);
        const int kk = k0 + k, row = row0 + 4 * rq;
        v[j] = bload4<VEC>(rs, ((unsigned)kk * ld + row) * 4u, kk < K && row < nrows, row, nrows);
      }
    }
  }
  __device__ __forceinline__ void store(u16* s, int row0, int nrows, int K) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + NT * j;
      if (!MMAJ) {
        const int r = idx / (BKH / 4), kq = idx % (BKH / 4);
        const f32x4 w = clip4<VEC>(v[j], k0s + 4 * kq, K);
        *reinterpret_cast<uint2*>(s + r * LDH + 4 * kq) = pack4(w[0], w[1], w[2], w[3]);
      } else {
        const int k = idx / (ROWS / 4), rq = idx % (ROWS / 4);
        const f32x4 w = clip4<VEC>(v[j], row0 + 4 * rq, nrows);
        *reinterpret_cast<uint2*>(s + k * mmaj_stride<ROWS>() + 4 * rq) = pack4(w[0], w[1], w[2], w[3]);
      }
    }
  }
};

// The same LDS images for an operand that is ALREADY bf16 in memory (a shadow copy kept by its producer: the Adam
// update writes one of every weight): 16-byte loads of eight elements go to LDS as they are - half the bytes, no
// conversion.  Loads need 8-byte alignment only (base address a multiple of 8 bytes, leading dimension a multiple of
// 4 elements, so that the row of a column slice of a 16-byte aligned matrix qualifies) and the contiguous extent must
// be even: what a load picks up past the end of a row is cleared in pairs.
template <int ROWS, bool MMAJ, int NT>
struct StagerB16 {
  static constexpr int NV = ROWS * BKH / 8 / NT;   // 16-byte loads per thread
  static_assert(NV >= 1 && NV * NT * 8 == ROWS * BKH, "tile does not divide over the workgroup");
  u32x4 v[NV];
  int k0s;

  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int ld, int row0, int nrows, int k0, int K) {
    const int t = threadIdx.x;
    k0s = k0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + NT * j;
      if (!MMAJ) {
        const int r = idx / (BKH / 8), kq = idx % (BKH / 8);
        const int row = row0 + r, k = k0 + 8 * kq;
        v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (row < nrows && k < K) ? ((unsigned)row * ld + k) * 2u : OOB_OFF, 0, 0);
      } else {
        const int k = idx / (ROWS / 8), rq = idx % (ROWS / 8);
        const int kk = k0 + k, row = row0 + 8 * rq;
        v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (kk < K && row < nrows) ? ((unsigned)kk * ld + row) * 2u : OOB_OFF, 0, 0);
      }
    }
  }
  __device__ __forceinline__ static u32x4 clip8(u32x4 w, int c0, int climit) {
#pragma unroll
    for (int e = 1; e < 4; ++e) w[e] = c0 + 2 * e < climit ? w[e] : 0u;
    return w;
  }
  __device__ __forceinline__ void store(u16* s, int row0, int nrows, int K) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + NT * j;
      if (!MMAJ) {
        const int r = idx / (BKH / 8), kq = idx % (BKH / 8);
        *reinterpret_cast<u32x4*>(s + r * LDH + 8 * kq) = clip8(v[j], k0s + 8 * kq, K);
      } else {
        const int k = idx / (ROWS / 8), rq = idx % (ROWS / 8);
        *reinterpret_cast<u32x4*>(s + k * mmaj_stride<ROWS>() + 8 * rq) = clip8(v[j], row0 + 8 * rq, nrows);
      }
    }
  }
};

template <int ROWS, bool MMAJ, bool VEC, int NT, bool SRC16>
struct StagerSel { typedef StagerH<ROWS, MMAJ, VEC, NT> type; };
template <int ROWS, bool MMAJ, bool VEC, int NT>
struct StagerSel<ROWS, MMAJ, VEC, NT, true> { typedef StagerB16<ROWS, MMAJ, NT> type; };

typedef short s16x4 __attribute__((ext_vector_type(4)));

// The bf16x8 MFMA fragment of lane (lr = lane & 15, lg = lane >> 4): row `row0 + lr`, k = kbase + 8*lg + 0..7.
template <int ROWS, bool MMAJ>
__device__ __forceinline__ bf16x8 fragment(const u16* s, int row0, int kbase, int lr, int lg) {
  if (!MMAJ) return *reinterpret_cast<const bf16x8*>(s + (row0 + lr) * LDH + kbase + 8 * lg);
  // lane 4q+p of a 16-lane group names k row q, rows 4p..4p+3 of the group's block; it receives row lr, k rows 0..3
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  const u16* p0 = s + (kbase + 8 * lg + (lr >> 2)) * mmaj_stride<ROWS>() + row0 + 4 * (lr & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * mmaj_stride<ROWS>()));
  union { struct { s16x4 a, b; } h; bf16x8 f; } u;
  u.h.a = lo;
  u.h.b = hi;
  return u.f;
}

// One output element (VEC4: four consecutive columns, 16-byte accesses) from the split-K partial sums: the bias, then
// the slabs in split order, then the epilogue of cg_gemm.
// own (with own_split >= 0): this workgroup's own slab, still in LDS, taken from there instead of from memory.
template <bool VEC4, bool SC1 = false>
__device__ __forceinline__ void combine_slabs(const GemmArgs& g, int splits, int m, int n, int own_split = -1,
                                              const float* own = nullptr) {
  constexpr int W = VEC4 ? 4 : 1;
  const size_t e0 = (size_t)m * g.N + n;
  float h[W];
#pragma unroll
  for (int e = 0; e < W; ++e) h[e] = g.bias ? g.bias[n + e] : 0.f;
  for (int k = 0; k < splits; ++k) {
    const float* p = g.ws + (size_t)k * g.M * g.N + e0;
    if (VEC4) {
      // SC1: slabs stored write-through by other workgroups of this launch are read past this CU's L1
      const f32x4 v = k == own_split ? *reinterpret_cast<const f32x4*>(own) : SC1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                make_rsrc(g.ws, 0xfffffff0u), (unsigned)(((size_t)k * g.M * g.N + e0) * 4), 0, 16))
                          : *reinterpret_cast<const f32x4*>(p);
#pragma unroll
      for (int e = 0; e < W; ++e) h[e] += v[e];
    } else {
      h[0] += SC1 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[0];
    }
  }
  float* cp = g.C + (size_t)m * g.ldc + n;
  float out[W], pmv[W];
  if (g.epi == CG_EPI_NONE) {
    if (g.accumulate) {
      if (VEC4) {
        const f32x4 c = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
        for (int e = 0; e < W; ++e) out[e] = c[e] + h[e];
      } else {
        out[0] = cp[0] + h[0];
      }
    } else {
#pragma unroll
      for (int e = 0; e < W; ++e) out[e] = h[e];
    }
  } else {
    float ev[W];
    if (g.E) {
      const float* ep = g.E + (size_t)m * g.lde + n;
      if (VEC4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(ep);
#pragma unroll
        for (int e = 0; e < W; ++e) ev[e] = t[e];
      } else {
        ev[0] = ep[0];
      }
    } else {
#pragma unroll
      for (int e = 0; e < W; ++e) ev[e] = 1.f;
    }
#pragma unroll
    for (int e = 0; e < W; ++e) {
      pmv[e] = g.epi == CG_EPI_LEAKY ? (h[e] > 0.f ? 1.f : g.slope) * ev[e] : ev[e];
      out[e] = h[e] * pmv[e];
    }
    if (g.epi == CG_EPI_LEAKY && g.aux) {
      float* ap = g.aux + (size_t)m * g.ldaux + n;
      if (VEC4) *reinterpret_cast<f32x4*>(ap) = f32x4{pmv[0], pmv[W > 1 ? 1 : 0], pmv[W > 2 ? 2 : 0], pmv[W > 3 ? 3 : 0]};
      else ap[0] = pmv[0];
    }
  }
  if (VEC4) *reinterpret_cast<f32x4*>(cp) = f32x4{out[0], out[W > 1 ? 1 : 0], out[W > 2 ? 2 : 0], out[W > 3 ? 3 : 0]};
  else cp[0] = out[0];
  if (g.C16) {
    u16* hp = g.C16 + (size_t)m * g.ldc + n;
    if (VEC4) *reinterpret_cast<uint2*>(hp) = pack4(out[0], out[W > 1 ? 1 : 0], out[W > 2 ? 2 : 0], out[W > 3 ? 3 : 0]);
    else hp[0] = f2bf(out[0]);
  }
}

// Split-K without a second launch (GemmArgs::counters): every workgroup, having written its slab, makes its stores
// visible to the whole device (release fence) and one of its threads adds 1 to the arrival counter of the output
// tile.  The workgroup that finds gridDim.y - 1 there is the last of its tile: behind an acquire fence it combines
// the tile's slabs in split order - the additions of splitk_reduce_kernel in the same order, whichever workgroup
// happens to arrive last - and puts the counter back to 0 for the next product that uses it.  No workgroup ever
// waits for another.  Every thread of the workgroup must call this (two barriers); `flag`: one int of LDS that
// nothing else uses any more.
// own_tile (may be nullptr): the workgroup's partial sums as [BM][BN + 4] floats in LDS, slab own_split.
template <int BM, int BN, int NT>
__device__ __forceinline__ void split_tail(const GemmArgs& g, int tile, int bm, int bn, int* flag, int own_split = -1,
                                           const float* own_tile = nullptr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through slab stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned arrived = __hip_atomic_fetch_add(g.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = arrived == gridDim.y - 1;
    if (last) __hip_atomic_exchange(g.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  const int splits = gridDim.y;
  if (g.vec4) {
    for (int i = threadIdx.x; i < BM * BN / 4; i += NT) {
      const int m = bm + i / (BN / 4), n = bn + 4 * (i % (BN / 4));
      if (m < g.M && n < g.N)
        combine_slabs<true, true>(g, splits, m, n, own_tile ? own_split : -1, own_tile + (i / (BN / 4)) * (BN + 4) + 4 * (i % (BN / 4)));
    }
  } else {
    for (int i = threadIdx.x; i < BM * BN; i += NT) {
      const int m = bm + i / BN, n = bn + i % BN;
      if (m < g.M && n < g.N) combine_slabs<false, true>(g, splits, m, n);
    }
  }
}

// AH / BH: the operand is stored as bf16 (see StagerB16), otherwise as float32 and rounded while staged.
template <int BM, int BN, bool TA, bool TB, bool VEC, int KW, bool AH, bool BH>
__global__ void __launch_bounds__(256 * KW) gemm_bf16_kernel(GemmArgs g) {
  constexpr int NT = 256 * KW;                       // KW wave groups, each taking 1/KW of every slab (see gemm_f32_kernel)
  constexpr int TM = BM / 32, TN = BN / 32;          // 16x16 MFMA tiles per wave in each direction
  constexpr bool A_MMAJ = TA, B_MMAJ = !TB;
  constexpr int A_ELEMS = lds_elems<BM, A_MMAJ>(), B_ELEMS = lds_elems<BN, B_MMAJ>();
  __shared__ __align__(16) u16 smem[2 * (A_ELEMS + B_ELEMS)];

  int tile, split;
  xcd_tile_split(tile, split);
  int tile_m, tile_n;
  tile_coords(tile, (g.M + BM - 1) / BM, g.tiles_n, tile_m, tile_n);
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, kg = threadIdx.x >> 8;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int lr = lane & 15, lg = lane >> 4;
  constexpr int KSG = BKH / 32 / KW;                 // 32-deep MFMA steps per slab and wave group
  static_assert(KSG * KW * 32 == BKH, "K slab does not divide over the wave groups");
  static_assert(KW == 1 || TM * TN * 4 * 256 * (KW - 1) * 2 <= 2 * (A_ELEMS + B_ELEMS), "accumulator exchange does not fit in LDS");

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.A, g.a_bytes), rsB = make_rsrc(g.B, g.b_bytes);
  constexpr int D = prefetch_depth_bf16(BM, BN);   // register stages of global loads in flight (see gemm_f32_kernel)
  typename StagerSel<BM, A_MMAJ, VEC, NT, AH>::type sa[D];
  typename StagerSel<BN, B_MMAJ, VEC, NT, BH>::type sb[D];
  const int nk_all = (g.K + BKH - 1) / BKH;
  const int kt0 = split * g.kt_per_split;
  const int nk = min(nk_all, kt0 + g.kt_per_split);

  const int klim = min(g.K, nk * BKH);   // straight-line loop body, see gemm_f32_kernel
#pragma unroll
  for (int s = 0; s < D; ++s) {
    sa[s].load(rsA, g.lda, bm, g.M, (kt0 + s) * BKH, klim);
    sb[s].load(rsB, g.ldb, bn, g.N, (kt0 + s) * BKH, klim);
  }
  sa[0].store(smem, bm, g.M, klim);
  sb[0].store(smem + A_ELEMS, bn, g.N, klim);
  __syncthreads();

  int cur = 0;
  for (int base = kt0; base < nk; base += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int kt = base + s;
      sa[s].load(rsA, g.lda, bm, g.M, (kt + D) * BKH, klim);
      sb[s].load(rsB, g.ldb, bn, g.N, (kt + D) * BKH, klim);
      const u16* As = smem + cur * (A_ELEMS + B_ELEMS);
      const u16* Bs = As + A_ELEMS;
#pragma unroll
      for (int kk = 0; kk < KSG; ++kk) {
        const int ks = kg * KSG + kk;
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = fragment<BM, A_MMAJ>(As, wm + 16 * i, 32 * ks, lr, lg);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = fragment<BN, B_MMAJ>(Bs, wn + 16 * j, 32 * ks, lr, lg);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      u16* nxt = smem + (cur ^ 1) * (A_ELEMS + B_ELEMS);
      sa[(s + 1) % D].store(nxt, bm, g.M, klim);
      sb[(s + 1) % D].store(nxt + A_ELEMS, bn, g.N, klim);
      __syncthreads();
      cur ^= 1;
    }
  }

  const bool fused = gridDim.y > 1 && g.counters != nullptr;   // split-K combined by the last workgroup of each tile
  if (KW > 1) {   // add the wave groups' accumulators through LDS (the K loop ended on a barrier)
    float* xs = reinterpret_cast<float*>(smem) + ((kg > 0 ? kg - 1 : 0) * 4 + wave) * (TM * TN * 4 * 64) + lane;
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) xs[((i * TN + j) * 4 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (kg > 0 && !fused) return;
    if (kg == 0) {
#pragma unroll
      for (int q = 1; q < KW; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc[i][j][r] += xs[(q - 1) * 4 * (TM * TN * 4 * 64) + ((i * TN + j) * 4 + r) * 64];
    }
  }

  // 16x16 accumulator: col = lane & 15, row = 4 * (lane >> 4) + reg
  if (fused && g.vec4) {
    // as in finish_f32: the tile goes through LDS and leaves for the slab as write-through 16-byte pieces of rows
    constexpr int LDT = BN + 4;
    static_assert(BM * LDT * 4 <= 2 * (A_ELEMS + B_ELEMS) * 2, "the output tile does not fit in the staging LDS");
    float* tl = reinterpret_cast<float*>(smem);
    __syncthreads();      // the last slab (or the accumulator exchange) has been read
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) tl[(wm + 16 * i + 4 * lg + r) * LDT + wn + 16 * j + lr] = acc[i][j][r];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(g.ws, 0xfffffff0u);
    const size_t slab = (size_t)split * g.M * g.N;
    for (int idx = threadIdx.x; idx < BM * BN / 4; idx += NT) {
      const int r = idx / (BN / 4), c = 4 * (idx % (BN / 4)), row = bm + r, col = bn + c;
      if (row < g.M && col < g.N)
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(tl + r * LDT + c), rs,
                                               (unsigned)((slab + (size_t)row * g.N + col) * 4), 0, 16);   // 16: sc1
    }
    split_tail<BM, BN, NT>(g, tile, bm, bn, reinterpret_cast<int*>(tl + BN), split, tl);   // the flag in the padding of the tile's first row
    return;
  }
  float* wsp = gridDim.y > 1 ? g.ws + (size_t)split * g.M * g.N : nullptr;
  if (kg == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = bn + wn + 16 * j + lr;
        if (col >= g.N) continue;
        const float bv = (g.bias && !wsp) ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = bm + wm + 16 * i + 4 * lg + r;
          if (row >= g.M) continue;
          if (wsp && fused) __hip_atomic_store(wsp + (size_t)row * g.N + col, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else if (wsp) wsp[(size_t)row * g.N + col] = acc[i][j][r];
          else epilogue_store(g, row, col, acc[i][j][r] + bv);
        }
      }
  }
  if (fused) split_tail<BM, BN, NT>(g, tile, bm, bn, reinterpret_cast<int*>(smem));
}

// Tail of the float32 kernels: the wave groups' accumulators are added through LDS in group order (the caller's K
// loop has ended on a barrier and nothing else touches LDS any more), then either the raw partial sums go to this
// split's slab or the epilogue is applied.  32x32 accumulator: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
template <int TM, int TN, int KW>
__device__ __forceinline__ void finish_f32(const GemmArgs& g, f32x16 (&acc)[TM][TN], float* smem, int bm, int bn, int wm,
                                           int wn, int lane, int wave, int kg, int split, int tile) {
  const int li = lane & 31, lh = lane >> 5;
  const bool fused = gridDim.y > 1 && g.counters != nullptr;   // split-K combined by the last workgroup of each tile
  if (KW > 1) {
    float* xs = smem + ((kg > 0 ? kg - 1 : 0) * 4 + wave) * (TM * TN * 16 * 64) + lane;
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) xs[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (kg > 0 && !fused) return;
    if (kg == 0) {
#pragma unroll
      for (int q = 1; q < KW; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[i][j][r] += xs[(q - 1) * 4 * (TM * TN * 16 * 64) + ((i * TN + j) * 16 + r) * 64];
    }
  }
  if (fused && g.vec4) {
    // slab stores that other workgroups read in this launch are write-through, and a write-through store of 4 bytes
    // is a fabric write of its own: the tile goes through LDS and leaves as 16-byte pieces of rows
    constexpr int BM = 64 * TM, BN = 64 * TN, LDT = BN + 4, NT = 256 * KW;
    __syncthreads();      // the accumulator exchange has been read
    if (kg == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            smem[(wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDT + wn + 32 * j + li] = acc[i][j][r];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(g.ws, 0xfffffff0u);
    const size_t slab = (size_t)split * g.M * g.N;
    for (int idx = threadIdx.x; idx < BM * BN / 4; idx += NT) {
      const int r = idx / (BN / 4), c = 4 * (idx % (BN / 4)), row = bm + r, col = bn + c;
      if (row < g.M && col < g.N)
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(smem + r * LDT + c), rs,
                                               (unsigned)((slab + (size_t)row * g.N + col) * 4), 0, 16);   // 16: sc1
    }
    split_tail<BM, BN, NT>(g, tile, bm, bn, reinterpret_cast<int*>(smem + BN), split, smem);   // the flag in the padding of the tile's first row
    return;
  }
  if (gridDim.y > 1) {   // split-K: raw partial sums, combined by split_tail, splitk_reduce_kernel or the consumer
    if (kg == 0) {
      float* wsp = g.ws + (size_t)split * g.M * g.N;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = bn + wn + 32 * j + li;
          if (col >= g.N) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = bm + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < g.M) {
              if (fused) __hip_atomic_store(wsp + (size_t)row * g.N + col, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else wsp[(size_t)row * g.N + col] = acc[i][j][r];
            }
          }
        }
    }
    if (fused) split_tail<64 * TM, 64 * TN, 256 * KW>(g, tile, bm, bn, reinterpret_cast<int*>(smem));
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = bn + wn + 32 * j + li;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = bm + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= g.M) continue;
        epilogue_store(g, row, col, acc[i][j][r] + bv);
      }
    }
}

// KW wave groups of four waves share a block tile: every group multiplies its own 1/KW of each K slab into its own
// accumulators (two or more waves per SIMD hide each other's LDS and barrier waits even when the workgroup has
// the CU to itself); the groups' accumulators are added through LDS, in group order, before the epilogue.
template <int BM, int BN, bool TA, bool TB, bool VEC, int KW>
__global__ void __launch_bounds__(256 * KW) gemm_f32_kernel(GemmArgs g) {
  constexpr int NT = 256 * KW;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr bool A_MMAJ = TA, B_MMAJ = !TB;
  constexpr int A_ELEMS = A_MMAJ ? BK * (BM + 4) : BM * (BK + 4);
  constexpr int B_ELEMS = B_MMAJ ? BK * (BN + 4) : BN * (BK + 4);
  __shared__ __align__(16) float smem[2 * (A_ELEMS + B_ELEMS)];

  int tile, split;
  xcd_tile_split(tile, split);
  int tile_m, tile_n;
  tile_coords(tile, (g.M + BM - 1) / BM, g.tiles_n, tile_m, tile_n);
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, kg = threadIdx.x >> 8;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int li = lane & 31, lh = lane >> 5;
  constexpr int CG = BK / 8 / KW;   // groups of eight k per slab handled by one wave group
  static_assert(CG * KW * 8 == BK, "K slab does not divide over the wave groups");
  static_assert(KW == 1 || TM * TN * 16 * 256 * (KW - 1) <= 2 * (A_ELEMS + B_ELEMS), "accumulator exchange does not fit in LDS");
  static_assert(BM * (BN + 4) <= 2 * (A_ELEMS + B_ELEMS), "the output tile (finish_f32, split-K tail) does not fit in LDS");

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Global loads run D slabs ahead of the MFMAs (D register stages; LDS stays double-buffered): a workgroup that
  // has a CU to itself - most launches of this workload have at most one workgroup per CU - must cover the
  // global-load latency with its own work, and one slab of MFMAs is shorter than that latency.
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.A, g.a_bytes), rsB = make_rsrc(g.B, g.b_bytes);
  constexpr int D = prefetch_depth(BM, BN);
  Stager<BM, A_MMAJ, VEC, NT> sa[D];
  Stager<BN, B_MMAJ, VEC, NT> sb[D];
  const int nk_all = (g.K + BK - 1) / BK;
  const int kt0 = split * g.kt_per_split;
  const int nk = min(nk_all, kt0 + g.kt_per_split);

  // The loop body is straight-line: every load, LDS store and barrier is unconditional, so the compiler can count
  // the loads in flight (a conditional load or store makes it drain them all).  The trip count is rounded up to a
  // multiple of D; slabs at or beyond klim read as zeros (the range-checked loads) and add nothing.
  const int klim = min(g.K, nk * BK);
#pragma unroll
  for (int s = 0; s < D; ++s) {
    sa[s].load(rsA, g.lda, bm, g.M, (kt0 + s) * BK, klim);
    sb[s].load(rsB, g.ldb, bn, g.N, (kt0 + s) * BK, klim);
  }
  sa[0].store(smem, bm, g.M, klim);
  sb[0].store(smem + A_ELEMS, bn, g.N, klim);
  __syncthreads();

  int cur = 0;
  for (int base = kt0; base < nk; base += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int kt = base + s;
      // stage s went to LDS one step ago: refill it with the slab D ahead
      sa[s].load(rsA, g.lda, bm, g.M, (kt + D) * BK, klim);
      sb[s].load(rsB, g.ldb, bn, g.N, (kt + D) * BK, klim);
      const float* As = smem + cur * (A_ELEMS + B_ELEMS);
      const float* Bs = As + A_ELEMS;
#pragma unroll
      for (int cc = 0; cc < CG; ++cc) {
        const int c = kg * CG + cc;
        float a[TM][4], b[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (!A_MMAJ) {
            f32x4 t = *reinterpret_cast<const f32x4*>(As + (wm + 32 * i + li) * (BK + 4) + 8 * c + 4 * lh);
            a[i][0] = t[0]; a[i][1] = t[1]; a[i][2] = t[2]; a[i][3] = t[3];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[i][q] = As[(8 * c + 4 * lh + q) * (BM + 4) + wm + 32 * i + li];
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (!B_MMAJ) {
            f32x4 t = *reinterpret_cast<const f32x4*>(Bs + (wn + 32 * j + li) * (BK + 4) + 8 * c + 4 * lh);
            b[j][0] = t[0]; b[j][1] = t[1]; b[j][2] = t[2]; b[j][3] = t[3];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) b[j][q] = Bs[(8 * c + 4 * lh + q) * (BN + 4) + wn + 32 * j + li];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
      }
      float* nxt = smem + (cur ^ 1) * (A_ELEMS + B_ELEMS);
      sa[(s + 1) % D].store(nxt, bm, g.M, klim);
      sb[(s + 1) % D].store(nxt + A_ELEMS, bn, g.N, klim);
      __syncthreads();
      cur ^= 1;
    }
  }

  finish_f32<TM, TN, KW>(g, acc, smem, bm, bn, wm, wn, lane, wave, kg, split, tile);
}

// ------------------------------------------------------------------------------------------------
// float32 kernel whose operand tiles come in by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no
// ds_write pass, and S - 1 K slabs in flight behind a counted s_waitcnt, so a workgroup that has a CU to itself (the
// usual case at batch 512) covers the load latency itself.  One raw s_barrier per slab (never __syncthreads(), whose
// fence would drain the loads in flight).  An LDS-DMA instruction writes the wave's 64 x 16 bytes
// contiguously, so the LDS images are unpadded and lane-linear:
//   k-major operand  [row][BK]: a row is 8 pieces of 16 bytes; slot (row, c) holds the piece q = c ^ ((row >> 1) & 7)
//                    of that row (the permutation is applied to the SOURCE address; the fragment read applies the
//                    same one), which spreads the 16 rows of a ds_read_b128 lane group over all 64 banks
//   m-major operand  [k][ROWS]: a piece is 4 consecutive rows at one k; fragments are ds_read_b32 along the rows
// Conditions (checked by the host): both operands 16-byte aligned with leading dimensions a multiple of 4 floats; the
// contiguous extent of each operand (K when k-major, its row count when m-major) a multiple of 4, so a piece is
// inside the matrix or outside it as a whole.  Pieces outside are requested at an out-of-range offset; the range
// check makes them zeros in LDS.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int ROWS, bool MMAJ, int NT>
struct DmaTile {
  static constexpr int R = ROWS * BK / 4 / NT;   // 16-byte pieces per thread and slab
  static_assert(R >= 1 && R * NT * 4 == ROWS * BK, "tile does not divide over the workgroup");
  unsigned off[R];   // byte offset of the piece in the slab to be issued next
  int kk[R];         // k of the piece inside a slab (1 << 30: the piece's rows are outside the matrix)
  unsigned step;     // advance per slab in bytes

  __device__ __forceinline__ void init(int t, int ld, int row0, int nrows, int k0) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int p = t + NT * j;
      if (!MMAJ) {
        const int row = p >> 3, q = (p & 7) ^ ((row >> 1) & 7), r = row0 + row;
        kk[j] = r < nrows ? 4 * q : (1 << 30);
        off[j] = ((unsigned)r * ld + k0 + 4 * q) * 4u;
      } else {
        const int k = p / (ROWS / 4), r = row0 + 4 * (p % (ROWS / 4));
        kk[j] = r < nrows ? k : (1 << 30);
        off[j] = ((unsigned)(k0 + k) * ld + r) * 4u;
      }
    }
    step = (MMAJ ? (unsigned)BK * ld : (unsigned)BK) * 4u;
  }
  // Issue the next slab into `dst` (this operand's image in an LDS stage); kleft = k values of the product that
  // remain from the slab's first k on (<= 0: the whole slab is outside, every piece reads as zeros).  lwave: index
  // of the issuing wave in the workgroup.
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, float* dst, int lwave, int kleft) {
#if defined(CG_ABL) && (CG_ABL & 1)     // timing-only build: no operand traffic
    return;
#endif
#pragma unroll
    for (int j = 0; j < R; ++j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + (j * NT + lwave * 64) * 4), 16,
                                               kk[j] < kleft ? off[j] : OOB_OFF, 0, 0, 0);
      off[j] += step;
    }
  }
};

// Every wave issues its share of the next slab's LDS-DMA loads right after the barrier that frees a stage.  (Tried and
// dropped: dedicated loader waves beside the computing waves, so that no computing wave ever issues a memory
// instruction in the K loop.  512 x 2048 x 5000 on 64 x 64 tiles: 135 us with one loader wave per workgroup - one wave
// cannot issue the slab's 16 pieces in a slab's time -, 127 us with two, 113 us with four, 109 us without any.)
template <int BM, int BN, bool TA, bool TB, int KW, int S>
__global__ void __launch_bounds__(256 * KW) gemm_f32_dma_kernel(GemmArgs g) {
  constexpr int NC = 256 * KW;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr bool A_MMAJ = TA, B_MMAJ = !TB;
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
  static_assert(S >= 3, "needs at least three LDS stages");
  static_assert(KW == 1 || TM * TN * 16 * 256 * (KW - 1) <= S * STAGE, "accumulator exchange does not fit in LDS");
  static_assert(BM * (BN + 4) <= S * STAGE, "the output tile (finish_f32, split-K tail) does not fit in LDS");
  __shared__ __align__(16) float smem[S * STAGE];      // the kernel's only LDS object

  int tile, split;
  xcd_tile_split(tile, split);
  int tile_m, tile_n;
  tile_coords(tile, (g.M + BM - 1) / BM, g.tiles_n, tile_m, tile_n);
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int nk_all = (g.K + BK - 1) / BK;
  const int kt0 = split * g.kt_per_split;
  const int nk = min(nk_all, kt0 + g.kt_per_split);
  const int nslab = nk - kt0;
  typedef DmaTile<BM, A_MMAJ, NC> TileA;
  typedef DmaTile<BN, B_MMAJ, NC> TileB;
  constexpr int LPS = TileA::R + TileB::R;   // LDS-DMA instructions per wave and slab
  static_assert((S - 1) * LPS <= 63, "more loads in flight than s_waitcnt vmcnt can count");
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.A, g.a_bytes), rsB = make_rsrc(g.B, g.b_bytes);
  int kleft = min(g.K, nk * BK) - kt0 * BK;      // k values this split still has to bring in
  const int lwave = threadIdx.x >> 6;
  TileA da;
  TileB db;
  da.init(threadIdx.x, g.lda, bm, g.M, kt0 * BK);
  db.init(threadIdx.x, g.ldb, bn, g.N, kt0 * BK);

  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, kg = threadIdx.x >> 8;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int li = lane & 31, lh = lane >> 5;
  constexpr int CG = BK / 8 / KW;   // groups of eight k per slab handled by one wave group
  static_assert(CG * KW * 8 == BK, "K slab does not divide over the wave groups");
  static_assert(CG % 2 == 0, "the fragment double buffer needs an even number of k groups per slab and wave group");

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Fragments of one group of eight k (TM + TN ds_read_b128 for k-major operands) and its 4 * TM * TN MFMAs.
  const int swz = (li >> 1) & 7;
  auto load_frag = [&](float (&a)[TM][4], float (&b)[TN][4], const float* As, int c) {
#if defined(CG_ABL) && (CG_ABL & 4)     // timing-only build without fragment reads
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(a[i][q]));
#pragma unroll
      for (int jj = 0; jj < TN; ++jj) asm volatile("" : "=v"(b[jj][q]));
    }
    return;
#endif
    const float* Bs = As + A_ELEMS;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (!A_MMAJ) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(As + (wm + 32 * i + li) * BK + 4 * ((2 * c + lh) ^ swz));
        a[i][0] = t[0]; a[i][1] = t[1]; a[i][2] = t[2]; a[i][3] = t[3];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[i][q] = As[(8 * c + 4 * lh + q) * BM + wm + 32 * i + li];
      }
    }
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
      if (!B_MMAJ) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(Bs + (wn + 32 * jj + li) * BK + 4 * ((2 * c + lh) ^ swz));
        b[jj][0] = t[0]; b[jj][1] = t[1]; b[jj][2] = t[2]; b[jj][3] = t[3];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) b[jj][q] = Bs[(8 * c + 4 * lh + q) * BN + wn + 32 * jj + li];
      }
    }
  };
  auto mfma_group = [&](const float (&a)[TM][4], const float (&b)[TN][4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][q], b[jj][q], acc[i][jj], 0, 0, 0);
  };

  // Stage j % S holds slab j.  All S stages are filled up front; when every wave has read slab j (the barrier that
  // also publishes slab j + 1) its stage is refilled with slab j + S, so S - 1 slabs are in flight behind the compute.
  // The fragments of a slab's first k group are read right after the barrier that publishes it, while the MFMAs of
  // the previous slab's last group are still to be issued: the matrix pipe has work across the barrier.
#pragma unroll
  for (int s = 0; s < S; ++s) {
    da.issue(rsA, smem + s * STAGE, lwave, kleft);
    db.issue(rsB, smem + s * STAGE + A_ELEMS, lwave, kleft);
    kleft -= BK;
  }
  float fa[2][TM][4], fb[2][TN][4];
  wait_vmcnt<(S - 1) * LPS>();          // slab 0 has landed for this wave ...
  __builtin_amdgcn_s_barrier();         // ... and for every wave
  load_frag(fa[0], fb[0], smem, kg * CG);
  int st_cur = 0;
  for (int j = 0; j < nslab; ++j) {
    const float* As = smem + st_cur * STAGE;
    const int st_next = st_cur + 1 == S ? 0 : st_cur + 1;
#pragma unroll
    for (int cc = 0; cc < CG; ++cc) {
      if (cc + 1 < CG) {
        load_frag(fa[(cc + 1) & 1], fb[(cc + 1) & 1], As, kg * CG + cc + 1);
      } else {
        // every read of slab j has been issued; once they are retired and all waves are here, slab j + 1 (landed for
        // this wave when all but the S - 2 youngest slabs' loads are done) is visible and stage st_cur is free
#if !(defined(CG_ABL) && (CG_ABL & 2))    // timing-only build without the per-slab barrier
        wait_vmcnt<(S - 2) * LPS>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
        da.issue(rsA, smem + st_cur * STAGE, lwave, kleft);
        db.issue(rsB, smem + st_cur * STAGE + A_ELEMS, lwave, kleft);
        kleft -= BK;
        load_frag(fa[0], fb[0], smem + st_next * STAGE, kg * CG);
      }
      // keep the reads of the next group ahead of this group's MFMAs (left alone, the compiler sinks them behind
      // the MFMAs to save registers, and every group then starts by waiting for its own reads)
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(fa[cc & 1], fb[cc & 1]);
    }
    st_cur = st_next;
  }
  // the loads still in flight are out-of-range ones (they write zeros into LDS): retire them before LDS is reused
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  finish_f32<TM, TN, KW>(g, acc, smem, bm, bn, wm, wn, lane, wave, kg, split, tile);
}

// Combine the split-K partials in split order and apply the epilogue of cg_gemm.  VEC4: four consecutive columns
// per thread with 16-byte accesses (N, every leading dimension and every base address a multiple of four floats).
template <bool VEC4>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmArgs g, int splits) {
  constexpr int W = VEC4 ? 4 : 1;
  const size_t total = (size_t)g.M * g.N / W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t e0 = i * W;
    combine_slabs<VEC4>(g, splits, (int)(e0 / g.N), (int)(e0 % g.N));
  }
}

// ------------------------------------------------------------------------------------------------
// Products with one tiny dimension (the critic's one-column head, rank-one products): a 64x64 MFMA tile would be
// mostly padding and a handful of workgroups would walk the whole K range alone.  Plain FMA kernels, exact float32, the same epilogues; sums in a fixed order.
// ------------------------------------------------------------------------------------------------
struct Strides {
  long am, ak, bk, bn;   // op(A)(m,k) = A[m*am + k*ak], op(B)(k,n) = B[k*bk + n*bn]
};

// K <= 16: one thread per output element.
__global__ void __launch_bounds__(256) gemm_small_k_kernel(GemmArgs g, Strides st) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / g.N), n = (int)(i % g.N);
    float h = g.bias ? g.bias[n] : 0.f;
    for (int k = 0; k < g.K; ++k) h += g.A[m * st.am + k * st.ak] * g.B[k * st.bk + n * st.bn];
    epilogue_store(g, m, n, h);
  }
}

// op(A) with k contiguous and few outputs (N <= 2: the critic's one-column head; or a handful of rows and up to 16
// columns: the embedding tables' gradient from the per-label sums, 12 x 12 with K = 1024): one wave per output row
// and pair of columns, lanes stride over k, butterfly sum over the lanes.
__global__ void __launch_bounds__(256) gemm_small_n_kernel(GemmArgs g, Strides st) {
  const int pairs = (g.N + 1) >> 1, w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, m = w / pairs, n0 = 2 * (w - m * pairs);
  if (m >= g.M) return;
  float acc[2] = {0.f, 0.f};
  const float* a = g.A + (size_t)m * st.am;
  const bool two = n0 + 1 < g.N;
#pragma unroll 4
  for (int k = lane; k < g.K; k += 64) {
    const float av = a[k * st.ak];
    const float* b = g.B + (size_t)k * st.bk + (size_t)n0 * st.bn;
    acc[0] += av * b[0];
    if (two) acc[1] += av * b[st.bn];
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    if (n0 + n >= g.N) break;
    float v = acc[n];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) epilogue_store(g, m, n0 + n, v + (g.bias ? g.bias[n0 + n] : 0.f));
  }
}

// 3 <= N <= 16, op(A) with k contiguous (the input gradient of the critic's embedding columns, dh0 * W1[:, G:]: 12
// columns, K = 1024): a 64-wide MFMA tile would be mostly padding and needs split-K plus a reduce launch to cover a
// few CUs.  Here a lane owns one output element - a wave covers 4 rows x 16 columns, a workgroup 16 rows - and walks
// the whole of K itself, so no sum crosses lanes (k-ordered chain, nothing to reduce).  Both operands go through LDS
// in chunks of 256 k: op(B) transposed to [column][k], A as [row][k], row stride 260 floats, so the 16 columns (and
// the 4 rows) of a ds_read_b128 start 4 banks apart and the 16 lanes of a row read one broadcast address.  The kernel
// is bound by load latency, not by bytes: a chunk's global loads are all issued at once, unconditionally (index
// clamped, value discarded), into registers while the previous chunk is multiplied out of LDS.
// VEC: A rows 16-byte aligned and K a multiple of 4.
constexpr int SN_KC = 256, SN_KS = SN_KC + 4, SN_ROWS = 16;
template <bool VEC>
__global__ void __launch_bounds__(256) gemm_small_n16_kernel(GemmArgs g, Strides st) {
  __shared__ __align__(16) float as[SN_ROWS * SN_KS];
  __shared__ __align__(16) float bs[16 * SN_KS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = lane & 15, rl = wave * 4 + (lane >> 4), row = blockIdx.x * SN_ROWS + rl;
  // staging roles: B - column tn, k = tk + 16 j; A - row tk, k = 4 (tn + 16 j) .. + 3 (VEC) or k = tn + 16 j
  const int tn = t & 15, tk = t >> 4, tnc = min(tn, g.N - 1);
  const float* arow = g.A + (size_t)min(blockIdx.x * SN_ROWS + tk, g.M - 1) * st.am;
  float vb[16], va[16];
  f32x4 va4[4];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int j = 0; j < 16; ++j) vb[j] = g.B[(size_t)min(kc + tk + 16 * j, g.K - 1) * st.bk + (size_t)tnc * st.bn];
    if (VEC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) va4[j] = *reinterpret_cast<const f32x4*>(arow + min(kc + 4 * (tn + 16 * j), g.K - 4));
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) va[j] = arow[min(kc + tn + 16 * j, g.K - 1)];
    }
  };
  auto stash = [&](int kc) {      // what lies past the end of K goes to LDS as zeros
    const int kn = g.K - kc;
#pragma unroll
    for (int j = 0; j < 16; ++j) bs[tn * SN_KS + tk + 16 * j] = (tn < g.N && tk + 16 * j < kn) ? vb[j] : 0.f;
    if (VEC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * (tn + 16 * j);
        *reinterpret_cast<f32x4*>(as + tk * SN_KS + k) = k < kn ? va4[j] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) as[tk * SN_KS + tn + 16 * j] = tn + 16 * j < kn ? va[j] : 0.f;
    }
  };
  float acc = 0.f;
  fetch(0);
  for (int kc = 0; kc < g.K; kc += SN_KC) {
    __syncthreads();      // every wave has finished reading the previous chunk
    stash(kc);
    __syncthreads();
    fetch(kc + SN_KC);    // the next chunk (clamped re-reads once K is exhausted), in flight during the products below
    const float* al = as + rl * SN_KS;
    const float* bl = bs + n * SN_KS;
#pragma unroll 8
    for (int q = 0; q < SN_KC / 4; ++q) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(al + 4 * q);
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 4 * q);
      acc = fmaf(a[0], b[0], acc);
      acc = fmaf(a[1], b[1], acc);
      acc = fmaf(a[2], b[2], acc);
      acc = fmaf(a[3], b[3], acc);
    }
  }
  if (row < g.M && n < g.N) epilogue_store(g, row, n, acc + (g.bias ? g.bias[n] : 0.f));
}

// M <= 2 with A stored [K][M] and B stored [K][N] (a weight gradient of the critic's one-column head: dval^T a3): a
// workgroup owns 4 columns, its 64 k lanes stride over k (eight loads in flight per lane, unconditional: index
// clamped, value discarded), partial sums are added in lane order.  (32 columns per workgroup and one load in flight
// per lane took 14 us for 256 columns and K = 1024: eight workgroups, each a chain of 32 dependent load waits.)
constexpr int SM_CW = 4, SM_KL = 64, SM_UN = 8;
__global__ void __launch_bounds__(256) gemm_small_m_kernel(GemmArgs g) {
  __shared__ float lds[2][SM_KL][SM_CW];
  const int c = threadIdx.x % SM_CW, kl = threadIdx.x / SM_CW, col = blockIdx.x * SM_CW + c;
  const int colc = min(col, g.N - 1);
  float acc[2] = {0.f, 0.f};
  for (int k0 = kl; k0 < g.K; k0 += SM_KL * SM_UN) {
    float b[SM_UN], a[SM_UN][2];
#pragma unroll
    for (int u = 0; u < SM_UN; ++u) {
      const int k = min(k0 + u * SM_KL, g.K - 1);
      b[u] = g.B[(size_t)k * g.ldb + colc];
      a[u][0] = g.A[(size_t)k * g.lda];
      a[u][1] = g.A[(size_t)k * g.lda + (g.M > 1 ? 1 : 0)];
    }
#pragma unroll
    for (int u = 0; u < SM_UN; ++u) {
      const float bv = k0 + u * SM_KL < g.K ? b[u] : 0.f;
      acc[0] += a[u][0] * bv;
      acc[1] += a[u][1] * bv;
    }
  }
  lds[0][kl][c] = acc[0];
  lds[1][kl][c] = acc[1];
  __syncthreads();
  const int m = threadIdx.x / SM_CW;
  if (m < g.M && col < g.N) {
    float h = g.bias ? g.bias[col] : 0.f;
    for (int r = 0; r < SM_KL; ++r) h += lds[m][r][c];
    epilogue_store(g, m, col, h);
  }
}

// Returns true when one of the kernels above took the product.
bool launch_small(const GemmArgs& g, int transA, int transB, hipStream_t st) {
  const Strides s{transA ? 1 : g.lda, transA ? g.lda : 1, transB ? 1 : g.ldb, transB ? g.ldb : 1};
  if (g.K <= 16) {
    size_t blocks = ((size_t)g.M * g.N + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    CG_GEMM_LAUNCH(true, gemm_small_k_kernel, dim3((unsigned)blocks), dim3(256), st, g, s);
    return true;
  }
  if (g.M <= 2 && transA && !transB) {
    CG_GEMM_LAUNCH(true, gemm_small_m_kernel, dim3((g.N + SM_CW - 1) / SM_CW), dim3(SM_CW * SM_KL), st, g);
    return true;
  }
  // a few columns: every wave reads B once per row it owns, which stays cheap while B is a few vectors (measured:
  // the one-column head 7 us against 13 us for the tiled kernel; twelve columns 40 us against 13 us)
  if (!transA && (g.N <= 2 || (g.N <= 16 && g.M <= 32))) {
    const int waves = g.M * ((g.N + 1) / 2);
    CG_GEMM_LAUNCH(true, gemm_small_n_kernel, dim3((waves + 3) / 4), dim3(256), st, g, s);
    return true;
  }
  if (g.N <= 16 && !transA) {
    const bool v4 = (uintptr_t)g.A % 16 == 0 && g.lda % 4 == 0 && g.K % 4 == 0;
    if (v4) CG_GEMM_LAUNCH(true, gemm_small_n16_kernel<true>, dim3((g.M + SN_ROWS - 1) / SN_ROWS), dim3(256), st, g, s);
    else CG_GEMM_LAUNCH(true, gemm_small_n16_kernel<false>, dim3((g.M + SN_ROWS - 1) / SN_ROWS), dim3(256), st, g, s);
    return true;
  }
  return false;
}

// May the pass that combines split-K slabs use 16-byte accesses?
bool reduce_vec4(const GemmArgs& g) {
  auto al = [](const void* p, int ld) { return p == nullptr || (((uintptr_t)p % 16 == 0) && ld % 4 == 0); };
  return g.N % 4 == 0 && al(g.C, g.ldc) && al(g.E, g.lde) && al(g.aux, g.ldaux) && al(g.ws, 4) && (uintptr_t)g.C16 % 8 == 0;
}

// Tile arrival counters at the end of a workspace handed in with CG_WS_TILE_COUNTERS, and whether a product of
// `tiles` output tiles combines its `splits` slabs itself (split_tail) instead of by a second launch.
constexpr size_t WS_COUNTER_BYTES = 16384;
bool fuse_split_tail(GemmArgs& a, long tiles, int splits, const int* keep) {
  static const bool separate = getenv("CGRNA_GEMM_SEPARATE_REDUCE") != nullptr;     // A/B aid
  const bool fuse = !keep && splits > 1 && a.counters && !separate && tiles <= (long)(WS_COUNTER_BYTES / sizeof(unsigned));
  if (!fuse) a.counters = nullptr;
  a.vec4 = reduce_vec4(a);
  return fuse;
}

void launch_splitk_reduce(const GemmArgs& g, int splits, hipStream_t st) {
  const size_t total = (size_t)g.M * g.N;
  const bool v4 = reduce_vec4(g);
  int blocks = (int)((total / (v4 ? 4 : 1) + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (v4) CG_GEMM_LAUNCH(true, splitk_reduce_kernel<true>, dim3(blocks), dim3(256), st, g, splits);
  else CG_GEMM_LAUNCH(true, splitk_reduce_kernel<false>, dim3(blocks), dim3(256), st, g, splits);
}

// keep != nullptr: the caller combines the split-K slabs itself (cg_gemm_slabs): no reduce launch, *keep = slabs
template <int BM, int BN, bool HALF = false, bool AH = false, bool BH = false>
int launch_f32(const GemmArgs& g, int transA, int transB, bool vec, hipStream_t st, int splits = 1, int* keep = nullptr) {
  GemmArgs a = g;
  const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
  a.tiles_n = tn;
  const int bk = HALF ? BKH : BK;
  const int nk = (g.K + bk - 1) / bk;
  a.kt_per_split = (nk + splits - 1) / splits;
  splits = (nk + a.kt_per_split - 1) / a.kt_per_split;   // no empty split
  constexpr int KW = HALF ? kwaves_bf16(BM, BN) : kwaves(BM, BN);
  dim3 grid(tm * tn, splits), block(256 * KW);
  const bool fused = fuse_split_tail(a, (long)tm * tn, splits, keep);
  const bool last = keep || splits == 1 || fused;      // no slab reduce follows
#define CG_LAUNCH(TA, TB)                                                                    \
  do {                                                                                       \
    if (HALF) {                                                                              \
      if (vec || AH || BH) CG_GEMM_LAUNCH(last, (gemm_bf16_kernel<BM, BN, TA, TB, true, KW, AH, BH>), grid, block, st, a);   \
      else CG_GEMM_LAUNCH(last, (gemm_bf16_kernel<BM, BN, TA, TB, false, KW, false, false>), grid, block, st, a);            \
    } else {                                                                                 \
      if (vec) CG_GEMM_LAUNCH(last, (gemm_f32_kernel<BM, BN, TA, TB, true, KW>), grid, block, st, a);    \
      else CG_GEMM_LAUNCH(last, (gemm_f32_kernel<BM, BN, TA, TB, false, KW>), grid, block, st, a);       \
    }                                                                                        \
  } while (0)
  if (!transA && transB) CG_LAUNCH(false, true);
  else if (!transA && !transB) CG_LAUNCH(false, false);
  else if (transA && !transB) CG_LAUNCH(true, false);
  else CG_LAUNCH(true, true);
#undef CG_LAUNCH
  if (keep) *keep = splits;
  else if (splits > 1 && !fused) launch_splitk_reduce(g, splits, st);
  return cg_launch_status();
}

#ifndef CG_DKW_SMALL
#define CG_DKW_SMALL 2
#define CG_DKW_MID 2
#define CG_DKW_BIG 2
#endif
#ifndef CG_DMA_STAGES
#define CG_DMA_STAGES 3
#endif
constexpr int kwaves_dma(int bm, int bn) { return bm * bn <= 64 * 64 ? CG_DKW_SMALL : (bm * bn <= 128 * 64 ? CG_DKW_MID : CG_DKW_BIG); }

// every variant is instantiated explicitly: launched only from inside the function template below, some of them
// were left without a host-side launch stub by this compiler
#define CG_DMA_INST(BM, BN)                                                                                              \
  template __global__ void gemm_f32_dma_kernel<BM, BN, false, true, kwaves_dma(BM, BN), CG_DMA_STAGES>(GemmArgs);   \
  template __global__ void gemm_f32_dma_kernel<BM, BN, false, false, kwaves_dma(BM, BN), CG_DMA_STAGES>(GemmArgs);  \
  template __global__ void gemm_f32_dma_kernel<BM, BN, true, false, kwaves_dma(BM, BN), CG_DMA_STAGES>(GemmArgs);   \
  template __global__ void gemm_f32_dma_kernel<BM, BN, true, true, kwaves_dma(BM, BN), CG_DMA_STAGES>(GemmArgs);
CG_DMA_INST(128, 128)
CG_DMA_INST(128, 64)
CG_DMA_INST(64, 64)
#undef CG_DMA_INST

template <int BM, int BN>
int launch_f32_dma(const GemmArgs& g, int transA, int transB, hipStream_t st, int splits, int* keep) {
  GemmArgs a = g;
  const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
  a.tiles_n = tn;
  const int nk = (g.K + BK - 1) / BK;
  a.kt_per_split = (nk + splits - 1) / splits;
  splits = (nk + a.kt_per_split - 1) / a.kt_per_split;   // no empty split
  constexpr int KW = kwaves_dma(BM, BN), S = CG_DMA_STAGES;
  dim3 grid(tm * tn, splits), block(256 * KW);
  const bool fused = fuse_split_tail(a, (long)tm * tn, splits, keep);
  const bool last = keep || splits == 1 || fused;      // no slab reduce follows
  if (!transA && transB) CG_GEMM_LAUNCH(last, (gemm_f32_dma_kernel<BM, BN, false, true, KW, S>), grid, block, st, a);
  else if (!transA && !transB) CG_GEMM_LAUNCH(last, (gemm_f32_dma_kernel<BM, BN, false, false, KW, S>), grid, block, st, a);
  else if (transA && !transB) CG_GEMM_LAUNCH(last, (gemm_f32_dma_kernel<BM, BN, true, false, KW, S>), grid, block, st, a);
  else CG_GEMM_LAUNCH(last, (gemm_f32_dma_kernel<BM, BN, true, true, KW, S>), grid, block, st, a);
  if (keep) *keep = splits;
  else if (splits > 1 && !fused) launch_splitk_reduce(g, splits, st);
  return cg_launch_status();
}

int gemm_dispatch(float* C, int ldc, const void* A, int lda, const void* B, int ldb,
                  int M, int N, int K, int transA, int transB, int dtype,
                  const float* bias, int epi, const float* E, int lde, float* aux, int ldaux,
                  float slope, int accumulate, void* workspace, size_t workspace_bytes, void* stream, int* keep,
                  void* c_bf16 = nullptr) {
  timing_begin_call();
  if (!C || !A || !B || M < 0 || N < 0 || K < 0) return CG_ERR_BAD_ARG;
  if (c_bf16 && (keep || (uintptr_t)c_bf16 % 2)) return CG_ERR_BAD_ARG;
  if (M == 0 || N == 0) return CG_OK;
  if (K == 0) return CG_ERR_BAD_ARG;      // an empty sum has no tile to launch; callers never need it
  if (epi < CG_EPI_NONE || epi > CG_EPI_MUL) return CG_ERR_BAD_ARG;
  if (epi == CG_EPI_MUL && !E) return CG_ERR_BAD_ARG;
  if (epi != CG_EPI_NONE && accumulate) return CG_ERR_BAD_ARG;
  const int a_rows = transA ? K : M, a_cols = transA ? M : K;
  const int b_rows = transB ? N : K, b_cols = transB ? K : N;
  if (lda < a_cols || ldb < b_cols || ldc < N) return CG_ERR_BAD_ARG;
  if (E && lde < N) return CG_ERR_BAD_ARG;
  if (aux && ldaux < N) return CG_ERR_BAD_ARG;
  if (dtype < 0 || dtype > (CG_BF16 | CG_A_IS_BF16 | CG_B_IS_BF16 | CG_WS_TILE_COUNTERS)) return CG_ERR_UNSUPPORTED;
  unsigned* counters = nullptr;
  if (dtype & CG_WS_TILE_COUNTERS) {      // the workspace ends with the tile arrival counters (see cgrna.h)
    if (!workspace || workspace_bytes < 2 * WS_COUNTER_BYTES || workspace_bytes % 16 || (uintptr_t)workspace % 16)
      return CG_ERR_BAD_ARG;
    workspace_bytes -= WS_COUNTER_BYTES;
    counters = (unsigned*)((char*)workspace + workspace_bytes);
    dtype &= ~CG_WS_TILE_COUNTERS;
  }
  const bool half = (dtype & CG_BF16) != 0, ah = (dtype & CG_A_IS_BF16) != 0, bh = (dtype & CG_B_IS_BF16) != 0;
  if ((ah || bh) && !half) return CG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const size_t a_bytes = ((size_t)(a_rows - 1) * lda + a_cols) * (ah ? 2 : sizeof(float));
  const size_t b_bytes = ((size_t)(b_rows - 1) * ldb + b_cols) * (bh ? 2 : sizeof(float));
  // a bf16 operand: 8-byte aligned rows, even contiguous extent (see StagerB16)
  if (ah && (((uintptr_t)A % 8) || lda % 4 || a_cols % 2)) return CG_ERR_UNSUPPORTED;
  if (bh && (((uintptr_t)B % 8) || ldb % 4 || b_cols % 2)) return CG_ERR_UNSUPPORTED;
  if (a_bytes >= OOB_OFF || b_bytes >= OOB_OFF) return CG_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  GemmArgs g{(const float*)A, (const float*)B, C, bias, E, aux, M, N, K, lda, ldb, ldc, lde, ldaux,
             epi, accumulate, slope, 0, 1 << 30, (float*)workspace, (unsigned)a_bytes, (unsigned)b_bytes, (u16*)c_bf16,
             counters, 0};
  const bool tiny = K <= 16 || N <= 2 || (M <= 2 && transA && !transB) || (N <= 16 && !transA);
  if (tiny && (ah || bh)) return CG_ERR_UNSUPPORTED;      // the plain-FMA kernels read float32
  if (tiny && launch_small(g, transA, transB, st)) {
    if (keep) *keep = 1;
    return cg_launch_status();
  }
  // 16-byte loads of the float32 operands (a bf16 operand has its own conditions, checked above)
  const bool vec = (ah || ((uintptr_t)A % 16 == 0 && lda % 4 == 0)) && (bh || ((uintptr_t)B % 16 == 0 && ldb % 4 == 0));
  // Tile and split-K choice by a cost model fitted to a sweep of the step's shapes on MI355X (microseconds).
  // A workgroup alone on a CU needs L per K slab (latency-bound); n workgroups sharing a CU overlap and finish a
  // slab every S (throughput-bound), so the busiest CU takes slabs * max(L, n*S).  Small tiles have the worse S but
  // fill 256 CUs evenly, which decides every shape of this workload except the very large ones.  Splitting K adds
  // a fixed cost and the round trip of the partial sums.
  const int nk = half ? (K + BKH - 1) / BKH : (K + BK - 1) / BK;
  const size_t per = (size_t)M * N * sizeof(float);
  struct Cand { int bm, bn; double L, S, P; };
  static const Cand cand_f32[3] = {{128, 128, 2.275, 2.057, 8.26}, {128, 64, 0.733, 1.081, 7.08}, {64, 64, 0.591, 0.564, 5.61}};
  static const Cand cand_bf16[3] = {{128, 128, 1.2, 0.65, 6.0}, {128, 64, 0.8, 0.45, 4.0}, {64, 64, 0.9, 0.25, 2.5}};
  const Cand* cands = half ? cand_bf16 : cand_f32;
  static const int split_choices[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  int best = 2, best_splits = 1;
  double best_cost = 1e300;
  // tuning aids (tools/): read from the environment on every call only when CGRNA_GEMM_TUNING was set at the first call
  static const bool tuning = getenv("CGRNA_GEMM_TUNING") != nullptr;
  double tile_bias[3] = {1.0, 1.0, 1.0};
  if (tuning)
    if (const char* f = getenv("CGRNA_GEMM_BIAS")) sscanf(f, "%lf,%lf,%lf", &tile_bias[0], &tile_bias[1], &tile_bias[2]);
  for (int c = 0; c < 3; ++c) {
    const long tiles = (long)((M + cands[c].bm - 1) / cands[c].bm) * ((N + cands[c].bn - 1) / cands[c].bn);
    for (int sp : split_choices) {
      if (sp > 1 && (!workspace || (size_t)sp * per > workspace_bytes || nk / sp < 2)) break;
      const long blocks = tiles * sp;
      const long n = (blocks + 255) / 256;
      const int slabs = (nk + sp - 1) / sp;
      double cost = slabs * (n * cands[c].S > cands[c].L ? n * cands[c].S : cands[c].L) + cands[c].P * (accumulate ? 1.5 : 1.0);
      if (sp > 1) cost += (half ? 6.0 + (sp + 1) * (double)per / 4.0e6 : 3.36 + (sp + 1) * (double)per / 15.9e6);
      cost *= tile_bias[c];
      if (cost < best_cost) { best_cost = cost; best = c; best_splits = sp; }
    }
  }
  if (workspace) {      // shapes that were swept on the device: take the measured best
    const bool use_table = !tuning || getenv("CGRNA_GEMM_NO_TABLE") == nullptr;
    const TunedGemm* tab = half ? tuned_bf16 : tuned_f32;
    const size_t ntab = half ? sizeof(tuned_bf16) / sizeof(TunedGemm) : sizeof(tuned_f32) / sizeof(TunedGemm);
    for (size_t i = 0; use_table && i < ntab; ++i) {
      const TunedGemm& t = tab[i];
      if (t.M == M && t.N == N && t.K == K && t.ta == transA && t.tb == transB && t.acc == accumulate &&
          (t.splits == 1 || (size_t)t.splits * per <= workspace_bytes)) {
        best = t.tile;
        best_splits = t.splits;
        break;
      }
    }
  }
  if (const char* f = tuning ? getenv("CGRNA_GEMM_OVERRIDE") : nullptr) {   // tuning aid: "M,N,K,transA,transB,accumulate,tile,splits" for one shape
    int m, n, k, ta, tb, ac, t = 0, sp = 1;
    if (sscanf(f, "%d,%d,%d,%d,%d,%d,%d,%d", &m, &n, &k, &ta, &tb, &ac, &t, &sp) == 8 && m == M && n == N && k == K &&
        ta == transA && tb == transB && ac == accumulate && t >= 0 && t < 3 && sp >= 1 &&
        (sp == 1 || (workspace && (size_t)sp * per <= workspace_bytes && nk / sp >= 1))) {
      best = t;
      best_splits = sp;
    }
  }
  if (const char* f = tuning ? getenv("CGRNA_GEMM_FORCE") : nullptr) {   // tuning aid: "tile,splits" with tile 0=128x128 1=128x64 2=64x64
    int t = 0, sp = 1;
    if (sscanf(f, "%d,%d", &t, &sp) == 2 && t >= 0 && t < 3 && sp >= 1 && (sp == 1 || (workspace && (size_t)sp * per <= workspace_bytes))) {
      best = t;
      best_splits = sp;
    }
  }
  // LDS-DMA staging needs every 16-byte piece of a tile to lie inside the matrix or outside it as a whole
  const bool dma = !half && vec && (transA ? M % 4 == 0 : K % 4 == 0) && (transB ? K % 4 == 0 : N % 4 == 0) &&
                   !(tuning && getenv("CGRNA_GEMM_NO_DMA"));
  if (dma) {
    if (best == 0) return launch_f32_dma<128, 128>(g, transA, transB, st, best_splits, keep);
    if (best == 1) return launch_f32_dma<128, 64>(g, transA, transB, st, best_splits, keep);
    return launch_f32_dma<64, 64>(g, transA, transB, st, best_splits, keep);
  }
  if (half) {
#define CG_HALF_TILES(AH, BH)                                                                            \
  do {                                                                                                   \
    if (best == 0) return launch_f32<128, 128, true, AH, BH>(g, transA, transB, vec, st, best_splits, keep);   \
    if (best == 1) return launch_f32<128, 64, true, AH, BH>(g, transA, transB, vec, st, best_splits, keep);    \
    return launch_f32<64, 64, true, AH, BH>(g, transA, transB, vec, st, best_splits, keep);                    \
  } while (0)
    if ((ah || bh) && !vec) return CG_ERR_UNSUPPORTED;     // the float32 partner of a bf16 operand must allow 16-byte loads
    if (ah && bh) CG_HALF_TILES(true, true);
    if (ah) CG_HALF_TILES(true, false);
    if (bh) CG_HALF_TILES(false, true);
    CG_HALF_TILES(false, false);
#undef CG_HALF_TILES
  }
  if (best == 0) return launch_f32<128, 128>(g, transA, transB, vec, st, best_splits, keep);
  if (best == 1) return launch_f32<128, 64>(g, transA, transB, vec, st, best_splits, keep);
  return launch_f32<64, 64>(g, transA, transB, vec, st, best_splits, keep);
}

}  // namespace

extern "C" int cg_gemm(float* C, int ldc, const void* A, int lda, const void* B, int ldb,
                       int M, int N, int K, int transA, int transB, int dtype,
                       const float* bias, int epi, const float* E, int lde, float* aux, int ldaux,
                       float slope, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_dispatch(C, ldc, A, lda, B, ldb, M, N, K, transA, transB, dtype, bias, epi, E, lde, aux, ldaux, slope,
                       accumulate, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int cg_gemm_ex(float* C, int ldc, const void* A, int lda, const void* B, int ldb,
                          int M, int N, int K, int transA, int transB, int dtype,
                          const float* bias, int epi, const float* E, int lde, float* aux, int ldaux,
                          float slope, int accumulate, void* c_bf16, void* workspace, size_t workspace_bytes, void* stream) {
  return gemm_dispatch(C, ldc, A, lda, B, ldb, M, N, K, transA, transB, dtype, bias, epi, E, lde, aux, ldaux, slope,
                       accumulate, workspace, workspace_bytes, stream, nullptr, c_bf16);
}

extern "C" int cg_gemm_slabs(float* C, int ldc, const void* A, int lda, const void* B, int ldb,
                             int M, int N, int K, int transA, int transB, int dtype, const float* bias,
                             void* workspace, size_t workspace_bytes, int* n_slabs, void* stream) {
  if (!n_slabs) return CG_ERR_BAD_ARG;
  *n_slabs = 1;
  if (M == 0 || N == 0) return CG_OK;
  return gemm_dispatch(C, ldc, A, lda, B, ldb, M, N, K, transA, transB, dtype, bias, CG_EPI_NONE, nullptr, 0, nullptr, 0,
                       0.f, 0, workspace, workspace_bytes, stream, n_slabs);
}

extern "C" int cg_gemm_timing_begin(int max_calls) {
  GemmTiming& t = tl_timing;
  if (t.open || max_calls <= 0 || max_calls > (1 << 20)) return CG_ERR_BAD_ARG;
  t.ev.assign(2 * (size_t)max_calls, nullptr);
  t.state.assign((size_t)max_calls, 0);
  for (auto& e : t.ev) {
    hipError_t err = hipEventCreate(&e);
    if (err != hipSuccess) {
      for (auto& d : t.ev)
        if (d) (void)hipEventDestroy(d);
      t.ev.clear();
      return (int)err;
    }
  }
  t.calls = 0;
  t.open = true;
  return CG_OK;
}

extern "C" int cg_gemm_timing_end(float* ms, int capacity) {
  GemmTiming& t = tl_timing;
  if (!t.open || (capacity > 0 && !ms)) return CG_ERR_BAD_ARG;
  t.open = false;
  t.start = t.stop = nullptr;
  hipError_t err = hipDeviceSynchronize();
  const int n = (int)t.calls;
  for (int i = 0; i < n && i < capacity; ++i) {
    ms[i] = -1.f;       // a call that launched nothing, or whose events could not be read
    if (t.state[i] == 3 && err == hipSuccess) {
      float v = 0.f;
      if (hipEventElapsedTime(&v, t.ev[2 * i], t.ev[2 * i + 1]) == hipSuccess) ms[i] = v;
    }
  }
  for (auto& e : t.ev)
    if (e) (void)hipEventDestroy(e);
  t.ev.clear();
  t.state.clear();
  t.calls = 0;
  return err == hipSuccess ? n : -(int)err - 100;
}
