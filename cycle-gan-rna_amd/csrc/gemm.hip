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
#include "common.h"

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
  int kt_per_split;   // K slabs handled by one blockIdx.y (split-K); all of them when gridDim.y == 1
  float* ws;          // split-K: partial results [gridDim.y][M][N]
};

constexpr int BK = 32;

// Guarded 4-element load along the contiguous dimension.
template <bool VEC>
__device__ __forceinline__ f32x4 load4(const float* p, bool outer_ok, int c0, int climit) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!outer_ok) return v;
  if (VEC && c0 + 3 < climit) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c0 + e < climit) v[e] = p[e];
  }
  return v;
}

// Stage registers for one operand tile of ROWS x BK.
//  k-major: element (r,k) = ptr[(row0+r)*ld + k0+k]
//  m-major: element (r,k) = ptr[(k0+k)*ld + row0+r]
template <int ROWS, bool MMAJ, bool VEC>
struct Stager {
  static constexpr int NV = ROWS * BK / 4 / 256;  // float4 per thread
  f32x4 v[NV];

  __device__ __forceinline__ void load(const float* ptr, int ld, int row0, int nrows, int k0, int K) {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + 256 * j;
      if (!MMAJ) {
        const int r = idx / (BK / 4), kq = idx % (BK / 4);
        const int row = row0 + r, k = k0 + 4 * kq;
        v[j] = load4<VEC>(ptr + (size_t)row * ld + k, row < nrows, k, K);
      } else {
        const int k = idx / (ROWS / 4), rq = idx % (ROWS / 4);
        const int row = row0 + 4 * rq, kk = k0 + k;
        v[j] = load4<VEC>(ptr + (size_t)kk * ld + row, kk < K, row, nrows);
      }
    }
  }
  __device__ __forceinline__ void store(float* s) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int idx = t + 256 * j;
      if (!MMAJ) {
        const int r = idx / (BK / 4), kq = idx % (BK / 4);
        *reinterpret_cast<f32x4*>(s + r * (BK + 4) + 4 * kq) = v[j];
      } else {
        const int k = idx / (ROWS / 4), rq = idx % (ROWS / 4);
        *reinterpret_cast<f32x4*>(s + k * (ROWS + 4) + 4 * rq) = v[j];
      }
    }
  }
};

// One output element through the epilogue (bias already added by the caller where it applies).
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int row, int col, float h) {
  float* cp = g.C + (size_t)row * g.ldc + col;
  if (g.epi == CG_EPI_NONE) {
    if (g.accumulate) h += *cp;
    *cp = h;
  } else if (g.epi == CG_EPI_LEAKY) {
    float pm = h > 0.f ? 1.f : g.slope;
    if (g.E) pm *= g.E[(size_t)row * g.lde + col];
    *cp = h * pm;
    if (g.aux) g.aux[(size_t)row * g.ldaux + col] = pm;
  } else {
    *cp = h * g.E[(size_t)row * g.lde + col];
  }
}

// XCD-aware tile index: workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of
// tiles (neighbours share an A row panel and adjacent B panels) so panel re-reads hit that XCD's L2.  Bijective.
__device__ __forceinline__ int xcd_tile() {
  const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
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
// with v_mfma_f32_16x16x32_bf16 (float32 accumulate).  Both operand kinds end up in LDS as
// [row][k] bf16 with k contiguous (row stride 80 elements = 160 B: conflict-free ds_read_b128), so
// one fragment read serves a whole 32-deep MFMA; m-major operands are transposed in registers
// (4k x 4row micro-tiles) on their way in.  BKH = 64 per slab, double-buffered.
// ------------------------------------------------------------------------------------------------
constexpr int BKH = 64;
constexpr int LDH = BKH + 16;   // LDS row stride in bf16 elements

__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  uint2 r;
  r.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  r.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  return r;
}

template <int ROWS, bool MMAJ, bool VEC>
struct StagerH {
  static constexpr int NV = ROWS * BKH / 4 / 256;   // float4 per thread (8 for 128 rows, 4 for 64)
  f32x4 v[NV];

  __device__ __forceinline__ void load(const float* ptr, int ld, int row0, int nrows, int k0, int K) {
    const int t = threadIdx.x;
    if (!MMAJ) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int idx = t + 256 * j;
        const int r = idx / (BKH / 4), kq = idx % (BKH / 4);
        const int row = row0 + r, k = k0 + 4 * kq;
        v[j] = load4<VEC>(ptr + (size_t)row * ld + k, row < nrows, k, K);
      }
    } else {
      // micro-tile mt = 4 k-rows x 4 rows; v[4*u + i] = k-row 4*kq+i of micro-tile u
#pragma unroll
      for (int u = 0; u < NV / 4; ++u) {
        const int idx = t + 256 * u;
        const int kq = idx / (ROWS / 4), rq = idx % (ROWS / 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = k0 + 4 * kq + i, row = row0 + 4 * rq;
          v[4 * u + i] = load4<VEC>(ptr + (size_t)kk * ld + row, kk < K, row, nrows);
        }
      }
    }
  }
  __device__ __forceinline__ void store(u16* s) const {
    const int t = threadIdx.x;
    if (!MMAJ) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int idx = t + 256 * j;
        const int r = idx / (BKH / 4), kq = idx % (BKH / 4);
        *reinterpret_cast<uint2*>(s + r * LDH + 4 * kq) = pack4(v[j][0], v[j][1], v[j][2], v[j][3]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < NV / 4; ++u) {
        const int idx = t + 256 * u;
        const int kq = idx / (ROWS / 4), rq = idx % (ROWS / 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<uint2*>(s + (4 * rq + c) * LDH + 4 * kq) =
              pack4(v[4 * u][c], v[4 * u + 1][c], v[4 * u + 2][c], v[4 * u + 3][c]);
      }
    }
  }
};

template <int BM, int BN, bool TA, bool TB, bool VEC>
__global__ void __launch_bounds__(256) gemm_bf16_kernel(GemmArgs g) {
  constexpr int TM = BM / 32, TN = BN / 32;          // 16x16 MFMA tiles per wave in each direction
  constexpr bool A_MMAJ = TA, B_MMAJ = !TB;
  constexpr int A_ELEMS = BM * LDH, B_ELEMS = BN * LDH;
  __shared__ __align__(16) u16 smem[2 * (A_ELEMS + B_ELEMS)];

  const int tile = xcd_tile();
  int tile_m, tile_n;
  tile_coords(tile, (g.M + BM - 1) / BM, g.tiles_n, tile_m, tile_n);
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int lr = lane & 15, lg = lane >> 4;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  StagerH<BM, A_MMAJ, VEC> sa;
  StagerH<BN, B_MMAJ, VEC> sb;
  const int nk_all = (g.K + BKH - 1) / BKH;
  const int kt0 = blockIdx.y * g.kt_per_split;
  const int nk = min(nk_all, kt0 + g.kt_per_split);

  sa.load(g.A, g.lda, bm, g.M, kt0 * BKH, g.K);
  sb.load(g.B, g.ldb, bn, g.N, kt0 * BKH, g.K);
  sa.store(smem);
  sb.store(smem + A_ELEMS);
  __syncthreads();

  int cur = 0;
  for (int kt = kt0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      sa.load(g.A, g.lda, bm, g.M, (kt + 1) * BKH, g.K);
      sb.load(g.B, g.ldb, bn, g.N, (kt + 1) * BKH, g.K);
    }
    const u16* As = smem + cur * (A_ELEMS + B_ELEMS);
    const u16* Bs = As + A_ELEMS;
#pragma unroll
    for (int ks = 0; ks < BKH / 32; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(As + (wm + 16 * i + lr) * LDH + 32 * ks + 8 * lg);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn + 16 * j + lr) * LDH + 32 * ks + 8 * lg);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      u16* nxt = smem + (cur ^ 1) * (A_ELEMS + B_ELEMS);
      sa.store(nxt);
      sb.store(nxt + A_ELEMS);
    }
    __syncthreads();
    cur ^= 1;
  }

  // 16x16 accumulator: col = lane & 15, row = 4 * (lane >> 4) + reg
  float* wsp = gridDim.y > 1 ? g.ws + (size_t)blockIdx.y * g.M * g.N : nullptr;
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
        if (wsp) wsp[(size_t)row * g.N + col] = acc[i][j][r];
        else epilogue_store(g, row, col, acc[i][j][r] + bv);
      }
    }
}

template <int BM, int BN, bool TA, bool TB, bool VEC>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs g) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr bool A_MMAJ = TA, B_MMAJ = !TB;
  constexpr int A_ELEMS = A_MMAJ ? BK * (BM + 4) : BM * (BK + 4);
  constexpr int B_ELEMS = B_MMAJ ? BK * (BN + 4) : BN * (BK + 4);
  __shared__ __align__(16) float smem[2 * (A_ELEMS + B_ELEMS)];

  const int tile = xcd_tile();
  int tile_m, tile_n;
  tile_coords(tile, (g.M + BM - 1) / BM, g.tiles_n, tile_m, tile_n);
  const int bm = tile_m * BM, bn = tile_n * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<BM, A_MMAJ, VEC> sa;
  Stager<BN, B_MMAJ, VEC> sb;
  const int nk_all = (g.K + BK - 1) / BK;
  const int kt0 = blockIdx.y * g.kt_per_split;
  const int nk = min(nk_all, kt0 + g.kt_per_split);

  sa.load(g.A, g.lda, bm, g.M, kt0 * BK, g.K);
  sb.load(g.B, g.ldb, bn, g.N, kt0 * BK, g.K);
  sa.store(smem);
  sb.store(smem + A_ELEMS);
  __syncthreads();

  int cur = 0;
  for (int kt = kt0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      sa.load(g.A, g.lda, bm, g.M, (kt + 1) * BK, g.K);
      sb.load(g.B, g.ldb, bn, g.N, (kt + 1) * BK, g.K);
    }
    const float* As = smem + cur * (A_ELEMS + B_ELEMS);
    const float* Bs = As + A_ELEMS;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      float a[TM][4], b[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (!A_MMAJ) {
          f32x4 t = *reinterpret_cast<const f32x4*>(As + (wm + 32 * i + li) * (BK + 4) + 8 * c + 4 * lh);
          a[i][0] = t[0]; a[i][1] = t[1]; a[i][2] = t[2]; a[i][3] = t[3];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) a[i][s] = As[(8 * c + 4 * lh + s) * (BM + 4) + wm + 32 * i + li];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!B_MMAJ) {
          f32x4 t = *reinterpret_cast<const f32x4*>(Bs + (wn + 32 * j + li) * (BK + 4) + 8 * c + 4 * lh);
          b[j][0] = t[0]; b[j][1] = t[1]; b[j][2] = t[2]; b[j][3] = t[3];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) b[j][s] = Bs[(8 * c + 4 * lh + s) * (BN + 4) + wn + 32 * j + li];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      float* nxt = smem + (cur ^ 1) * (A_ELEMS + B_ELEMS);
      sa.store(nxt);
      sb.store(nxt + A_ELEMS);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: 32x32 accumulator, col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (gridDim.y > 1) {   // split-K: raw partial sums, combined by splitk_reduce_kernel
    float* wsp = g.ws + (size_t)blockIdx.y * g.M * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = bn + wn + 32 * j + li;
        if (col >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = bm + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < g.M) wsp[(size_t)row * g.N + col] = acc[i][j][r];
        }
      }
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

// Combine the split-K partials in split order and apply the epilogue of cg_gemm.
__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmArgs g, int splits) {
  const size_t total = (size_t)g.M * g.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / g.N), n = (int)(i % g.N);
    float h = g.bias ? g.bias[n] : 0.f;
    for (int k = 0; k < splits; ++k) h += g.ws[(size_t)k * total + i];
    float* cp = g.C + (size_t)m * g.ldc + n;
    if (g.epi == CG_EPI_NONE) {
      *cp = g.accumulate ? *cp + h : h;
    } else if (g.epi == CG_EPI_LEAKY) {
      float pm = h > 0.f ? 1.f : g.slope;
      if (g.E) pm *= g.E[(size_t)m * g.lde + n];
      *cp = h * pm;
      if (g.aux) g.aux[(size_t)m * g.ldaux + n] = pm;
    } else {
      *cp = h * g.E[(size_t)m * g.lde + n];
    }
  }
}

template <int BM, int BN, bool HALF = false>
int launch_f32(const GemmArgs& g, int transA, int transB, bool vec, hipStream_t st, int splits = 1) {
  GemmArgs a = g;
  const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
  a.tiles_n = tn;
  const int bk = HALF ? BKH : BK;
  const int nk = (g.K + bk - 1) / bk;
  a.kt_per_split = (nk + splits - 1) / splits;
  splits = (nk + a.kt_per_split - 1) / a.kt_per_split;   // no empty split
  dim3 grid(tm * tn, splits), block(256);
#define CG_LAUNCH(TA, TB)                                                                    \
  do {                                                                                       \
    if (HALF) {                                                                              \
      if (vec) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, TA, TB, true>), grid, block, 0, st, a);   \
      else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, TA, TB, false>), grid, block, 0, st, a);      \
    } else {                                                                                 \
      if (vec) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, true>), grid, block, 0, st, a);    \
      else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB, false>), grid, block, 0, st, a);       \
    }                                                                                        \
  } while (0)
  if (!transA && transB) CG_LAUNCH(false, true);
  else if (!transA && !transB) CG_LAUNCH(false, false);
  else if (transA && !transB) CG_LAUNCH(true, false);
  else CG_LAUNCH(true, true);
#undef CG_LAUNCH
  if (splits > 1) {
    const size_t total = (size_t)g.M * g.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, g, splits);
  }
  return cg_launch_status();
}

}  // namespace

extern "C" int cg_gemm(float* C, int ldc, const void* A, int lda, const void* B, int ldb,
                       int M, int N, int K, int transA, int transB, int dtype,
                       const float* bias, int epi, const float* E, int lde, float* aux, int ldaux,
                       float slope, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (!C || !A || !B || M < 0 || N < 0 || K < 0) return CG_ERR_BAD_ARG;
  if (M == 0 || N == 0) return CG_OK;
  if (epi < CG_EPI_NONE || epi > CG_EPI_MUL) return CG_ERR_BAD_ARG;
  if (epi == CG_EPI_MUL && !E) return CG_ERR_BAD_ARG;
  if (epi != CG_EPI_NONE && accumulate) return CG_ERR_BAD_ARG;
  const int a_rows = transA ? K : M, a_cols = transA ? M : K;
  const int b_rows = transB ? N : K, b_cols = transB ? K : N;
  (void)a_rows; (void)b_rows;
  if (lda < a_cols || ldb < b_cols || ldc < N) return CG_ERR_BAD_ARG;
  if (E && lde < N) return CG_ERR_BAD_ARG;
  if (aux && ldaux < N) return CG_ERR_BAD_ARG;
  if (dtype != CG_F32 && dtype != CG_BF16) return CG_ERR_UNSUPPORTED;
  const bool half = dtype == CG_BF16;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g{(const float*)A, (const float*)B, C, bias, E, aux, M, N, K, lda, ldb, ldc, lde, ldaux,
             epi, accumulate, slope, 0, 1 << 30, (float*)workspace};
  const bool vec = (((uintptr_t)A | (uintptr_t)B) % 16 == 0) && (lda % 4 == 0) && (ldb % 4 == 0);
  // Tile and split-K choice by a cost model fitted to a sweep of the step's shapes on MI355X (microseconds).
  // A workgroup alone on a CU needs L per K slab (latency-bound); n workgroups sharing a CU overlap and finish a
  // slab every S (throughput-bound), so the busiest CU takes slabs * max(L, n*S).  Small tiles have the worse S but
  // fill 256 CUs evenly, which decides every shape of this workload except the very large ones.  Splitting K adds
  // a fixed cost and the round trip of the partial sums.
  const int nk = half ? (K + BKH - 1) / BKH : (K + BK - 1) / BK;
  const size_t per = (size_t)M * N * sizeof(float);
  struct Cand { int bm, bn; double L, S, P; };
  static const Cand cand_f32[3] = {{128, 128, 2.45, 2.21, 6.0}, {128, 64, 1.48, 1.232, 4.0}, {64, 64, 1.0, 0.64, 2.5}};
  static const Cand cand_bf16[3] = {{128, 128, 1.2, 0.65, 6.0}, {128, 64, 0.8, 0.45, 4.0}, {64, 64, 0.9, 0.25, 2.5}};
  const Cand* cands = half ? cand_bf16 : cand_f32;
  static const int split_choices[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
  int best = 2, best_splits = 1;
  double best_cost = 1e300;
  double tile_bias[3] = {1.0, 1.0, 1.0};
  if (const char* f = getenv("CGRNA_GEMM_BIAS")) sscanf(f, "%lf,%lf,%lf", &tile_bias[0], &tile_bias[1], &tile_bias[2]);
  for (int c = 0; c < 3; ++c) {
    const long tiles = (long)((M + cands[c].bm - 1) / cands[c].bm) * ((N + cands[c].bn - 1) / cands[c].bn);
    for (int sp : split_choices) {
      if (sp > 1 && (!workspace || (size_t)sp * per > workspace_bytes || nk / sp < 4)) break;
      const long blocks = tiles * sp;
      const long n = (blocks + 255) / 256;
      const int slabs = (nk + sp - 1) / sp;
      double cost = slabs * (n * cands[c].S > cands[c].L ? n * cands[c].S : cands[c].L) + cands[c].P * (accumulate ? 1.5 : 1.0);
      if (sp > 1) cost += 6.0 + (sp + 1) * (double)per / 4.0e6;
      cost *= tile_bias[c];
      if (cost < best_cost) { best_cost = cost; best = c; best_splits = sp; }
    }
  }
  if (const char* f = getenv("CGRNA_GEMM_FORCE")) {   // tuning aid: "tile,splits" with tile 0=128x128 1=128x64 2=64x64
    int t = 0, sp = 1;
    if (sscanf(f, "%d,%d", &t, &sp) == 2 && t >= 0 && t < 3 && sp >= 1 && (sp == 1 || (workspace && (size_t)sp * per <= workspace_bytes))) {
      best = t;
      best_splits = sp;
    }
  }
  if (half) {
    if (best == 0) return launch_f32<128, 128, true>(g, transA, transB, vec, st, best_splits);
    if (best == 1) return launch_f32<128, 64, true>(g, transA, transB, vec, st, best_splits);
    return launch_f32<64, 64, true>(g, transA, transB, vec, st, best_splits);
  }
  if (best == 0) return launch_f32<128, 128>(g, transA, transB, vec, st, best_splits);
  if (best == 1) return launch_f32<128, 64>(g, transA, transB, vec, st, best_splits);
  return launch_f32<64, 64>(g, transA, transB, vec, st, best_splits);
}
