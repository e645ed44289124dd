// Train-mode BatchNorm1d (forward, backward, backward-of-backward) fused with
// the activation / dropout that follows it.  HBM/L2-bound column kernels.
//
// Geometry: a block owns a strip of 32 columns (one 128-byte line per row) and
// a chunk of rows, walked by 8 row lanes.  Each op is two launches over the
// same (strips x row-chunks) grid:
//   stats : per-block column sums in float64 -> workspace[stat][chunk][col]
//   apply : every block re-adds the (<= 64) chunk partials of its 32 columns in
//           chunk order (deterministic, no atomics), derives the column
//           statistics and applies them to its rows
// so tall-skinny (8192 x 16) and short-wide (512 x 2048) inputs both spread
// over hundreds of workgroups.
#include <mutex>
#include "common.h"

// Scratch of cg_bn_fwd_static: the per-chunk column sums of the two-launch forward, for callers that cannot
// allocate.  One copy per GPU, held until the process ends.  In host code this name is only a handle for
// hipGetSymbolAddress; kernels are handed the device address it returns (static_scratch()).
constexpr size_t BN_STATIC_BYTES = (size_t)1 << 20;
static __device__ __align__(256) unsigned char scratch[BN_STATIC_BYTES];

namespace {

constexpr int CW = 32;   // columns per block
constexpr int RL = 8;    // row lanes per block
constexpr int MAX_RC = 64;

struct Geo {
  int strips, rc, rows;  // column strips, row chunks, rows per chunk
};
static inline Geo geo(int M, int N) {
  Geo g;
  g.strips = (N + CW - 1) / CW;
  int rc = 512 / g.strips;
  if (rc > MAX_RC) rc = MAX_RC;
  if (rc > (M + RL - 1) / RL) rc = (M + RL - 1) / RL;
  if (rc < 1) rc = 1;
  g.rows = ((M + rc - 1) / rc + RL - 1) / RL * RL;
  g.rc = (M + g.rows - 1) / g.rows;
  return g;
}

// Sum `NS` per-thread doubles over the 8 row lanes of each column; result in every lane of the column.
template <int NS>
__device__ __forceinline__ void strip_reduce(double (&s)[NS], double* lds /* NS*RL*CW */) {
  const int c = threadIdx.x & (CW - 1), r = threadIdx.x / CW;
#pragma unroll
  for (int i = 0; i < NS; ++i) lds[(i * RL + r) * CW + c] = s[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    double a = 0.0;
#pragma unroll
    for (int rr = 0; rr < RL; ++rr) a += lds[(i * RL + rr) * CW + c];
    s[i] = a;
  }
  __syncthreads();
}

// Write this block's column sums to the workspace.
template <int NS>
__device__ __forceinline__ void put_partials(double (&s)[NS], double* lds, double* ws, int N, int col) {
  strip_reduce<NS>(s, lds);
  if (threadIdx.x < CW && col < N) {
#pragma unroll
    for (int i = 0; i < NS; ++i) ws[((size_t)i * gridDim.y + blockIdx.y) * N + col] = s[i];
  }
}

// Re-add the chunk partials of this block's columns (row lanes split the chunks).
template <int NS>
__device__ __forceinline__ void get_totals(double (&s)[NS], double* lds, const double* ws, int N, int col) {
  const int r = threadIdx.x / CW;
#pragma unroll
  for (int i = 0; i < NS; ++i) s[i] = 0.0;
  if (col < N)
    for (int ch = r; ch < (int)gridDim.y; ch += RL) {
#pragma unroll
      for (int i = 0; i < NS; ++i) s[i] += ws[((size_t)i * gridDim.y + ch) * N + col];
    }
  strip_reduce<NS>(s, lds);
}

// M is the number of rows of one segment; segment blockIdx.z owns rows [z*M, (z+1)*M) and its own statistics.
#define BN_COMMON                                              \
  const int c = threadIdx.x & (CW - 1), r0 = threadIdx.x / CW; \
  const int col = blockIdx.x * CW + c;                         \
  const bool ok = col < N;                                     \
  const int seg = blockIdx.z;                                  \
  const int m0 = seg * M + blockIdx.y * rows;                  \
  const int m1 = min((seg + 1) * M, m0 + rows);

__device__ __forceinline__ float pm_of(const float* Y, int ldy, const float* PM, int ldpm, float slope, int m, int col) {
  if (PM) return PM[(size_t)m * ldpm + col];
  if (Y) return Y[(size_t)m * ldy + col] > 0.f ? 1.f : slope;
  return 1.f;
}

// ---------------- forward ----------------
__global__ void __launch_bounds__(256) bn_fwd_stats(double* ws, const float* X, int ldx, int M, int N, int rows) {
  __shared__ double lds[2 * RL * CW];
  BN_COMMON
  double s[2] = {0.0, 0.0};
  if (ok)
    for (int m = m0 + r0; m < m1; m += RL) {
      const double x = X[(size_t)m * ldx + col];
      s[0] += x;
      s[1] += x * x;
    }
  put_partials<2>(s, lds, ws + (size_t)seg * 2 * gridDim.y * N, N, col);
}

__global__ void __launch_bounds__(256) bn_fwd_apply(const double* ws, float* Y, int ldy, const float* X, int ldx, int M,
                                                    int N, int rows, const float* gamma, const float* beta, float eps,
                                                    float* mean_out, float* invstd_out, float* rmean, float* rvar,
                                                    float momentum, float slope, const float* mask, int ldmask,
                                                    float* pm_out, int ldpm, u16* ybf, int ldybf, unsigned order) {
  __shared__ double lds[2 * RL * CW];
  BN_COMMON
  const size_t seg_ws = (size_t)2 * gridDim.y * N;
  double s[2];
  if (blockIdx.y == 0 && seg == 0 && (rmean || rvar)) {
    // running statistics: one update per segment, applied in the caller's order (the order in which the
    // reference would have made the separate calls); 4 bits of `order` per position
    for (int k = 0; k < (int)gridDim.z; ++k) {
      const int sg = (order >> (4 * k)) & 15;
      get_totals<2>(s, lds, ws + sg * seg_ws, N, col);
      if (ok && r0 == 0) {
        const double mean_d = s[0] / M;
        double var_d = s[1] / M - mean_d * mean_d;
        if (var_d < 0.0) var_d = 0.0;
        if (rmean) rmean[col] = (1.f - momentum) * rmean[col] + momentum * (float)mean_d;
        if (rvar) {
          const float unb = (float)(var_d * ((double)M / (double)(M > 1 ? M - 1 : 1)));
          rvar[col] = (1.f - momentum) * rvar[col] + momentum * unb;
        }
      }
    }
  }
  get_totals<2>(s, lds, ws + seg * seg_ws, N, col);
  if (!ok) return;
  const double mean_d = s[0] / M;
  double var_d = s[1] / M - mean_d * mean_d;
  if (var_d < 0.0) var_d = 0.0;
  const float mean = (float)mean_d;
  const float invstd = (float)(1.0 / sqrt(var_d + (double)eps));
  if (blockIdx.y == 0 && r0 == 0) {
    mean_out[(size_t)seg * N + col] = mean;
    invstd_out[(size_t)seg * N + col] = invstd;
  }
  const float ga = gamma[col], be = beta[col];
  for (int m = m0 + r0; m < m1; m += RL) {
    const float xh = (X[(size_t)m * ldx + col] - mean) * invstd;
    const float h = xh * ga + be;
    float pm = h > 0.f ? 1.f : slope;
    if (mask) pm *= mask[(size_t)m * ldmask + col];
    const float y = h * pm;
    Y[(size_t)m * ldy + col] = y;
    if (pm_out) pm_out[(size_t)m * ldpm + col] = pm;
    if (ybf) ybf[(size_t)m * ldybf + col] = f2bf(y);
  }
}

__global__ void __launch_bounds__(256) bn_eval_kernel(float* Y, int ldy, const float* X, int ldx, int M, int N,
                                                      const float* gamma, const float* beta, const float* rmean,
                                                      const float* rvar, float eps, float slope, u16* ybf,
                                                      int ldybf) {
  const int c = threadIdx.x & (CW - 1), r0 = threadIdx.x / CW;
  const int col = blockIdx.x * CW + c;
  if (col >= N) return;
  const float mean = rmean[col], invstd = 1.f / sqrtf(rvar[col] + eps), ga = gamma[col], be = beta[col];
  for (int m = blockIdx.y * RL + r0; m < M; m += RL * gridDim.y) {
    const float h = (X[(size_t)m * ldx + col] - mean) * invstd * ga + be;
    const float y = h > 0.f ? h : h * slope;
    Y[(size_t)m * ldy + col] = y;
    if (ybf) ybf[(size_t)m * ldybf + col] = f2bf(y);
  }
}

// One more update of the running statistics with batch statistics that a forward pass has stored (mean, invstd).
__global__ void __launch_bounds__(256) bn_running_update_kernel(float* rmean, float* rvar, const float* mean,
                                                               const float* invstd, int N, int M, float eps,
                                                               float momentum) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  const double is = invstd[col];
  double var_d = 1.0 / (is * is) - (double)eps;
  if (var_d < 0.0) var_d = 0.0;
  rmean[col] = (1.f - momentum) * rmean[col] + momentum * mean[col];
  const float unb = (float)(var_d * ((double)M / (double)(M > 1 ? M - 1 : 1)));
  rvar[col] = (1.f - momentum) * rvar[col] + momentum * unb;
}

// ---------------- backward ----------------
__global__ void __launch_bounds__(256) bn_bwd_stats(double* ws, const float* dY, int lddy, const float* X, int ldx,
                                                    const float* Y, int ldy, const float* PM, int ldpm,
                                                    const float* mean, const float* invstd, float slope, int M, int N,
                                                    int rows) {
  __shared__ double lds[2 * RL * CW];
  BN_COMMON
  const float mu = ok ? mean[(size_t)seg * N + col] : 0.f, is = ok ? invstd[(size_t)seg * N + col] : 0.f;
  double s[2] = {0.0, 0.0};
  if (ok)
    for (int m = m0 + r0; m < m1; m += RL) {
      const float g = dY[(size_t)m * lddy + col] * pm_of(Y, ldy, PM, ldpm, slope, m, col);
      const float xh = (X[(size_t)m * ldx + col] - mu) * is;
      s[0] += g;
      s[1] += (double)g * xh;
    }
  put_partials<2>(s, lds, ws + (size_t)seg * 2 * gridDim.y * N, N, col);
}

__global__ void __launch_bounds__(256) bn_bwd_apply(const double* ws, float* dX, int lddx, const float* dY, int lddy,
                                                    const float* X, int ldx, const float* Y, int ldy, const float* PM,
                                                    int ldpm, const float* mean, const float* invstd,
                                                    const float* gamma, float* dgamma, float* dbeta, int accumulate,
                                                    float* gbar_out, float* gxbar_out, float slope, int M, int N,
                                                    int rows, u16* dxbf, int lddxbf) {
  __shared__ double lds[2 * RL * CW];
  BN_COMMON
  const size_t seg_ws = (size_t)2 * gridDim.y * N;
  double s[2];
  if (blockIdx.y == 0 && seg == 0 && (dbeta || dgamma)) {
    // parameter gradients sum over all segments, added in segment order as separate calls would
    float db = (ok && dbeta && accumulate) ? dbeta[col] : 0.f, dg = (ok && dgamma && accumulate) ? dgamma[col] : 0.f;
    for (int sg = 0; sg < (int)gridDim.z; ++sg) {
      get_totals<2>(s, lds, ws + sg * seg_ws, N, col);
      db += (float)s[0];
      dg += (float)s[1];
    }
    if (ok && r0 == 0) {
      if (dbeta) dbeta[col] = db;
      if (dgamma) dgamma[col] = dg;
    }
  }
  get_totals<2>(s, lds, ws + seg * seg_ws, N, col);
  if (!ok) return;
  const float mu = mean[(size_t)seg * N + col], is = invstd[(size_t)seg * N + col];
  const float gbar = (float)(s[0] / M), gxbar = (float)(s[1] / M);
  if (blockIdx.y == 0 && r0 == 0) {
    if (gbar_out) gbar_out[(size_t)seg * N + col] = gbar;
    if (gxbar_out) gxbar_out[(size_t)seg * N + col] = gxbar;
  }
  const float k = is * gamma[col];
  for (int m = m0 + r0; m < m1; m += RL) {
    const float g = dY[(size_t)m * lddy + col] * pm_of(Y, ldy, PM, ldpm, slope, m, col);
    const float xh = (X[(size_t)m * ldx + col] - mu) * is;
    const float d = k * (g - gbar - xh * gxbar);
    dX[(size_t)m * lddx + col] = d;
    if (dxbf) dxbf[(size_t)m * lddxbf + col] = f2bf(d);
  }
}

// ---------------- one-launch forms for a single segment of a few hundred rows ----------------
// The two-launch forms above spread any shape over hundreds of workgroups; for the dense layers at batch 512 the
// matrix is a few MB and the two launches (plus the gap between them) cost more than the data movement.  Here a
// workgroup owns a strip of CWF columns over ALL rows, keeps its rows in registers (NR per thread, all loads issued
// before the first use), reduces the column sums through LDS in row-lane order and applies them: one launch, one read.
template <int CWF, int NS>
__device__ __forceinline__ void strip_reduce_f(double (&s)[NS], double* lds /* NS*(256/CWF)*CWF */) {
  constexpr int LANES = 256 / CWF;
  const int c = threadIdx.x % CWF, r = threadIdx.x / CWF;
#pragma unroll
  for (int i = 0; i < NS; ++i) lds[(i * LANES + r) * CWF + c] = s[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    double a = 0.0;
#pragma unroll
    for (int rr = 0; rr < LANES; ++rr) a += lds[(i * LANES + rr) * CWF + c];
    s[i] = a;
  }
}

// SLABS: the input is not in X yet but in `nslab` split-K partial sums of the product that makes it (dense M x N
// slabs, one after the other) plus the product's bias; they are added in slab order, as splitk_reduce_kernel of
// gemm.hip adds them, and the sum is also written to X (the backward pass reads it).
template <int CWF, int NR, bool SLABS>
__global__ void __launch_bounds__(256) bn_fwd_fused(float* Y, int ldy, float* X, int ldx, int M, int N,
                                                    const float* gamma, const float* beta, float eps, float* mean_out,
                                                    float* invstd_out, float* rmean, float* rvar, float momentum,
                                                    float slope, const float* mask, int ldmask, float* pm_out, int ldpm,
                                                    u16* ybf, int ldybf, const float* slabs, int nslab,
                                                    const float* xbias) {
  constexpr int LANES = 256 / CWF;
  __shared__ double lds[2 * 256];
  const int c = threadIdx.x % CWF, r0 = threadIdx.x / CWF, col = blockIdx.x * CWF + c;
  const bool ok = col < N;
  // every load is unconditional (row and column clamped into the matrix, the value discarded afterwards): a branch
  // around a load makes the compiler wait for each load before issuing the next
  const int colc = min(col, N - 1);
  float x[NR];
  if (!SLABS) {
#pragma unroll
    for (int i = 0; i < NR; ++i) x[i] = X[(size_t)min(r0 + i * LANES, M - 1) * ldx + colc];
  } else {
    const float bv = xbias ? xbias[colc] : 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) x[i] = bv;
    for (int k = 0; k < nslab; ++k) {
      const float* sl = slabs + (size_t)k * M * N;
#pragma unroll
      for (int i = 0; i < NR; ++i) x[i] += sl[(size_t)min(r0 + i * LANES, M - 1) * N + colc];
    }
  }
  double s[2] = {0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const double xd = (r0 + i * LANES < M) ? x[i] : 0.f;
    s[0] += xd;
    s[1] += xd * xd;
  }
  strip_reduce_f<CWF, 2>(s, lds);
  if (!ok) return;
  const double mean_d = s[0] / M;
  double var_d = s[1] / M - mean_d * mean_d;
  if (var_d < 0.0) var_d = 0.0;
  const float mean = (float)mean_d, invstd = (float)(1.0 / sqrt(var_d + (double)eps));
  if (r0 == 0) {
    mean_out[col] = mean;
    invstd_out[col] = invstd;
    if (rmean) rmean[col] = (1.f - momentum) * rmean[col] + momentum * mean;
    if (rvar) {
      const float unb = (float)(var_d * ((double)M / (double)(M > 1 ? M - 1 : 1)));
      rvar[col] = (1.f - momentum) * rvar[col] + momentum * unb;
    }
  }
  const float ga = gamma[col], be = beta[col];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int m = r0 + i * LANES;
    if (m >= M) break;
    if (SLABS) X[(size_t)m * ldx + col] = x[i];
    const float h = (x[i] - mean) * invstd * ga + be;
    float pm = h > 0.f ? 1.f : slope;
    if (mask) pm *= mask[(size_t)m * ldmask + col];
    const float y = h * pm;
    Y[(size_t)m * ldy + col] = y;
    if (pm_out) pm_out[(size_t)m * ldpm + col] = pm;
    if (ybf) ybf[(size_t)m * ldybf + col] = f2bf(y);
  }
}

// SLABS: the incoming gradient is not in dY yet but in `nslab` split-K partial sums of the product that makes it (dY
// then points at dense M x N slabs, one after the other); they are added in slab order, as splitk_reduce_kernel of
// gemm.hip adds them (a product without bias).
template <int CWF, int NR, bool SLABS>
__global__ void __launch_bounds__(256) bn_bwd_fused(float* dX, int lddx, const float* dY, int lddy, const float* X,
                                                    int ldx, const float* Y, int ldy, const float* PM, int ldpm,
                                                    const float* mean, const float* invstd, const float* gamma,
                                                    float* dgamma, float* dbeta, int accumulate, float* gbar_out,
                                                    float* gxbar_out, float slope, int M, int N, u16* dxbf,
                                                    int lddxbf, int nslab) {
  constexpr int LANES = 256 / CWF;
  __shared__ double lds[2 * 256];
  const int c = threadIdx.x % CWF, r0 = threadIdx.x / CWF, col = blockIdx.x * CWF + c;
  const bool ok = col < N;
  const float mu = ok ? mean[col] : 0.f, is = ok ? invstd[col] : 0.f;
  const int colc = min(col, N - 1);       // unconditional loads, see bn_fwd_fused
  float g[NR], xh[NR];
  if (!SLABS) {
#pragma unroll
    for (int i = 0; i < NR; ++i) g[i] = dY[(size_t)min(r0 + i * LANES, M - 1) * lddy + colc];
  } else {
#pragma unroll
    for (int i = 0; i < NR; ++i) g[i] = 0.f;
    for (int k = 0; k < nslab; ++k) {
      const float* sl = dY + (size_t)k * M * N;
#pragma unroll
      for (int i = 0; i < NR; ++i) g[i] += sl[(size_t)min(r0 + i * LANES, M - 1) * N + colc];
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) xh[i] = X[(size_t)min(r0 + i * LANES, M - 1) * ldx + colc];
  if (PM) {
#pragma unroll
    for (int i = 0; i < NR; ++i) g[i] *= PM[(size_t)min(r0 + i * LANES, M - 1) * ldpm + colc];
  } else if (Y) {
#pragma unroll
    for (int i = 0; i < NR; ++i) g[i] *= Y[(size_t)min(r0 + i * LANES, M - 1) * ldy + colc] > 0.f ? 1.f : slope;
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const bool in = ok && r0 + i * LANES < M;
    g[i] = in ? g[i] : 0.f;
    xh[i] = in ? (xh[i] - mu) * is : 0.f;
  }
  double s[2] = {0.0, 0.0};
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    s[0] += g[i];
    s[1] += (double)g[i] * xh[i];
  }
  strip_reduce_f<CWF, 2>(s, lds);
  if (!ok) return;
  const float gbar = (float)(s[0] / M), gxbar = (float)(s[1] / M);
  if (r0 == 0) {
    if (dbeta) dbeta[col] = (accumulate ? dbeta[col] : 0.f) + (float)s[0];
    if (dgamma) dgamma[col] = (accumulate ? dgamma[col] : 0.f) + (float)s[1];
    if (gbar_out) gbar_out[col] = gbar;
    if (gxbar_out) gxbar_out[col] = gxbar;
  }
  const float k = is * gamma[col];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int m = r0 + i * LANES;
    if (m >= M) break;
    const float d = k * (g[i] - gbar - xh[i] * gxbar);
    dX[(size_t)m * lddx + col] = d;
    if (dxbf) dxbf[(size_t)m * lddxbf + col] = f2bf(d);
  }
}

// ---------------- backward of backward ----------------
__global__ void __launch_bounds__(256) bn_bb_stats(double* ws, const float* W, int ldw, const float* U1, int ldu,
                                                   const float* X, int ldx, const float* mean, const float* invstd,
                                                   int M, int N, int rows) {
  __shared__ double lds[3 * RL * CW];
  BN_COMMON
  const float mu = ok ? mean[col] : 0.f, is = ok ? invstd[col] : 0.f;
  double s[3] = {0.0, 0.0, 0.0};
  if (ok)
    for (int m = m0 + r0; m < m1; m += RL) {
      const float w = W[(size_t)m * ldw + col];
      const float xh = (X[(size_t)m * ldx + col] - mu) * is;
      s[0] += w;
      s[1] += (double)w * xh;
      s[2] += (double)w * U1[(size_t)m * ldu + col];
    }
  put_partials<3>(s, lds, ws, N, col);
}

__global__ void __launch_bounds__(256) bn_bb_apply(const double* ws, float* Vu, int ldvu, float* Vh, int ldvh,
                                                   float* dgamma, int accumulate, const float* W, int ldw,
                                                   const float* U1, int ldu, const float* X, int ldx,
                                                   const float* mean, const float* invstd, const float* gamma,
                                                   const float* ubar, const float* mbar, int M, int N, int rows,
                                                   u16* vhbf, const float* PM, int ldpm) {
  __shared__ double lds[3 * RL * CW];
  BN_COMMON
  double s[3];
  get_totals<3>(s, lds, ws, N, col);
  if (!ok) return;
  const float mu = mean[col], is = invstd[col];
  const float wbar = (float)(s[0] / M), r = (float)(s[1] / M), q = (float)(s[2] / M);
  const float ub = ubar[col], mb = mbar[col], ga = gamma[col];
  if (blockIdx.y == 0 && r0 == 0 && dgamma) {
    const float d = (float)((double)M * is * ((double)q - (double)wbar * ub - (double)r * mb));
    dgamma[col] = (accumulate ? dgamma[col] : 0.f) + d;
  }
  const float k1 = is * ga, k2 = -is * is * ga, k3 = q - wbar * ub - 3.f * mb * r;
  for (int m = m0 + r0; m < m1; m += RL) {
    const float w = W[(size_t)m * ldw + col];
    const float u1 = U1[(size_t)m * ldu + col];
    const float xh = (X[(size_t)m * ldx + col] - mu) * is;
    const float vu = k1 * (w - wbar - xh * r);
    Vu[(size_t)m * ldvu + col] = PM ? vu * PM[(size_t)m * ldpm + col] : vu;
    const float vh = k2 * (mb * (w - wbar) + r * (u1 - ub) + xh * k3);
    Vh[(size_t)m * ldvh + col] = vh;
    if (vhbf) vhbf[(size_t)m * ldvh + col] = f2bf(vh);
  }
}

}  // namespace

extern "C" size_t cg_bn_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  const Geo g = geo(M, N);
  return (size_t)3 * g.rc * N * sizeof(double);
}

extern "C" int cg_bn_fwd_seg(float* Y, int ldy, const float* X, int ldx, int seg_rows, int nseg, unsigned order, int N,
                             const float* gamma, const float* beta, float eps, float* mean_out, float* invstd_out,
                             float* running_mean, float* running_var, float momentum, float slope, const float* mask,
                             int ldmask, float* pm_out, int ldpm, void* ybf16, int ldybf16, void* workspace,
                             size_t workspace_bytes, void* stream) {
  const int M = seg_rows;
  if (!Y || !X || !gamma || !beta || !mean_out || !invstd_out || M <= 0 || N <= 0) return CG_ERR_BAD_ARG;
  if (nseg < 1 || nseg > 8) return CG_ERR_BAD_ARG;
  unsigned seen = 0;
  for (int k = 0; k < nseg; ++k) seen |= 1u << ((order >> (4 * k)) & 15);
  if (seen != (1u << nseg) - 1) return CG_ERR_BAD_ARG;          // `order` must be a permutation of 0..nseg-1
  if (ldy < N || ldx < N || (mask && ldmask < N) || (pm_out && ldpm < N) || (ybf16 && ldybf16 < N)) return CG_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < nseg * cg_bn_workspace_bytes(M, N)) return CG_ERR_WORKSPACE;
  if (nseg == 1 && M <= 512 && N >= 256) {      // one launch (see bn_fwd_fused), 32 rows per thread
#define CG_BN_FWD_FUSED(CWF, NR)                                                                                    \
  hipLaunchKernelGGL((bn_fwd_fused<CWF, NR, false>), dim3((N + CWF - 1) / CWF), dim3(256), 0, (hipStream_t)stream,   \
                     Y, ldy, const_cast<float*>(X), ldx, M, N, gamma, beta, eps, mean_out, invstd_out, running_mean, \
                     running_var, momentum, slope, mask, ldmask, pm_out, ldpm, (u16*)ybf16, ldybf16,                 \
                     (const float*)nullptr, 0, (const float*)nullptr)
    CG_BN_FWD_FUSED(16, 32);      // measured: 7 us at 512 x 512..2048 (strips of 32 columns, 64 rows per thread: 12 us)
#undef CG_BN_FWD_FUSED
    return cg_launch_status();
  }
  const Geo g = geo(M, N);
  dim3 grid(g.strips, g.rc, nseg);
  hipLaunchKernelGGL(bn_fwd_stats, grid, dim3(256), 0, (hipStream_t)stream, (double*)workspace, X, ldx, M, N, g.rows);
  hipLaunchKernelGGL(bn_fwd_apply, grid, dim3(256), 0, (hipStream_t)stream, (const double*)workspace, Y, ldy, X, ldx,
                     M, N, g.rows, gamma, beta, eps, mean_out, invstd_out, running_mean, running_var, momentum, slope,
                     mask, ldmask, pm_out, ldpm, (u16*)ybf16, ldybf16, order);
  return cg_launch_status();
}

extern "C" int cg_bn_fwd(float* Y, int ldy, const float* X, int ldx, int M, int N, const float* gamma,
                         const float* beta, float eps, float* mean_out, float* invstd_out, float* running_mean,
                         float* running_var, float momentum, float slope, const float* mask, int ldmask,
                         float* pm_out, int ldpm, void* ybf16, int ldybf16, void* workspace, size_t workspace_bytes,
                         void* stream) {
  return cg_bn_fwd_seg(Y, ldy, X, ldx, M, 1, 0u, N, gamma, beta, eps, mean_out, invstd_out, running_mean, running_var,
                       momentum, slope, mask, ldmask, pm_out, ldpm, ybf16, ldybf16, workspace, workspace_bytes, stream);
}

extern "C" int cg_bn_fwd_slabs_ok(int M, int N) { return M > 0 && M <= 512 && N >= 256; }

extern "C" int cg_bn_fwd_slabs(float* Y, int ldy, float* X, int ldx, const float* slabs, int n_slabs,
                               const float* x_bias, int M, int N, const float* gamma, const float* beta, float eps,
                               float* mean_out, float* invstd_out, float* running_mean, float* running_var,
                               float momentum, float slope, const float* mask, int ldmask, float* pm_out, int ldpm,
                               void* ybf16, int ldybf16, void* stream) {
  if (!Y || !X || !slabs || !gamma || !beta || !mean_out || !invstd_out || M <= 0 || N <= 0 || n_slabs < 1)
    return CG_ERR_BAD_ARG;
  if (ldy < N || ldx < N || (mask && ldmask < N) || (pm_out && ldpm < N) || (ybf16 && ldybf16 < N)) return CG_ERR_BAD_ARG;
  if (!cg_bn_fwd_slabs_ok(M, N)) return CG_ERR_UNSUPPORTED;      // only the one-launch form reads slabs
  hipLaunchKernelGGL((bn_fwd_fused<16, 32, true>), dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, Y, ldy, X,
                     ldx, M, N, gamma, beta, eps, mean_out, invstd_out, running_mean, running_var, momentum, slope, mask,
                     ldmask, pm_out, ldpm, (u16*)ybf16, ldybf16, slabs, n_slabs, x_bias);
  return cg_launch_status();
}

// Device address of `scratch` on the current device, looked up once per device.
static int static_scratch(void** out) {
  constexpr int MAX_DEV = 64;
  static void* ptrs[MAX_DEV] = {nullptr};
  static std::mutex mu;
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= MAX_DEV) return CG_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(mu);
  if (!ptrs[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (!p) return CG_ERR_UNSUPPORTED;
    ptrs[dev] = p;
  }
  *out = ptrs[dev];
  return CG_OK;
}

extern "C" int cg_bn_fwd_static(float* Y, int ldy, const float* X, int ldx, int M, int N, const float* gamma,
                                const float* beta, float eps, float* mean_out, float* invstd_out, float* running_mean,
                                float* running_var, float momentum, float slope, const float* mask, int ldmask,
                                float* pm_out, int ldpm, void* ybf16, int ldybf16, void* stream) {
  if (M <= 0 || N <= 0) return CG_ERR_BAD_ARG;
  if (cg_bn_workspace_bytes(M, N) > BN_STATIC_BYTES) return CG_ERR_WORKSPACE;   // more columns than the array holds sums for
  void* ws = nullptr;
  const int st = static_scratch(&ws);
  if (st != CG_OK) return st;
  // every partial sum the second launch reads is written by the first, so nothing in the array has to start at zero
  return cg_bn_fwd_seg(Y, ldy, X, ldx, M, 1, 0u, N, gamma, beta, eps, mean_out, invstd_out, running_mean, running_var,
                       momentum, slope, mask, ldmask, pm_out, ldpm, ybf16, ldybf16, ws, BN_STATIC_BYTES, stream);
}

extern "C" int cg_bn_running_update(float* running_mean, float* running_var, const float* mean, const float* invstd,
                                    int M, int N, float eps, float momentum, void* stream) {
  if (!running_mean || !running_var || !mean || !invstd || M <= 0 || N <= 0) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, running_mean,
                     running_var, mean, invstd, N, M, eps, momentum);
  return cg_launch_status();
}

extern "C" int cg_bn_eval(float* Y, int ldy, const float* X, int ldx, int M, int N, const float* gamma,
                          const float* beta, const float* running_mean, const float* running_var, float eps,
                          float slope, void* ybf16, int ldybf16, void* stream) {
  if (!Y || !X || !gamma || !beta || !running_mean || !running_var || M <= 0 || N <= 0) return CG_ERR_BAD_ARG;
  if (ldy < N || ldx < N || (ybf16 && ldybf16 < N)) return CG_ERR_BAD_ARG;
  int gy = (M + RL * 16 - 1) / (RL * 16);
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(bn_eval_kernel, dim3((N + CW - 1) / CW, gy), dim3(256), 0, (hipStream_t)stream, Y, ldy, X, ldx,
                     M, N, gamma, beta, running_mean, running_var, eps, slope, (u16*)ybf16, ldybf16);
  return cg_launch_status();
}

extern "C" int cg_bn_bwd(float* dX, int lddx, const float* dY, int lddy, const float* X, int ldx, const float* Y,
                         int ldy, const float* PM, int ldpm, const float* mean, const float* invstd,
                         const float* gamma, float* dgamma, float* dbeta, int accumulate, float* gbar_out,
                         float* gxbar_out, float slope, int M, int N, void* dxbf16, int lddxbf16, void* workspace,
                         size_t workspace_bytes, void* stream) {
  return cg_bn_bwd_seg(dX, lddx, dY, lddy, X, ldx, Y, ldy, PM, ldpm, mean, invstd, gamma, dgamma, dbeta, accumulate,
                       gbar_out, gxbar_out, slope, M, 1, N, dxbf16, lddxbf16, workspace, workspace_bytes, stream);
}

extern "C" int cg_bn_bwd_seg(float* dX, int lddx, const float* dY, int lddy, const float* X, int ldx, const float* Y,
                             int ldy, const float* PM, int ldpm, const float* mean, const float* invstd,
                             const float* gamma, float* dgamma, float* dbeta, int accumulate, float* gbar_out,
                             float* gxbar_out, float slope, int seg_rows, int nseg, int N, void* dxbf16,
                             int lddxbf16, void* workspace, size_t workspace_bytes, void* stream) {
  const int M = seg_rows;
  if (!dX || !dY || !X || !mean || !invstd || !gamma || M <= 0 || N <= 0 || nseg < 1 || nseg > 8) return CG_ERR_BAD_ARG;
  if (lddx < N || lddy < N || ldx < N || (Y && ldy < N) || (PM && ldpm < N) || (dxbf16 && lddxbf16 < N)) return CG_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < nseg * cg_bn_workspace_bytes(M, N)) return CG_ERR_WORKSPACE;
  if (nseg == 1 && M <= 512 && N >= 256) {      // one launch (see bn_bwd_fused)
    hipLaunchKernelGGL((bn_bwd_fused<16, 32, false>), dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, dX, lddx,
                       dY, lddy, X, ldx, Y, ldy, PM, ldpm, mean, invstd, gamma, dgamma, dbeta, accumulate, gbar_out,
                       gxbar_out, slope, M, N, (u16*)dxbf16, lddxbf16, 0);
    return cg_launch_status();
  }
  const Geo g = geo(M, N);
  dim3 grid(g.strips, g.rc, nseg);
  hipLaunchKernelGGL(bn_bwd_stats, grid, dim3(256), 0, (hipStream_t)stream, (double*)workspace, dY, lddy, X, ldx, Y,
                     ldy, PM, ldpm, mean, invstd, slope, M, N, g.rows);
  hipLaunchKernelGGL(bn_bwd_apply, grid, dim3(256), 0, (hipStream_t)stream, (const double*)workspace, dX, lddx, dY,
                     lddy, X, ldx, Y, ldy, PM, ldpm, mean, invstd, gamma, dgamma, dbeta, accumulate, gbar_out,
                     gxbar_out, slope, M, N, g.rows, (u16*)dxbf16, lddxbf16);
  return cg_launch_status();
}

extern "C" int cg_bn_bwd_slabs_ok(int M, int N) { return M > 0 && M <= 512 && N >= 256; }

extern "C" int cg_bn_bwd_slabs(float* dX, int lddx, const float* slabs, int n_slabs, const float* X, int ldx,
                               const float* Y, int ldy, const float* PM, int ldpm, const float* mean,
                               const float* invstd, const float* gamma, float* dgamma, float* dbeta, int accumulate,
                               float slope, int M, int N, void* dxbf16, int lddxbf16, void* stream) {
  if (!dX || !slabs || !X || !mean || !invstd || !gamma || M <= 0 || N <= 0 || n_slabs < 1) return CG_ERR_BAD_ARG;
  if (lddx < N || ldx < N || (Y && ldy < N) || (PM && ldpm < N) || (dxbf16 && lddxbf16 < N)) return CG_ERR_BAD_ARG;
  if (!cg_bn_bwd_slabs_ok(M, N)) return CG_ERR_UNSUPPORTED;      // only the one-launch form reads slabs
  hipLaunchKernelGGL((bn_bwd_fused<16, 32, true>), dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, dX, lddx,
                     slabs, N, X, ldx, Y, ldy, PM, ldpm, mean, invstd, gamma, dgamma, dbeta, accumulate,
                     (float*)nullptr, (float*)nullptr, slope, M, N, (u16*)dxbf16, lddxbf16, n_slabs);
  return cg_launch_status();
}

extern "C" int cg_bn_bwd_bwd(float* V_u1, int ldvu, float* V_h0, int ldvh, float* dgamma, int accumulate,
                             const float* W, int ldw, const float* U1, int ldu, const float* X, int ldx,
                             const float* mean, const float* invstd, const float* gamma, const float* ubar,
                             const float* mbar, int M, int N, void* vh_bf16, const float* PM, int ldpm, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (PM && ldpm < N) return CG_ERR_BAD_ARG;
  if (!V_u1 || !V_h0 || !W || !U1 || !X || !mean || !invstd || !gamma || !ubar || !mbar || M <= 0 || N <= 0)
    return CG_ERR_BAD_ARG;
  if (ldvu < N || ldvh < N || ldw < N || ldu < N || ldx < N) return CG_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < cg_bn_workspace_bytes(M, N)) return CG_ERR_WORKSPACE;
  const Geo g = geo(M, N);
  dim3 grid(g.strips, g.rc);
  hipLaunchKernelGGL(bn_bb_stats, grid, dim3(256), 0, (hipStream_t)stream, (double*)workspace, W, ldw, U1, ldu, X, ldx,
                     mean, invstd, M, N, g.rows);
  hipLaunchKernelGGL(bn_bb_apply, grid, dim3(256), 0, (hipStream_t)stream, (const double*)workspace, V_u1, ldvu, V_h0,
                     ldvh, dgamma, accumulate, W, ldw, U1, ldu, X, ldx, mean, invstd, gamma, ubar, mbar, M, N, g.rows,
                     (u16*)vh_bf16, PM, ldpm);
  return cg_launch_status();
}
