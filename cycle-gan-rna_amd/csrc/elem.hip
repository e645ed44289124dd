// Elementwise, gather/scatter and reduction kernels of the training step.
// All are HBM/L2-bound; threads run along the contiguous (column) dimension,
// grid.y strides over rows.  Reductions write one float partial per block
// (at most CG_MAX_PARTS) and the consumer adds them in index order in float64,
// so every loss is reproducible run to run (no float atomics).
#include "common.h"

namespace {

struct Grid2 {
  dim3 grid, block;
};
static inline Grid2 grid2(int M, int N, int max_blocks = 2048) {
  int gx = (N + 255) / 256;
  int gy = M;
  if (gy > max_blocks / gx) gy = max_blocks / gx;
  if (gy < 1) gy = 1;
  return {dim3(gx, gy), dim3(256)};
}
static inline Grid2 grid_red(int M, int N) { return grid2(M, N, CG_MAX_PARTS); }

#define CG_2D_LOOP(M, N)                                  \
  const int col = blockIdx.x * 256 + threadIdx.x;         \
  if (col < (N))                                          \
    for (int m = blockIdx.y; m < (M); m += gridDim.y)

__global__ void axpby_kernel(float* out, int ldo, float a, const float* x, int ldx, float b, const float* y, int ldy,
                             int M, int N) {
  CG_2D_LOOP(M, N) {
    float v = a * x[(size_t)m * ldx + col];
    if (y) v += b * y[(size_t)m * ldy + col];
    out[(size_t)m * ldo + col] = v;
  }
}

__global__ void mul_kernel(float* out, int ldo, const float* x, int ldx, const float* y, int ldy, int M, int N) {
  CG_2D_LOOP(M, N) out[(size_t)m * ldo + col] = x[(size_t)m * ldx + col] * y[(size_t)m * ldy + col];
}

__global__ void fill_coef_kernel(float* out, int ldo, int M, int N, const float* coef, int idx, float scale) {
  const float v = coef[idx] * scale;
  CG_2D_LOOP(M, N) out[(size_t)m * ldo + col] = v;
}

__global__ void to_bf16_kernel(u16* out, int ldo, const float* x, int ldx, int M, int N) {
  CG_2D_LOOP(M, N) out[(size_t)m * ldo + col] = f2bf(x[(size_t)m * ldx + col]);
}

// out <- x and out16 <- bf16(x), both with leading dimension ldo
__global__ void copy_bf16_kernel(float* out, u16* out16, int ldo, const float* x, int ldx, int M, int N) {
  CG_2D_LOOP(M, N) {
    const float v = x[(size_t)m * ldx + col];
    out[(size_t)m * ldo + col] = v;
    out16[(size_t)m * ldo + col] = f2bf(v);
  }
}

__global__ void embed_gather_kernel(float* out, int ldo, const float* table, int E, const int64_t* labels,
                                    int lstride, int M, int n_types, u16* out16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * E) return;
  const int m = i / E, e = i % E;
  int64_t t = labels[(size_t)m * lstride];
  const bool none = t < 0;      // a negative label marks a row that belongs to no cell: zeros
  t = t < 0 ? 0 : (t >= n_types ? n_types - 1 : t);
  const float v = none ? 0.f : table[t * E + e];
  out[(size_t)m * ldo + e] = v;
  if (out16) out16[(size_t)m * ldo + e] = f2bf(v);
}

// One block per table row (cell type): 64 row lanes x 16 column lanes walk the batch, then the row lanes are
// combined through LDS in a fixed order (deterministic; embedding widths up to 128).  Loads are unconditional
// (columns clamped, rows of other types discarded) and four rows are in flight per thread.
__global__ void __launch_bounds__(1024) embed_scatter_kernel(float* dtable, int E, const float* dout, int lddo,
                                                            const int64_t* labels, int lstride, int M, int n_types,
                                                            int accumulate) {
  __shared__ float lds[64][128];
  const int t = blockIdx.x;
  const int el = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int nu = (E + 15) >> 4;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int m0 = rl; m0 < M; m0 += 256) {
    bool hit[4];
    float v[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = min(m0 + 64 * r, M - 1);
      hit[r] = m0 + 64 * r < M && labels[(size_t)m * lstride] == t;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < nu) v[r][u] = dout[(size_t)m * lddo + min(el + 16 * u, E - 1)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (u < nu) acc[u] += (hit[r] && el + 16 * u < E) ? v[r][u] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) lds[rl][el + 16 * u] = acc[u];
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 1024) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 64; ++r) s += lds[r][e];
    dtable[t * E + e] = (accumulate ? dtable[t * E + e] : 0.f) + s;
  }
}

__global__ void reparam_fwd_kernel(float* z, int ldz, const float* ml, int ldml, const float* eps, int M, int L) {
  CG_2D_LOOP(M, L) {
    const float mu = ml[(size_t)m * ldml + col], lv = ml[(size_t)m * ldml + L + col];
    z[(size_t)m * ldz + col] = mu + eps[(size_t)m * L + col] * expf(0.5f * lv);
  }
}

__global__ void reparam_bwd_kernel(float* dml, int ldd, const float* dz, int lddz, const float* ml, int ldml,
                                   const float* eps, const float* coef, int kl_idx, float kl_weight, int M, int L) {
  // kl = -0.5*mean(1 + lv - mu^2 - exp(lv)): d/dmu = mu/n, d/dlv = -0.5*(1-exp(lv))/n, n = M*L
  const float k = coef ? coef[kl_idx] * kl_weight / ((float)M * (float)L) : 0.f;
  CG_2D_LOOP(M, L) {
    const float mu = ml[(size_t)m * ldml + col], lv = ml[(size_t)m * ldml + L + col];
    const float g = dz[(size_t)m * lddz + col];
    const float std = expf(0.5f * lv);
    dml[(size_t)m * ldd + col] = g + k * mu;
    dml[(size_t)m * ldd + L + col] = g * eps[(size_t)m * L + col] * 0.5f * std + k * (-0.5f) * (1.f - expf(lv));
  }
}

// One block per row.
__global__ void __launch_bounds__(256) libsize_fwd_kernel(float* Y, int ldy, const float* X, int ldx, float* rowsum,
                                                          float lib, int N) {
  __shared__ float lds[17];
  const int m = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < N; c += 256) s += fmaxf(X[(size_t)m * ldx + c], 0.f);
  s = block_sum<float>(s, lds, true);
  if (threadIdx.x == 0) rowsum[m] = s;
  const float k = lib / (s + 1e-10f);
  for (int c = threadIdx.x; c < N; c += 256) Y[(size_t)m * ldy + c] = fmaxf(X[(size_t)m * ldx + c], 0.f) * k;
}

__global__ void __launch_bounds__(256) libsize_bwd_kernel(float* dX, int lddx, const float* dY, int lddy,
                                                          const float* X, int ldx, const float* rowsum, float lib,
                                                          int N) {
  __shared__ float lds[17];
  const int m = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < N; c += 256) s += dY[(size_t)m * lddy + c] * fmaxf(X[(size_t)m * ldx + c], 0.f);
  s = block_sum<float>(s, lds, true);
  const float den = rowsum[m] + 1e-10f;
  const float k = lib / den, t = s / den;
  for (int c = threadIdx.x; c < N; c += 256) {
    const float x = X[(size_t)m * ldx + c];
    dX[(size_t)m * lddx + c] = x > 0.f ? k * (dY[(size_t)m * lddy + c] - t) : 0.f;
  }
}

__global__ void gp_interp_kernel(float* out, int ldo, const float* real, int ldr, const float* fake, int ldf,
                                 const float* alpha, int M, int G) {
  CG_2D_LOOP(M, G) {
    const float a = alpha[m];
    out[(size_t)m * ldo + col] = a * real[(size_t)m * ldr + col] + (1.f - a) * fake[(size_t)m * ldf + col];
  }
}

// The critic step's three input batches in one pass: rows [0, M) of `out` <- real, [M, 2M) <- fake, [2M, 3M) <- the
// interpolates (row block ib instead of 2 on request); and their labels (n_cov int64 per row): real's, fake's, -1 for
// row blocks in between, fake's.
__global__ void critic_inputs_kernel(float* out, int ldo, const float* real, int ldr, const float* fake, int ldf,
                                     const float* alpha, int M, int G, int64_t* lab_out, const int64_t* real_lab,
                                     const int64_t* fake_lab, int n_cov, u16* out16, int ib) {
  CG_2D_LOOP(M, G) {
    const float a = alpha[m], r = real[(size_t)m * ldr + col], f = fake[(size_t)m * ldf + col];
    const float x = a * r + (1.f - a) * f;
    out[(size_t)m * ldo + col] = r;
    out[(size_t)(M + m) * ldo + col] = f;
    out[((size_t)ib * M + m) * ldo + col] = x;
    if (out16) {
      out16[(size_t)m * ldo + col] = f2bf(r);
      out16[(size_t)(M + m) * ldo + col] = f2bf(f);
      out16[((size_t)ib * M + m) * ldo + col] = f2bf(x);
    }
  }
  if (lab_out && blockIdx.x == 0 && blockIdx.y == 0) {
    const int t = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
    for (int i = t; i < M * n_cov; i += nt) {
      lab_out[i] = real_lab[i];
      lab_out[M * n_cov + i] = fake_lab[i];
      for (int k = 2; k < ib; ++k) lab_out[(size_t)k * M * n_cov + i] = -1;      // row blocks that hold no cells
      lab_out[(size_t)ib * M * n_cov + i] = fake_lab[i];
    }
  }
}

// First hidden layer of the critic on the interpolated batch without a product: the layer is linear, so with
// x_i = a x_r + (1 - a) x_f and the generated batch's embedding on the interpolates,
//   h_i = a h_r + (1 - a) h_f + a (emb_f - emb_r) We^T        (bias included in h_r, h_f and h_i)
// H holds h_r in rows [0, M), h_f in [M, 2M) and receives h_i in [2M, 3M); We is stored [N][E].
__global__ void interp_hidden_kernel(float* H, int ldh, const float* alpha, const float* emb, int lde, int E,
                                     const float* We, int ldw, int M, int N) {
  CG_2D_LOOP(M, N) {
    const float a = alpha[m];
    float v = a * H[(size_t)m * ldh + col] + (1.f - a) * H[(size_t)(M + m) * ldh + col];
    if (emb) {
      float d = 0.f;
      for (int j = 0; j < E; ++j)
        d += (emb[(size_t)(M + m) * lde + j] - emb[(size_t)m * lde + j]) * We[(size_t)col * ldw + j];
      v += a * d;
    }
    H[(size_t)(2 * M + m) * ldh + col] = v;
  }
}

// The same with labels, tiled: a workgroup owns 64 columns x 32 rows and first puts its 64 rows of We and its 32
// embedding differences into LDS - We's rows are 48 bytes at a stride of a whole first-layer row, and every thread of
// a column used to fetch them again - so the dot products read LDS (row stride 33 floats: conflict-free).
constexpr int IH_C = 64, IH_R = 32, IH_EMAX = 32;
__global__ void __launch_bounds__(256) interp_hidden_tiled_kernel(float* H, int ldh, const float* alpha,
                                                                  const float* emb, int lde, int E, const float* We,
                                                                  int ldw, int M, int N) {
  __shared__ float we[IH_C][IH_EMAX + 1], de[IH_R][IH_EMAX + 1], al[IH_R];
  const int c0 = blockIdx.x * IH_C, r0 = blockIdx.y * IH_R;
  for (int i = threadIdx.x; i < IH_C * E; i += 256) {
    const int c = i / E, j = i - c * E;
    we[c][j] = c0 + c < N ? We[(size_t)(c0 + c) * ldw + j] : 0.f;
  }
  for (int i = threadIdx.x; i < IH_R * E; i += 256) {
    const int r = i / E, j = i - r * E, m = r0 + r;
    de[r][j] = m < M ? emb[(size_t)(M + m) * lde + j] - emb[(size_t)m * lde + j] : 0.f;
  }
  if (threadIdx.x < IH_R) al[threadIdx.x] = r0 + (int)threadIdx.x < M ? alpha[r0 + threadIdx.x] : 0.f;
  __syncthreads();
  const int c = threadIdx.x & 63, col = c0 + c;
  if (col >= N) return;
  for (int r = threadIdx.x >> 6; r < IH_R; r += 4) {
    const int m = r0 + r;
    if (m >= M) break;
    const float a = al[r];
    float v = a * H[(size_t)m * ldh + col] + (1.f - a) * H[(size_t)(M + m) * ldh + col];
    float d = 0.f;
    for (int j = 0; j < E; ++j) d += de[r][j] * we[c][j];
    v += a * d;
    H[(size_t)(2 * M + m) * ldh + col] = v;
  }
}

// The penalty's weight-gradient term vh0^T x_interp folded into the Wasserstein terms (see cg_gp_combine): in place,
// D[m] += a_m VH[m] (real rows), D[M + m] += (1 - a_m) VH[m] (generated rows); AV[m] = a_m VH[m]; and, by the first
// column block, dE[m] = emb[M + m] - emb[m].
__global__ void gp_combine_kernel(float* D, int ldd, const float* VH, int ldv, const float* alpha, float* AV, int ldav,
                                  const float* emb, int lde, int E, float* dE, int ldde, int M, int N, u16* D16) {
  CG_2D_LOOP(M, N) {
    const float a = alpha[m], v = VH[(size_t)m * ldv + col];
    const float av = a * v;
    const float p = D[(size_t)m * ldd + col] + av;
    const float q = D[(size_t)(M + m) * ldd + col] + (v - av);
    D[(size_t)m * ldd + col] = p;
    D[(size_t)(M + m) * ldd + col] = q;
    AV[(size_t)m * ldav + col] = av;
    if (D16) {
      D16[(size_t)m * ldd + col] = f2bf(p);
      D16[(size_t)(M + m) * ldd + col] = f2bf(q);
    }
  }
  if (emb && blockIdx.x == 0 && (int)threadIdx.x < E)
    for (int m = blockIdx.y; m < M; m += gridDim.y)
      dE[(size_t)m * ldde + threadIdx.x] = emb[(size_t)(M + m) * lde + threadIdx.x] - emb[(size_t)m * lde + threadIdx.x];
}

// nparts (may be nullptr): rowpen is the partial-sum array of a reduction (see cg_reduce_*), whose count is set here.
__global__ void __launch_bounds__(256) gp_penalty_kernel(float* V, int ldv, float* rowpen, const float* UX, int ldu,
                                                         float lam2_over_m, int G, u16* V16, int* nparts) {
  __shared__ float lds[17];
  const int m = blockIdx.x;
  if (nparts && m == 0 && threadIdx.x == 0) *nparts = gridDim.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < G; c += 256) {
    const float u = UX[(size_t)m * ldu + c];
    s += u * u;
  }
  s = block_sum<float>(s, lds, true);
  const float n = sqrtf(s);
  if (threadIdx.x == 0) rowpen[m] = (n - 1.f) * (n - 1.f);
  // d/du of (n-1)^2 is 2 (n-1) u / n; torch's norm backward yields 0 at n = 0
  const float k = n > 0.f ? lam2_over_m * (n - 1.f) / n : 0.f;
  for (int c = threadIdx.x; c < G; c += 256) {
    const float v = k * UX[(size_t)m * ldu + c];
    V[(size_t)m * ldv + c] = v;
    if (V16) V16[(size_t)m * ldv + c] = f2bf(v);
  }
}

// The same for rows of at most 256 * NV values: the row stays in registers between the norm and the scaling (one
// read of UX instead of two; the squares are added in the same order).
template <int NV>
__global__ void __launch_bounds__(256) gp_penalty_regs_kernel(float* V, int ldv, float* rowpen, const float* UX, int ldu,
                                                              float lam2_over_m, int G, u16* V16, int* nparts) {
  __shared__ float lds[17];
  const int m = blockIdx.x;
  if (nparts && m == 0 && threadIdx.x == 0) *nparts = gridDim.x;
  float u[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = threadIdx.x + 256 * i;
    u[i] = UX[(size_t)m * ldu + min(c, G - 1)];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((int)threadIdx.x + 256 * i < G) s += u[i] * u[i];
  s = block_sum<float>(s, lds, true);
  const float n = sqrtf(s);
  if (threadIdx.x == 0) rowpen[m] = (n - 1.f) * (n - 1.f);
  const float k = n > 0.f ? lam2_over_m * (n - 1.f) / n : 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < G) {
      const float v = k * u[i];
      V[(size_t)m * ldv + c] = v;
      if (V16) V16[(size_t)m * ldv + c] = f2bf(v);
    }
  }
}

// ---- reductions ----------------------------------------------------------- //
__device__ __forceinline__ void write_part(float v, float* parts, int* nparts) {
  __shared__ float lds[17];
  v = block_sum<float>(v, lds);
  if (threadIdx.x == 0) {
    parts[blockIdx.y * gridDim.x + blockIdx.x] = v;
    if (blockIdx.x == 0 && blockIdx.y == 0) *nparts = gridDim.x * gridDim.y;
  }
}

__global__ void __launch_bounds__(256) reduce_sum_kernel(float* parts, int* nparts, const float* X, int ldx, int M,
                                                         int N) {
  float s = 0.f;
  CG_2D_LOOP(M, N) s += X[(size_t)m * ldx + col];
  write_part(s, parts, nparts);
}

// nseg equal row segments of X, segment k into slot k of consecutive reductions (parts + k * CG_MAX_PARTS, nparts + k):
// the partial sums of nseg calls of reduce_sum_kernel in one launch.
__global__ void __launch_bounds__(256) reduce_sum_seg_kernel(float* parts, int* nparts, const float* X, int ldx,
                                                             int seg_rows, int N) {
  const float* Xs = X + (size_t)blockIdx.z * seg_rows * ldx;
  float s = 0.f;
  CG_2D_LOOP(seg_rows, N) s += Xs[(size_t)m * ldx + col];
  write_part(s, parts + (size_t)blockIdx.z * CG_MAX_PARTS, nparts + blockIdx.z);
}

__global__ void __launch_bounds__(256) reduce_l1_kernel(float* parts, int* nparts, const float* A, int lda,
                                                        const float* B, int ldb, int M, int N) {
  float s = 0.f;
  CG_2D_LOOP(M, N) s += fabsf(A[(size_t)m * lda + col] - B[(size_t)m * ldb + col]);
  write_part(s, parts, nparts);
}

// Poisson negative log-likelihood up to the target-only term, poisson_loss (GAN_functions.py:216-217):
// sum of pred - target * log(pred + 1e-10)
__global__ void __launch_bounds__(256) reduce_poisson_kernel(float* parts, int* nparts, const float* A, int lda,
                                                             const float* B, int ldb, int M, int N) {
  float s = 0.f;
  CG_2D_LOOP(M, N) {
    const float pr = A[(size_t)m * lda + col];
    s += pr - B[(size_t)m * ldb + col] * logf(pr + 1e-10f);
  }
  write_part(s, parts, nparts);
}

__device__ __forceinline__ int64_t clamp_label(int64_t t, int n_types) {
  return t < 0 ? 0 : (t >= n_types ? n_types - 1 : t);
}

__global__ void __launch_bounds__(256) reduce_centroid_kernel(float* parts, int* nparts, const float* A, int lda,
                                                              const float* Cn, int ldc, const int64_t* labels,
                                                              int lstride, int M, int N, int n_types) {
  float s = 0.f;
  CG_2D_LOOP(M, N) {
    const float d = A[(size_t)m * lda + col] - Cn[(size_t)clamp_label(labels[(size_t)m * lstride], n_types) * ldc + col];
    s += d * d;
  }
  write_part(s, parts, nparts);
}

__global__ void __launch_bounds__(256) reduce_sqdiff_kernel(float* parts, int* nparts, const float* A, int lda,
                                                            const float* B, int ldb, int M, int N) {
  float s = 0.f;
  CG_2D_LOOP(M, N) {
    const float d = A[(size_t)m * lda + col] - B[(size_t)m * ldb + col];
    s += d * d;
  }
  write_part(s, parts, nparts);
}

__global__ void __launch_bounds__(256) reduce_kl_kernel(float* parts, int* nparts, const float* ml, int ldml, int M,
                                                        int L) {
  float s = 0.f;
  CG_2D_LOOP(M, L) {
    const float mu = ml[(size_t)m * ldml + col], lv = ml[(size_t)m * ldml + L + col];
    s += 1.f + lv - mu * mu - expf(lv);
  }
  write_part(s, parts, nparts);
}

// Column statistics with the 32-column strip layout of bn.hip.
__global__ void __launch_bounds__(1024) colsum_kernel(float* out, const float* X, int ldx, int M, int N, float scale,
                                                      int accumulate) {
  __shared__ double lds[32 * 32];
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  double s = 0.0;
  if (col < N)
    for (int m = r0; m < M; m += 32) s += X[(size_t)m * ldx + col];
  lds[r0 * 32 + c] = s;
  __syncthreads();
  if (r0 == 0 && col < N) {
    double a = 0.0;
    for (int r = 0; r < 32; ++r) a += lds[r * 32 + c];
    out[col] = (accumulate ? out[col] : 0.f) + (float)(a * (double)scale);
  }
}

// Up to three column sums in one launch (the bias gradients at the end of a backward chain): the strips of the jobs
// one after the other, each summed exactly as colsum_kernel does.
struct ColsumJob {
  float* out; const float* X; int ldx, M, N; float scale; int accumulate;
};
struct ColsumJobs {
  ColsumJob j[3];
  int first[4];     // first strip of each job; first[3] = number of strips
};
__global__ void __launch_bounds__(1024) colsum_multi_kernel(ColsumJobs js) {
  __shared__ double lds[32 * 32];
  const int k = (int)blockIdx.x >= js.first[2] ? 2 : ((int)blockIdx.x >= js.first[1] ? 1 : 0);
  const ColsumJob& jb = js.j[k];
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  const int col = ((int)blockIdx.x - js.first[k]) * 32 + c;
  double s = 0.0;
  if (col < jb.N)
    for (int m = r0; m < jb.M; m += 32) s += jb.X[(size_t)m * jb.ldx + col];
  lds[r0 * 32 + c] = s;
  __syncthreads();
  if (r0 == 0 && col < jb.N) {
    double a = 0.0;
    for (int r = 0; r < 32; ++r) a += lds[r * 32 + c];
    jb.out[col] = (jb.accumulate ? jb.out[col] : 0.f) + (float)(a * (double)jb.scale);
  }
}

// Per-strip sum of unbiased column variances -> parts; finalised by colvar_final.
__global__ void __launch_bounds__(256) colvar_kernel(double* parts, const float* X, int ldx, int M, int N) {
  __shared__ double lds[2 * 8 * 32];
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  double s = 0.0, q = 0.0;
  if (col < N)
    for (int m = r0; m < M; m += 8) {
      const double x = X[(size_t)m * ldx + col];
      s += x;
      q += x * x;
    }
  lds[r0 * 32 + c] = s;
  lds[256 + r0 * 32 + c] = q;
  __syncthreads();
  if (threadIdx.x < 32) {
    double var = 0.0;
    if (col < N) {
      double a = 0.0, b = 0.0;
      for (int r = 0; r < 8; ++r) {
        a += lds[r * 32 + c];
        b += lds[256 + r * 32 + c];
      }
      const double mean = a / M;
      var = (b - M * mean * mean) / (M > 1 ? M - 1 : 1);
    }
    for (int o = 16; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    if (threadIdx.x == 0) parts[blockIdx.x] = var;
  }
}
__global__ void colvar_final_kernel(float* out, const double* parts, int nstrips, int N) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nstrips; ++i) s += parts[i];
    out[0] = (float)(s / N);
  }
}

__global__ void l1_bwd_kernel(float* dA, int ldd, const float* A, int lda, const float* B, int ldb, const float* coef,
                              int idx, float scale, int M, int N, int accumulate) {
  const float k = coef[idx] * scale;
  CG_2D_LOOP(M, N) {
    const float d = A[(size_t)m * lda + col] - B[(size_t)m * ldb + col];
    const float g = d > 0.f ? k : (d < 0.f ? -k : 0.f);
    float* o = dA + (size_t)m * ldd + col;
    *o = accumulate ? *o + g : g;
  }
}

__global__ void poisson_bwd_kernel(float* dA, int ldd, const float* A, int lda, const float* B, int ldb,
                                   const float* coef, int idx, float scale, int M, int N, int accumulate) {
  const float k = coef[idx] * scale;
  CG_2D_LOOP(M, N) {
    const float g = k * (1.f - B[(size_t)m * ldb + col] / (A[(size_t)m * lda + col] + 1e-10f));
    float* o = dA + (size_t)m * ldd + col;
    *o = accumulate ? *o + g : g;
  }
}

__global__ void centroid_bwd_kernel(float* dA, int ldd, const float* A, int lda, const float* Cn, int ldc,
                                    const int64_t* labels, int lstride, const float* coef, int idx, float scale, int M,
                                    int N, int n_types, int accumulate) {
  const float k = coef[idx] * scale;
  CG_2D_LOOP(M, N) {
    const float g = k * (A[(size_t)m * lda + col] - Cn[(size_t)clamp_label(labels[(size_t)m * lstride], n_types) * ldc + col]);
    float* o = dA + (size_t)m * ldd + col;
    *o = accumulate ? *o + g : g;
  }
}

__global__ void fm_bwd_kernel(float* dF, int ldd, const float* mf, const float* mr, const float* coef, int idx,
                              float scale, int M, int N, int accumulate) {
  const float k = coef[idx] * scale;
  CG_2D_LOOP(M, N) {
    const float g = k * (mf[col] - mr[col]);
    float* o = dF + (size_t)m * ldd + col;
    *o = accumulate ? *o + g : g;
  }
}

// ---- loss bookkeeping (float64) ------------ //
// Sum of a slot's per-block partials by one wave: lane l adds entries l, l+64, ... then the lanes are combined by
// a fixed shuffle tree, so the result does not depend on timing.  Every lane returns the total.
__device__ double sum_slot(const float* parts, const int* nparts, int slot) {
  double s = 0.0;
  int n = nparts[slot];
  if (n > CG_MAX_PARTS) n = CG_MAX_PARTS;
  const float* p = parts + (size_t)slot * CG_MAX_PARTS;
  for (int i = threadIdx.x & 63; i < n; i += 64) s += (double)p[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
}

__global__ void __launch_bounds__(64) gloss_finalize_kernel(double* ema, double* losses, float* coef, const float* parts,
                                                            const int* nparts, cg_gloss_desc d) {
  double L[6];
  for (int k = 0; k < 6; ++k) {
    // each term is rounded to float32 like the tensor the reference calls .item() on
    float acc = 0.f;
    for (int j = 0; j < d.n_terms[k]; ++j) acc = acc + (float)(d.weight[k][j] * sum_slot(parts, nparts, d.slot[k][j]));
    L[k] = (double)acc;
  }
  if (threadIdx.x != 0) return;
  if (d.weights) {      // weights given by the caller: no moving averages, no re-scaling
    double g = 0.0;
    for (int k = 0; k < 5; ++k) {
      const float c = d.weights[k];
      coef[k] = c;
      g += (double)c * L[k];
      losses[k] = L[k];
      losses[6 + k] = 1.0;
    }
    coef[5] = (float)d.lambda[5];
    losses[11] = L[5];
    losses[5] = g + d.lambda[5] * L[5];
    return;
  }
  const double a = d.ema_alpha;
  double sc[5];
  for (int k = 0; k < 5; ++k) ema[k] = a * L[k] + (1.0 - a) * ema[k];
  sc[0] = 1.0 / (fabs(ema[0]) + 1e-8);
  for (int k = 1; k < 5; ++k) sc[k] = 1.0 / (ema[k] + 1e-8);
  double mn = sc[0];
  for (int k = 1; k < 5; ++k) mn = sc[k] < mn ? sc[k] : mn;
  double g = 0.0;
  for (int k = 0; k < 5; ++k) {
    sc[k] /= mn;
    const double c = d.lambda[k] * sc[k];
    coef[k] = (float)c;
    g += c * L[k];
    losses[k] = L[k];
    losses[6 + k] = sc[k];
  }
  // sixth term (identity mapping, no counterpart in GAN_functions.py): fixed weight, outside the re-scaling
  coef[5] = (float)d.lambda[5];
  losses[11] = L[5];
  losses[5] = g + d.lambda[5] * L[5];
}

__global__ void __launch_bounds__(64) sum_parts_kernel(double* totals, const float* parts, const int* nparts,
                                                       const int* slots, const double* scales, int n) {
  const int i = blockIdx.x;      // one wave per slot
  const double s = sum_slot(parts, nparts, slots[i]);
  if (threadIdx.x == 0) totals[i] = scales[i] * s;
}

}  // namespace

// Sums of the rows of X by label: out[t][n] (+)= scale * sum over rows m with labels[m] == t of X[m][n]; rows whose
// label lies outside [0, n_types) belong to no cell and are skipped; with `minus`, a second label per row, the row is also
// subtracted at that label.  A workgroup owns 16 columns over all rows with
// RL = blockDim.x / 16 row lanes: a thread walks every RL-th row and adds into the LDS accumulator of its own
// (row lane, column) for the row's label (eight loads in flight, unconditional: row clamped, contribution zeroed),
// then the row lanes are added in order - deterministic.  A wave's 4 row lanes x 16 columns fall on 64 distinct
// banks whatever their labels (a label moves the address by a multiple of 64 words).  LDS: n_types * RL * 64 bytes,
// so RL = 64 up to 16 labels and 32 up to LC_MAX_TYPES.
constexpr int LC_MAX_TYPES = 32;
__global__ void __launch_bounds__(1024) label_colsum_kernel(float* out, const float* X, int ldx, const int64_t* labels,
                                                            int lstride, int M, int N, int n_types, float scale,
                                                            int accumulate, const int64_t* minus, int mstride) {
  extern __shared__ float lc_acc[];     // [n_types][RL][16]
  const int RL = blockDim.x >> 4, c = threadIdx.x & 15, rl = threadIdx.x >> 4, col = blockIdx.x * 16 + c,
            colc = min(col, N - 1);
  for (int i = threadIdx.x; i < n_types * RL * 16; i += blockDim.x) lc_acc[i] = 0.f;
  __syncthreads();
  float* mine = lc_acc + rl * 16 + c;
  for (int m0 = rl; m0 < M; m0 += RL * 8) {
    float v[8];
    long long lab[8], labm[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = min(m0 + RL * u, M - 1);
      v[u] = X[(size_t)m * ldx + colc];
      lab[u] = labels[(size_t)m * lstride];
      labm[u] = minus ? minus[(size_t)m * mstride] : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (m0 + RL * u >= M) continue;
      if (lab[u] >= 0 && lab[u] < n_types) mine[(int)lab[u] * RL * 16] += v[u];
      if (labm[u] >= 0 && labm[u] < n_types) mine[(int)labm[u] * RL * 16] -= v[u];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_types * 16; i += blockDim.x) {
    const int t = i >> 4, cc = i & 15, n = blockIdx.x * 16 + cc;
    if (n >= N) continue;
    float h = 0.f;
    for (int r = 0; r < RL; ++r) h += lc_acc[(t * RL + r) * 16 + cc];
    float* o = out + (size_t)t * N + n;
    *o = (accumulate ? *o : 0.f) + scale * h;
  }
}

#define ST ((hipStream_t)stream)

extern "C" int cg_axpby(float* out, int ldo, float a, const float* x, int ldx, float b, const float* y, int ldy, int M,
                        int N, void* stream) {
  if (!out || !x || M < 0 || N < 0 || ldo < N || ldx < N || (y && ldy < N)) return CG_ERR_BAD_ARG;
  if (M == 0 || N == 0) return CG_OK;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(axpby_kernel, g.grid, g.block, 0, ST, out, ldo, a, x, ldx, b, y, ldy, M, N);
  return cg_launch_status();
}

extern "C" int cg_mul(float* out, int ldo, const float* x, int ldx, const float* y, int ldy, int M, int N,
                      void* stream) {
  if (!out || !x || !y || M < 0 || N < 0 || ldo < N || ldx < N || ldy < N) return CG_ERR_BAD_ARG;
  if (M == 0 || N == 0) return CG_OK;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(mul_kernel, g.grid, g.block, 0, ST, out, ldo, x, ldx, y, ldy, M, N);
  return cg_launch_status();
}

extern "C" int cg_fill_coef(float* out, int ldo, int M, int N, const float* coef, int idx, float scale, void* stream) {
  if (!out || !coef || M < 0 || N < 0 || ldo < N || idx < 0) return CG_ERR_BAD_ARG;
  if (M == 0 || N == 0) return CG_OK;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(fill_coef_kernel, g.grid, g.block, 0, ST, out, ldo, M, N, coef, idx, scale);
  return cg_launch_status();
}

extern "C" int cg_to_bf16(void* out, int ldo, const float* x, int ldx, int M, int N, void* stream) {
  if (!out || !x || M < 0 || N < 0 || ldo < N || ldx < N) return CG_ERR_BAD_ARG;
  if (M == 0 || N == 0) return CG_OK;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(to_bf16_kernel, g.grid, g.block, 0, ST, (u16*)out, ldo, x, ldx, M, N);
  return cg_launch_status();
}

extern "C" int cg_copy_bf16(float* out, void* out_bf16, int ldo, const float* x, int ldx, int M, int N, void* stream) {
  if (!out || !out_bf16 || !x || M < 0 || N < 0 || ldo < N || ldx < N) return CG_ERR_BAD_ARG;
  if (M == 0 || N == 0) return CG_OK;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(copy_bf16_kernel, g.grid, g.block, 0, ST, out, (u16*)out_bf16, ldo, x, ldx, M, N);
  return cg_launch_status();
}

extern "C" int cg_embed_gather(float* out, int ldo, const float* table, int E, const int64_t* labels, int lstride,
                               int M, int n_types, void* out_bf16, void* stream) {
  if (!out || !table || !labels || E <= 0 || M < 0 || ldo < E || lstride < 1 || n_types <= 0) return CG_ERR_BAD_ARG;
  if (M == 0) return CG_OK;
  hipLaunchKernelGGL(embed_gather_kernel, dim3((M * E + 255) / 256), dim3(256), 0, ST, out, ldo, table, E, labels,
                     lstride, M, n_types, (u16*)out_bf16);
  return cg_launch_status();
}

extern "C" int cg_embed_scatter(float* dtable, int E, const float* dout, int lddo, const int64_t* labels, int lstride,
                                int M, int n_types, int accumulate, void* stream) {
  if (!dtable || !dout || !labels || E <= 0 || M < 0 || lddo < E || lstride < 1 || n_types <= 0) return CG_ERR_BAD_ARG;
  if (E > 128) return CG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(embed_scatter_kernel, dim3(n_types), dim3(1024), 0, ST, dtable, E, dout, lddo, labels, lstride, M,
                     n_types, accumulate);
  return cg_launch_status();
}

extern "C" int cg_label_colsum(float* out, const float* X, int ldx, const int64_t* labels, int lstride, int M, int N,
                               int n_types, float scale, int accumulate, const int64_t* minus_labels, int mstride,
                               void* stream) {
  if (!out || !X || !labels || M < 0 || N <= 0 || ldx < N || lstride < 1 || n_types <= 0) return CG_ERR_BAD_ARG;
  if (minus_labels && mstride < 1) return CG_ERR_BAD_ARG;
  if (n_types > LC_MAX_TYPES) return CG_ERR_UNSUPPORTED;
  const int RL = n_types <= 16 ? 64 : 32;
  hipLaunchKernelGGL(label_colsum_kernel, dim3((N + 15) / 16), dim3(RL * 16), (size_t)n_types * RL * 64, ST, out, X, ldx,
                     labels, lstride, M, N, n_types, scale, accumulate, minus_labels, mstride);
  return cg_launch_status();
}

extern "C" int cg_reparam_fwd(float* z, int ldz, const float* ml, int ldml, const float* eps, int M, int L,
                              void* stream) {
  if (!z || !ml || !eps || M <= 0 || L <= 0 || ldz < L || ldml < 2 * L) return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, L);
  hipLaunchKernelGGL(reparam_fwd_kernel, g.grid, g.block, 0, ST, z, ldz, ml, ldml, eps, M, L);
  return cg_launch_status();
}

extern "C" int cg_reparam_bwd(float* dml, int ldd, const float* dz, int lddz, const float* ml, int ldml,
                              const float* eps, const float* coef, int kl_idx, float kl_weight, int M, int L,
                              void* stream) {
  if (!dml || !dz || !ml || !eps || M <= 0 || L <= 0 || ldd < 2 * L || lddz < L || ldml < 2 * L) return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, L);
  hipLaunchKernelGGL(reparam_bwd_kernel, g.grid, g.block, 0, ST, dml, ldd, dz, lddz, ml, ldml, eps, coef, kl_idx,
                     kl_weight, M, L);
  return cg_launch_status();
}

extern "C" int cg_libsize_fwd(float* Y, int ldy, const float* X, int ldx, float* rowsum, float library_size, int M,
                              int N, void* stream) {
  if (!Y || !X || !rowsum || M <= 0 || N <= 0 || ldy < N || ldx < N) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(libsize_fwd_kernel, dim3(M), dim3(256), 0, ST, Y, ldy, X, ldx, rowsum, library_size, N);
  return cg_launch_status();
}

extern "C" int cg_libsize_bwd(float* dX, int lddx, const float* dY, int lddy, const float* X, int ldx,
                              const float* rowsum, float library_size, int M, int N, void* stream) {
  if (!dX || !dY || !X || !rowsum || M <= 0 || N <= 0 || lddx < N || lddy < N || ldx < N) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(libsize_bwd_kernel, dim3(M), dim3(256), 0, ST, dX, lddx, dY, lddy, X, ldx, rowsum, library_size,
                     N);
  return cg_launch_status();
}

extern "C" int cg_gp_interp(float* out, int ldo, const float* real, int ldr, const float* fake, int ldf,
                            const float* alpha, int M, int G, void* stream) {
  if (!out || !real || !fake || !alpha || M <= 0 || G <= 0 || ldo < G || ldr < G || ldf < G) return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, G);
  hipLaunchKernelGGL(gp_interp_kernel, g.grid, g.block, 0, ST, out, ldo, real, ldr, fake, ldf, alpha, M, G);
  return cg_launch_status();
}

extern "C" int cg_critic_inputs(float* out, int ldo, const float* real, int ldr, const float* fake, int ldf,
                                const float* alpha, int M, int G, int64_t* lab_out, const int64_t* real_lab,
                                const int64_t* fake_lab, int n_cov, int interp_block, void* out_bf16, void* stream) {
  if (!out || !real || !fake || !alpha || M <= 0 || G <= 0 || ldo < G || ldr < G || ldf < G) return CG_ERR_BAD_ARG;
  if (interp_block < 2) return CG_ERR_BAD_ARG;
  if (lab_out && (!real_lab || !fake_lab || n_cov <= 0)) return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, G);
  hipLaunchKernelGGL(critic_inputs_kernel, g.grid, g.block, 0, ST, out, ldo, real, ldr, fake, ldf, alpha, M, G, lab_out,
                     real_lab, fake_lab, n_cov, (u16*)out_bf16, interp_block);
  return cg_launch_status();
}

extern "C" int cg_interp_hidden(float* H, int ldh, const float* alpha, const float* emb, int lde, int E,
                                const float* We, int ldw, int M, int N, void* stream) {
  if (!H || !alpha || M <= 0 || N <= 0 || ldh < N) return CG_ERR_BAD_ARG;
  if (emb && (!We || E <= 0 || lde < E || ldw < E)) return CG_ERR_BAD_ARG;
  if (emb && E <= IH_EMAX) {
    hipLaunchKernelGGL(interp_hidden_tiled_kernel, dim3((N + IH_C - 1) / IH_C, (M + IH_R - 1) / IH_R), dim3(256), 0, ST, H,
                       ldh, alpha, emb, lde, E, We, ldw, M, N);
    return cg_launch_status();
  }
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(interp_hidden_kernel, g.grid, g.block, 0, ST, H, ldh, alpha, emb, lde, E, We, ldw, M, N);
  return cg_launch_status();
}

extern "C" int cg_gp_combine(float* D, int ldd, const float* VH, int ldv, const float* alpha, float* AV, int ldav,
                             const float* emb, int lde, int E, float* dE, int ldde, int M, int N, void* d_bf16,
                             void* stream) {
  if (!D || !VH || !alpha || !AV || M <= 0 || N <= 0 || ldd < N || ldv < N || ldav < N) return CG_ERR_BAD_ARG;
  if (emb && (!dE || E <= 0 || E > 256 || lde < E || ldde < E)) return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(gp_combine_kernel, g.grid, g.block, 0, ST, D, ldd, VH, ldv, alpha, AV, ldav, emb, lde, E, dE, ldde,
                     M, N, (u16*)d_bf16);
  return cg_launch_status();
}

static int launch_gp_penalty(float* V, int ldv, float* rowpen, int* nparts, const float* UX, int ldu, float lambda_gp,
                             int M, int G, void* v_bf16, void* stream) {
  if (!V || !rowpen || !UX || M <= 0 || G <= 0 || ldv < G || ldu < G) return CG_ERR_BAD_ARG;
  if (G <= 256 * 20)
    hipLaunchKernelGGL(gp_penalty_regs_kernel<20>, dim3(M), dim3(256), 0, ST, V, ldv, rowpen, UX, ldu,
                       lambda_gp * 2.f / (float)M, G, (u16*)v_bf16, nparts);
  else
    hipLaunchKernelGGL(gp_penalty_kernel, dim3(M), dim3(256), 0, ST, V, ldv, rowpen, UX, ldu, lambda_gp * 2.f / (float)M,
                       G, (u16*)v_bf16, nparts);
  return cg_launch_status();
}

extern "C" int cg_gp_penalty(float* V, int ldv, float* rowpen, const float* UX, int ldu, float lambda_gp, int M, int G,
                             void* v_bf16, void* stream) {
  return launch_gp_penalty(V, ldv, rowpen, nullptr, UX, ldu, lambda_gp, M, G, v_bf16, stream);
}

extern "C" int cg_gp_penalty_parts(float* V, int ldv, float* parts, int* nparts, const float* UX, int ldu,
                                   float lambda_gp, int M, int G, void* v_bf16, void* stream) {
  if (!nparts || M > CG_MAX_PARTS) return CG_ERR_BAD_ARG;
  return launch_gp_penalty(V, ldv, parts, nparts, UX, ldu, lambda_gp, M, G, v_bf16, stream);
}

extern "C" int cg_reduce_sum(float* parts, int* nparts, const float* X, int ldx, int M, int N, void* stream) {
  if (!parts || !nparts || !X || M <= 0 || N <= 0 || ldx < N) return CG_ERR_BAD_ARG;
  Grid2 g = grid_red(M, N);
  hipLaunchKernelGGL(reduce_sum_kernel, g.grid, g.block, 0, ST, parts, nparts, X, ldx, M, N);
  return cg_launch_status();
}

extern "C" int cg_reduce_l1(float* parts, int* nparts, const float* A, int lda, const float* B, int ldb, int M, int N,
                            void* stream) {
  if (!parts || !nparts || !A || !B || M <= 0 || N <= 0 || lda < N || ldb < N) return CG_ERR_BAD_ARG;
  Grid2 g = grid_red(M, N);
  hipLaunchKernelGGL(reduce_l1_kernel, g.grid, g.block, 0, ST, parts, nparts, A, lda, B, ldb, M, N);
  return cg_launch_status();
}

extern "C" int cg_reduce_poisson(float* parts, int* nparts, const float* pred, int ldp, const float* target, int ldt,
                                 int M, int N, void* stream) {
  if (!parts || !nparts || !pred || !target || M <= 0 || N <= 0 || ldp < N || ldt < N) return CG_ERR_BAD_ARG;
  Grid2 g = grid_red(M, N);
  hipLaunchKernelGGL(reduce_poisson_kernel, g.grid, g.block, 0, ST, parts, nparts, pred, ldp, target, ldt, M, N);
  return cg_launch_status();
}

extern "C" int cg_reduce_centroid_sq(float* parts, int* nparts, const float* A, int lda, const float* Cn, int ldc,
                                     const int64_t* labels, int lstride, int M, int N, int n_types, void* stream) {
  if (!parts || !nparts || !A || !Cn || !labels || M <= 0 || N <= 0 || lda < N || ldc < N || lstride < 1 || n_types < 1)
    return CG_ERR_BAD_ARG;
  Grid2 g = grid_red(M, N);
  hipLaunchKernelGGL(reduce_centroid_kernel, g.grid, g.block, 0, ST, parts, nparts, A, lda, Cn, ldc, labels, lstride, M,
                     N, n_types);
  return cg_launch_status();
}

extern "C" int cg_reduce_kl(float* parts, int* nparts, const float* ml, int ldml, int M, int L, void* stream) {
  if (!parts || !nparts || !ml || M <= 0 || L <= 0 || ldml < 2 * L) return CG_ERR_BAD_ARG;
  Grid2 g = grid_red(M, L);
  hipLaunchKernelGGL(reduce_kl_kernel, g.grid, g.block, 0, ST, parts, nparts, ml, ldml, M, L);
  return cg_launch_status();
}

extern "C" int cg_colsum(float* out, const float* X, int ldx, int M, int N, float scale, int accumulate,
                         void* stream) {
  if (!out || !X || M <= 0 || N <= 0 || ldx < N) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 31) / 32), dim3(1024), 0, ST, out, X, ldx, M, N, scale, accumulate);
  return cg_launch_status();
}

// Up to four contiguous copies in one launch: blockIdx.y names the job, its workgroups stride over 16-byte pieces
// (4-byte words when an address or a size is not a multiple of 16).
struct CopyJobs {
  void* dst[4];
  const void* src[4];
  unsigned long long bytes[4];
  int wide[4];
};
__global__ void __launch_bounds__(256) copy4_kernel(CopyJobs j) {
  const int k = blockIdx.y;
  const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
  if (j.wide[k]) {
    const uint4* s = static_cast<const uint4*>(j.src[k]);
    uint4* d = static_cast<uint4*>(j.dst[k]);
    const size_t n = j.bytes[k] >> 4;
    for (size_t i = i0; i < n; i += step) d[i] = s[i];
  } else {
    const uint32_t* s = static_cast<const uint32_t*>(j.src[k]);
    uint32_t* d = static_cast<uint32_t*>(j.dst[k]);
    const size_t n = j.bytes[k] >> 2;
    for (size_t i = i0; i < n; i += step) d[i] = s[i];
  }
}

extern "C" int cg_copy4(void* dst0, const void* src0, long long bytes0, void* dst1, const void* src1, long long bytes1,
                        void* dst2, const void* src2, long long bytes2, void* dst3, const void* src3, long long bytes3,
                        void* stream) {
  void* ds[4] = {dst0, dst1, dst2, dst3};
  const void* ss[4] = {src0, src1, src2, src3};
  const long long bs[4] = {bytes0, bytes1, bytes2, bytes3};
  CopyJobs j{};
  int n = 0;
  long long most = 0;
  for (int k = 0; k < 4; ++k) {
    if (bs[k] == 0) continue;       // an unused job
    if (bs[k] < 0 || bs[k] % 4 || !ds[k] || !ss[k] || (uintptr_t)ds[k] % 4 || (uintptr_t)ss[k] % 4) return CG_ERR_BAD_ARG;
    j.dst[n] = ds[k];
    j.src[n] = ss[k];
    j.bytes[n] = (unsigned long long)bs[k];
    j.wide[n] = bs[k] % 16 == 0 && (uintptr_t)ds[k] % 16 == 0 && (uintptr_t)ss[k] % 16 == 0;
    most = bs[k] > most ? bs[k] : most;
    ++n;
  }
  if (n == 0) return CG_OK;
  long long blocks = (most / 16 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(copy4_kernel, dim3((unsigned)blocks, n), dim3(256), 0, ST, j);
  return cg_launch_status();
}

extern "C" int cg_colsum3(float* out0, const float* X0, int ldx0, int M0, int N0, float scale0, int accumulate0,
                          float* out1, const float* X1, int ldx1, int M1, int N1, float scale1, int accumulate1,
                          float* out2, const float* X2, int ldx2, int M2, int N2, float scale2, int accumulate2,
                          void* stream) {
  ColsumJobs js{};
  float* outs[3] = {out0, out1, out2};
  const float* Xs[3] = {X0, X1, X2};
  const int ld[3] = {ldx0, ldx1, ldx2}, Ms[3] = {M0, M1, M2}, Ns[3] = {N0, N1, N2}, ac[3] = {accumulate0, accumulate1, accumulate2};
  const float sc[3] = {scale0, scale1, scale2};
  int strips = 0;
  for (int k = 0; k < 3; ++k) {
    js.first[k] = strips;
    if (!outs[k]) {       // an unused job: no strips
      js.j[k] = ColsumJob{nullptr, nullptr, 0, 0, 0, 0.f, 0};
      continue;
    }
    if (!Xs[k] || Ms[k] <= 0 || Ns[k] <= 0 || ld[k] < Ns[k]) return CG_ERR_BAD_ARG;
    js.j[k] = ColsumJob{outs[k], Xs[k], ld[k], Ms[k], Ns[k], sc[k], ac[k]};
    strips += (Ns[k] + 31) / 32;
  }
  js.first[3] = strips;
  if (strips == 0) return CG_OK;
  hipLaunchKernelGGL(colsum_multi_kernel, dim3(strips), dim3(1024), 0, ST, js);
  return cg_launch_status();
}

extern "C" int cg_reduce_sum_seg(float* parts, int* nparts, const float* X, int ldx, int seg_rows, int nseg, int N,
                                 void* stream) {
  if (!parts || !nparts || !X || seg_rows <= 0 || nseg <= 0 || N <= 0 || ldx < N) return CG_ERR_BAD_ARG;
  Grid2 g = grid_red(seg_rows, N);
  hipLaunchKernelGGL(reduce_sum_seg_kernel, dim3(g.grid.x, g.grid.y, nseg), g.block, 0, ST, parts, nparts, X, ldx,
                     seg_rows, N);
  return cg_launch_status();
}

extern "C" int cg_reduce_sqdiff(float* parts, int* nparts, const float* A, int lda, const float* B, int ldb, int M,
                                int N, void* stream) {
  if (!parts || !nparts || !A || !B || M <= 0 || N <= 0 || lda < N || ldb < N) return CG_ERR_BAD_ARG;
  Grid2 g = grid_red(M, N);
  hipLaunchKernelGGL(reduce_sqdiff_kernel, g.grid, g.block, 0, ST, parts, nparts, A, lda, B, ldb, M, N);
  return cg_launch_status();
}

extern "C" int cg_colvar_mean(float* out, double* strip_ws, const float* X, int ldx, int M, int N, void* stream) {
  if (!out || !strip_ws || !X || M <= 0 || N <= 0 || ldx < N) return CG_ERR_BAD_ARG;
  const int ns = (N + 31) / 32;
  hipLaunchKernelGGL(colvar_kernel, dim3(ns), dim3(256), 0, ST, strip_ws, X, ldx, M, N);
  hipLaunchKernelGGL(colvar_final_kernel, dim3(1), dim3(64), 0, ST, out, strip_ws, ns, N);
  return cg_launch_status();
}

extern "C" int cg_l1_bwd(float* dA, int ldd, const float* A, int lda, const float* B, int ldb, const float* coef,
                         int idx, float scale, int M, int N, int accumulate, void* stream) {
  if (!dA || !A || !B || !coef || M <= 0 || N <= 0 || ldd < N || lda < N || ldb < N) return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(l1_bwd_kernel, g.grid, g.block, 0, ST, dA, ldd, A, lda, B, ldb, coef, idx, scale, M, N,
                     accumulate);
  return cg_launch_status();
}

extern "C" int cg_poisson_bwd(float* dA, int ldd, const float* pred, int ldp, const float* target, int ldt,
                              const float* coef, int idx, float scale, int M, int N, int accumulate, void* stream) {
  if (!dA || !pred || !target || !coef || M <= 0 || N <= 0 || ldd < N || ldp < N || ldt < N) return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(poisson_bwd_kernel, g.grid, g.block, 0, ST, dA, ldd, pred, ldp, target, ldt, coef, idx, scale, M,
                     N, accumulate);
  return cg_launch_status();
}

extern "C" int cg_centroid_bwd(float* dA, int ldd, const float* A, int lda, const float* Cn, int ldc,
                               const int64_t* labels, int lstride, const float* coef, int idx, float scale, int M,
                               int N, int n_types, int accumulate, void* stream) {
  if (!dA || !A || !Cn || !labels || !coef || M <= 0 || N <= 0 || ldd < N || lda < N || ldc < N || n_types < 1)
    return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(centroid_bwd_kernel, g.grid, g.block, 0, ST, dA, ldd, A, lda, Cn, ldc, labels, lstride, coef, idx,
                     scale, M, N, n_types, accumulate);
  return cg_launch_status();
}

extern "C" int cg_fm_bwd(float* dF, int ldd, const float* mean_fake, const float* mean_real, const float* coef, int idx,
                         float scale, int M, int N, int accumulate, void* stream) {
  if (!dF || !mean_fake || !mean_real || !coef || M <= 0 || N <= 0 || ldd < N) return CG_ERR_BAD_ARG;
  Grid2 g = grid2(M, N);
  hipLaunchKernelGGL(fm_bwd_kernel, g.grid, g.block, 0, ST, dF, ldd, mean_fake, mean_real, coef, idx, scale, M, N,
                     accumulate);
  return cg_launch_status();
}

extern "C" int cg_gloss_finalize(double* ema_state, double* losses_out, float* coef_out, const float* parts,
                                 const int* nparts, const cg_gloss_desc* desc, void* stream) {
  if (!ema_state || !losses_out || !coef_out || !parts || !nparts || !desc) return CG_ERR_BAD_ARG;
  for (int k = 0; k < 6; ++k)
    if (desc->n_terms[k] < 0 || desc->n_terms[k] > 4) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(gloss_finalize_kernel, dim3(1), dim3(64), 0, ST, ema_state, losses_out, coef_out, parts, nparts,
                     *desc);
  return cg_launch_status();
}

extern "C" int cg_sum_parts(double* totals, const float* parts, const int* nparts, const int* slots,
                            const double* scales, int n, void* stream) {
  if (!totals || !parts || !nparts || !slots || !scales || n <= 0) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(sum_parts_kernel, dim3(n), dim3(64), 0, ST, totals, parts, nparts, slots, scales, n);
  return cg_launch_status();
}

extern "C" int cg_version(void) { return 100; }
