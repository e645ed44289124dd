// Fused kernels for the ResidualBlock of the latent stage (GAN_functions.py:54-68) on the token view of the
// latent: x (n_tok x 16) -> Linear(16,16) -> BatchNorm1d -> ReLU -> Linear(16,16) -> BatchNorm1d, y = x + block(x).
//
// A token is one 64-byte row.  One thread owns one token (its 16 values live in registers), a workgroup owns
// 128 consecutive tokens, the two 16x16 weight matrices sit in LDS and are read as broadcasts.  BatchNorm needs
// statistics over all tokens, so the chain is cut where a statistic is needed and each piece is one launch:
//
//   forward   lin_fwd        t1 = x W1^T + b1                       + column sums of t1 per workgroup
//             bn_lin_fwd     r = relu(bn1(t1)); t2 = r W2^T + b2    + column sums of t2 per workgroup
//             bn_res_fwd     y = x + bn2(t2)
//   backward  bn_bwd_stats   sums of g, g*xhat2 for g = dy
//             bwd_mid        dt2 = bn2'(g); dW2 partial; g1 = (dt2 W2) * (r > 0); sums of g1, g1*xhat1
//             bwd_in         dt1 = bn1'(g1); dW1 partial; dx = dy + dt1 W1
//             wgrad_reduce   dW1, dW2 += sum of the per-workgroup partials
//
// Column sums are float64, written per workgroup and re-added in workgroup order by every consumer: no atomics,
// the same bits on every run.  Traffic is a few 512 KB sweeps per launch (L2-resident at batch 512): the point of
// the fusion is the launch count, 3 instead of 7 forward and 4 instead of 10 backward per block.
#include "common.h"

namespace {

constexpr int TD = 16;     // token width
constexpr int TB = 128;    // tokens per workgroup
constexpr int LS = TD + 1; // LDS row stride of a token tile (bank-conflict-free column walks)

struct Row {
  float v[TD];
};

// Token `tok` of p (zeros when !ok).  The loads are unconditional - a token past the end reads token 0 and discards
// it - because a branch around a load makes the compiler wait for that load before it issues the next one.
__device__ __forceinline__ void load_row(Row& r, const float* p, int tok, bool ok) {
  const float* src = p + (size_t)(ok ? tok : 0) * TD;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
    r.v[4 * q] = ok ? t[0] : 0.f;
    r.v[4 * q + 1] = ok ? t[1] : 0.f;
    r.v[4 * q + 2] = ok ? t[2] : 0.f;
    r.v[4 * q + 3] = ok ? t[3] : 0.f;
  }
}

__device__ __forceinline__ void store_row(float* p, int tok, const Row& r) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 t = {r.v[4 * q], r.v[4 * q + 1], r.v[4 * q + 2], r.v[4 * q + 3]};
    *reinterpret_cast<f32x4*>(p + (size_t)tok * TD + 4 * q) = t;
  }
}

// W (16x16, [out][in]) and optionally a bias into LDS; call before the first __syncthreads of the kernel.
__device__ __forceinline__ void stage_weights(float* wl, float* bl, const float* W, const float* b) {
  for (int i = threadIdx.x; i < TD * TD; i += TB) wl[i] = W[i];
  if (b && threadIdx.x < TD) bl[threadIdx.x] = b[threadIdx.x];
}

// out = x W^T + b   (nn.Linear forward)
__device__ __forceinline__ void linear(Row& out, const Row& x, const float* wl, const float* bl) {
#pragma unroll
  for (int o = 0; o < TD; ++o) {
    float a = bl[o];
#pragma unroll
    for (int i = 0; i < TD; ++i) a += x.v[i] * wl[o * TD + i];
    out.v[o] = a;
  }
}

// out = g W   (input gradient of nn.Linear)
__device__ __forceinline__ void linear_t(Row& out, const Row& g, const float* wl) {
#pragma unroll
  for (int i = 0; i < TD; ++i) out.v[i] = 0.f;
#pragma unroll
  for (int o = 0; o < TD; ++o) {
#pragma unroll
    for (int i = 0; i < TD; ++i) out.v[i] += g.v[o] * wl[o * TD + i];
  }
}

// Per-workgroup column sums: stat[c] = sum_tok a[tok][c], stat[16+c] = sum_tok a[tok][c] * b[tok][c] (float64
// products), written to part[blockIdx.x][32].  Rows of tokens beyond n_tok must be zero.  tile: 2*TB*LS floats,
// red: 4*32 doubles.
__device__ __forceinline__ void block_stats(double* part, const Row& a, const Row& b, float* tile, double* red) {
  float* ta = tile;
  float* tb = tile + TB * LS;
  __syncthreads();   // the tile may still be read by an earlier phase
#pragma unroll
  for (int c = 0; c < TD; ++c) {
    ta[threadIdx.x * LS + c] = a.v[c];
    tb[threadIdx.x * LS + c] = b.v[c];
  }
  __syncthreads();
  const int s = threadIdx.x & 31, quarter = threadIdx.x >> 5, c = s & 15;
  double acc = 0.0;
  if (s < 16) {
    for (int t = quarter * 32; t < quarter * 32 + 32; ++t) acc += (double)ta[t * LS + c];
  } else {
    for (int t = quarter * 32; t < quarter * 32 + 32; ++t) acc += (double)ta[t * LS + c] * (double)tb[t * LS + c];
  }
  red[quarter * 32 + s] = acc;
  __syncthreads();
  if (threadIdx.x < 32)
    part[(size_t)blockIdx.x * 32 + s] = ((red[s] + red[32 + s]) + red[64 + s]) + red[96 + s];
}

// Re-add the per-workgroup partials; tot: 32 doubles in LDS, valid after the call.  Each quarter of the workgroup
// adds a contiguous quarter of the partials in workgroup order (sixteen loads in flight at a time - a quarter's whole
// share at batch 512, one round trip: the loads, not the additions, are what takes time), then the four sums are
// added in quarter order: a fixed order, the same bits on every run.  red4: 4*32 doubles of LDS that nothing else is
// using at this point.
__device__ __forceinline__ void total_stats(double* tot, const double* part, int nblk, double* red4) {
  const int s = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int per = (nblk + 3) >> 2;
  const int b0 = q * per, b1 = min(nblk, b0 + per);
  double a = 0.0;
  for (int b = b0; b < b1; b += 16) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = part[(size_t)min(b + k, nblk - 1) * 32 + s];
#pragma unroll
    for (int k = 0; k < 16; ++k) a += b + k < b1 ? v[k] : 0.0;
  }
  red4[q * 32 + s] = a;
  __syncthreads();
  if (threadIdx.x < 32) tot[s] = ((red4[s] + red4[32 + s]) + red4[64 + s]) + red4[96 + s];
  __syncthreads();
}

// Per-workgroup partial of dW[o][i] = sum_tok d[tok][o] * x[tok][i] -> wpart[blockIdx.x][256].
__device__ __forceinline__ void block_wgrad(float* wpart, const Row& d, const Row& x, float* tile) {
  float* td = tile;
  float* tx = tile + TB * LS;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < TD; ++c) {
    td[threadIdx.x * LS + c] = d.v[c];
    tx[threadIdx.x * LS + c] = x.v[c];
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int idx = threadIdx.x + h * TB, o = idx >> 4, i = idx & 15;
    float a = 0.f;
    for (int t = 0; t < TB; ++t) a += td[t * LS + o] * tx[t * LS + i];
    wpart[(size_t)blockIdx.x * (TD * TD) + idx] = a;
  }
}

struct BnCols {        // per-column constants of a train-mode BatchNorm, in LDS
  float mean[TD], invstd[TD];
};

// mean / invstd from the totals (sum x, sum x^2); workgroup 0 saves them and updates the running statistics.
__device__ __forceinline__ void bn_fwd_cols(BnCols& bc, const double* tot, int n_tok, float eps, float* mean_out,
                                            float* invstd_out, float* rmean, float* rvar, float momentum) {
  if (threadIdx.x < TD) {
    const int c = threadIdx.x;
    const double mean_d = tot[c] / n_tok;
    double var_d = tot[16 + c] / n_tok - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    bc.mean[c] = mean;
    bc.invstd[c] = invstd;
    if (blockIdx.x == 0) {
      mean_out[c] = mean;
      invstd_out[c] = invstd;
      if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      if (rvar) {
        const float unb = (float)(var_d * ((double)n_tok / (double)(n_tok > 1 ? n_tok - 1 : 1)));
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
      }
    }
  }
  __syncthreads();
}

#define TOK_COMMON                                   \
  const int tok = blockIdx.x * TB + threadIdx.x;     \
  const bool ok = tok < n_tok;                       \
  const int nblk = gridDim.x;                        \
  (void)nblk;

// ---------------------------------------------------------------- forward
__global__ void __launch_bounds__(TB) tok_lin_fwd(float* T1, double* stat_out, const float* X, const float* W,
                                                  const float* b, int n_tok) {
  __shared__ float wl[TD * TD], bl[TD], tile[2 * TB * LS];
  __shared__ double red[4 * 32];
  TOK_COMMON
  stage_weights(wl, bl, W, b);
  Row x, t1;
  load_row(x, X, tok, ok);
  __syncthreads();
  linear(t1, x, wl, bl);
  if (ok) store_row(T1, tok, t1);
  else
#pragma unroll
    for (int c = 0; c < TD; ++c) t1.v[c] = 0.f;
  block_stats(stat_out, t1, t1, tile, red);
}

__global__ void __launch_bounds__(TB) tok_bn_lin_fwd(float* R, float* T2, double* stat_out, const float* T1,
                                                     const double* stat_in, const float* gamma, const float* beta,
                                                     float eps, float* mean_out, float* invstd_out, float* rmean,
                                                     float* rvar, float momentum, const float* W2, const float* b2,
                                                     int n_tok, int n_stat_in) {
  __shared__ float wl[TD * TD], bl[TD], tile[2 * TB * LS];
  __shared__ double red[4 * 32], tot[32];
  __shared__ BnCols bc;
  TOK_COMMON
  stage_weights(wl, bl, W2, b2);
  Row t1, r, t2;
  load_row(t1, T1, tok, ok);
  total_stats(tot, stat_in, n_stat_in, red);      // as many partials as the kernel that wrote t1 had workgroups
  bn_fwd_cols(bc, tot, n_tok, eps, mean_out, invstd_out, rmean, rvar, momentum);
#pragma unroll
  for (int c = 0; c < TD; ++c) {
    const float h = (t1.v[c] - bc.mean[c]) * bc.invstd[c] * gamma[c] + beta[c];
    r.v[c] = h > 0.f ? h : 0.f;
  }
  linear(t2, r, wl, bl);
  if (ok) {
    store_row(R, tok, r);
    store_row(T2, tok, t2);
  } else {
#pragma unroll
    for (int c = 0; c < TD; ++c) t2.v[c] = 0.f;
  }
  block_stats(stat_out, t2, t2, tile, red);
}

__global__ void __launch_bounds__(TB) tok_bn_res_fwd(float* Y, const float* T2, const double* stat_in,
                                                     const float* gamma, const float* beta, float eps,
                                                     float* mean_out, float* invstd_out, float* rmean, float* rvar,
                                                     float momentum, const float* X, int n_tok) {
  __shared__ double red[4 * 32], tot[32];
  __shared__ BnCols bc;
  TOK_COMMON
  Row t2, x, y;
  load_row(t2, T2, tok, ok);
  load_row(x, X, tok, ok);
  total_stats(tot, stat_in, nblk, red);
  bn_fwd_cols(bc, tot, n_tok, eps, mean_out, invstd_out, rmean, rvar, momentum);
#pragma unroll
  for (int c = 0; c < TD; ++c) {
    const float u = (t2.v[c] - bc.mean[c]) * bc.invstd[c] * gamma[c] + beta[c];
    y.v[c] = 1.0f * x.v[c] + 1.0f * u;
  }
  if (ok) store_row(Y, tok, y);
}

// ---------------------------------------------------------------- backward
__global__ void __launch_bounds__(TB) tok_bn_bwd_stats(double* stat_out, const float* dY, const float* T2,
                                                       const float* mean, const float* invstd, int n_tok) {
  __shared__ float tile[2 * TB * LS];
  __shared__ double red[4 * 32];
  TOK_COMMON
  Row g, t2, xh;
  load_row(g, dY, tok, ok);
  load_row(t2, T2, tok, ok);
#pragma unroll
  for (int c = 0; c < TD; ++c) xh.v[c] = ok ? (t2.v[c] - mean[c]) * invstd[c] : 0.f;
  block_stats(stat_out, g, xh, tile, red);
}

// BatchNorm backward of one column set: d = invstd*gamma*(g - mean(g) - xhat*mean(g*xhat)); workgroup 0 adds the
// parameter gradients.
__device__ __forceinline__ void bn_bwd_row(Row& d, const Row& g, const Row& xh, const double* tot, int n_tok,
                                           const float* invstd, const float* gamma, float* dgamma, float* dbeta) {
#pragma unroll
  for (int c = 0; c < TD; ++c) {
    const float gbar = (float)(tot[c] / n_tok), gxbar = (float)(tot[16 + c] / n_tok);
    d.v[c] = invstd[c] * gamma[c] * (g.v[c] - gbar - xh.v[c] * gxbar);
  }
  if (blockIdx.x == 0 && threadIdx.x < TD) {
    const int c = threadIdx.x;
    if (dbeta) dbeta[c] += (float)tot[c];
    if (dgamma) dgamma[c] += (float)tot[16 + c];
  }
}

__global__ void __launch_bounds__(TB) tok_bwd_mid(float* G1, double* stat_out, float* wpart, float* dgamma2,
                                                  float* dbeta2, const float* dY, const float* T2, const float* R,
                                                  const float* T1, const double* stat_in, const float* mean2,
                                                  const float* invstd2, const float* gamma2, const float* W2,
                                                  const float* mean1, const float* invstd1, int n_tok, int n_stat_in) {
  __shared__ float wl[TD * TD], tile[2 * TB * LS];
  __shared__ double red[4 * 32], tot[32];
  TOK_COMMON
  stage_weights(wl, nullptr, W2, nullptr);
  Row g, t2, xh, r, t1, dt2, dr, g1;
  load_row(g, dY, tok, ok);
  load_row(t2, T2, tok, ok);
  load_row(r, R, tok, ok);
  load_row(t1, T1, tok, ok);
  total_stats(tot, stat_in, n_stat_in, red);      // as many partials as the kernel that wrote them had workgroups
#pragma unroll
  for (int c = 0; c < TD; ++c) xh.v[c] = (t2.v[c] - mean2[c]) * invstd2[c];
  bn_bwd_row(dt2, g, xh, tot, n_tok, invstd2, gamma2, dgamma2, dbeta2);
  if (!ok)
#pragma unroll
    for (int c = 0; c < TD; ++c) dt2.v[c] = 0.f;
  block_wgrad(wpart, dt2, r, tile);
  linear_t(dr, dt2, wl);
#pragma unroll
  for (int c = 0; c < TD; ++c) {
    g1.v[c] = ok ? dr.v[c] * (r.v[c] > 0.f ? 1.f : 0.f) : 0.f;
    xh.v[c] = ok ? (t1.v[c] - mean1[c]) * invstd1[c] : 0.f;
  }
  if (ok) store_row(G1, tok, g1);
  block_stats(stat_out, g1, xh, tile, red);
}

__global__ void __launch_bounds__(TB) tok_bwd_in(float* dX, float* wpart, float* dgamma1, float* dbeta1,
                                                 const float* G1, const float* T1, const float* X, const float* dY,
                                                 const double* stat_in, const float* mean1, const float* invstd1,
                                                 const float* gamma1, const float* W1, int n_tok) {
  __shared__ float wl[TD * TD], tile[2 * TB * LS];
  __shared__ double red[4 * 32], tot[32];
  TOK_COMMON
  stage_weights(wl, nullptr, W1, nullptr);
  Row g1, t1, xh, x, dy, dt1, dxr;
  load_row(g1, G1, tok, ok);
  load_row(t1, T1, tok, ok);
  load_row(x, X, tok, ok);
  load_row(dy, dY, tok, ok);
  total_stats(tot, stat_in, nblk, red);
#pragma unroll
  for (int c = 0; c < TD; ++c) xh.v[c] = (t1.v[c] - mean1[c]) * invstd1[c];
  bn_bwd_row(dt1, g1, xh, tot, n_tok, invstd1, gamma1, dgamma1, dbeta1);
  if (!ok)
#pragma unroll
    for (int c = 0; c < TD; ++c) dt1.v[c] = 0.f;
  block_wgrad(wpart, dt1, x, tile);
  linear_t(dxr, dt1, wl);
#pragma unroll
  for (int c = 0; c < TD; ++c) dxr.v[c] = dy.v[c] + dxr.v[c];
  if (ok) store_row(dX, tok, dxr);
}

// dW += sum of the per-workgroup partials: workgroup 0 does dW2, workgroup 1 dW1.  1024 threads: thread (q, e) adds
// every fourth partial of element e (its loads are independent), the four sums are added in q order through LDS.
__global__ void __launch_bounds__(1024) tok_wgrad_reduce(float* dW2, float* dW1, const float* wpart2,
                                                         const float* wpart1, int nblk, long long dstride = 0,
                                                         long long wstride = 0) {
  __shared__ float part[4][TD * TD];
  // blockIdx.y: one of several ResidualBlocks reduced in one launch (their weights dstride apart, partials wstride)
  float* dst = (blockIdx.x == 0 ? dW2 : dW1) + (size_t)blockIdx.y * dstride;
  const float* src = (blockIdx.x == 0 ? wpart2 : wpart1) + (size_t)blockIdx.y * wstride;
  const int e = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int per = (nblk + 3) >> 2, b0 = q * per, b1 = min(nblk, b0 + per);
  float a = 0.f;
  for (int b = b0; b < b1; b += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[(size_t)min(b + k, nblk - 1) * (TD * TD) + e];
#pragma unroll
    for (int k = 0; k < 8; ++k) a += b + k < b1 ? v[k] : 0.f;
  }
  part[q][e] = a;
  __syncthreads();
  if (q == 0) dst[e] += ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
}

inline bool aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int cg_tok_blocks(int n_tok) { return n_tok > 0 ? (n_tok + TB - 1) / TB : 0; }

extern "C" int cg_tok_lin_fwd(float* T1, double* stat_out, const float* X, const float* W, const float* b, int n_tok,
                              void* stream) {
  if (!T1 || !stat_out || !X || !W || !b || n_tok <= 0 || !aligned16(T1) || !aligned16(X)) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_lin_fwd, dim3(cg_tok_blocks(n_tok)), dim3(TB), 0, ST, T1, stat_out, X, W, b, n_tok);
  return cg_launch_status();
}

extern "C" int cg_tok_bn_lin_fwd(float* R, float* T2, double* stat_out, const float* T1, const double* stat_in,
                                 const float* gamma, const float* beta, float eps, float* mean_out, float* invstd_out,
                                 float* running_mean, float* running_var, float momentum, const float* W2,
                                 const float* b2, int n_tok, void* stream) {
  if (!R || !T2 || !stat_out || !T1 || !stat_in || !gamma || !beta || !mean_out || !invstd_out || !W2 || !b2 ||
      n_tok <= 0 || !aligned16(R) || !aligned16(T2) || !aligned16(T1))
    return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_bn_lin_fwd, dim3(cg_tok_blocks(n_tok)), dim3(TB), 0, ST, R, T2, stat_out, T1, stat_in, gamma,
                     beta, eps, mean_out, invstd_out, running_mean, running_var, momentum, W2, b2, n_tok,
                     cg_tok_blocks(n_tok));
  return cg_launch_status();
}

extern "C" int cg_tok_bn_lin_fwd_from(float* R, float* T2, double* stat_out, const float* T1, const double* stat_in,
                                      int n_stat_in, const float* gamma, const float* beta, float eps, float* mean_out,
                                      float* invstd_out, float* running_mean, float* running_var, float momentum,
                                      const float* W2, const float* b2, int n_tok, void* stream) {
  if (!R || !T2 || !stat_out || !T1 || !stat_in || !gamma || !beta || !mean_out || !invstd_out || !W2 || !b2 ||
      n_tok <= 0 || n_stat_in <= 0 || !aligned16(R) || !aligned16(T2) || !aligned16(T1))
    return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_bn_lin_fwd, dim3(cg_tok_blocks(n_tok)), dim3(TB), 0, ST, R, T2, stat_out, T1, stat_in, gamma,
                     beta, eps, mean_out, invstd_out, running_mean, running_var, momentum, W2, b2, n_tok, n_stat_in);
  return cg_launch_status();
}

extern "C" int cg_tok_bn_res_fwd(float* Y, const float* T2, const double* stat_in, const float* gamma,
                                 const float* beta, float eps, float* mean_out, float* invstd_out, float* running_mean,
                                 float* running_var, float momentum, const float* X, int n_tok, void* stream) {
  if (!Y || !T2 || !stat_in || !gamma || !beta || !mean_out || !invstd_out || !X || n_tok <= 0 || !aligned16(Y) ||
      !aligned16(T2) || !aligned16(X))
    return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_bn_res_fwd, dim3(cg_tok_blocks(n_tok)), dim3(TB), 0, ST, Y, T2, stat_in, gamma, beta, eps,
                     mean_out, invstd_out, running_mean, running_var, momentum, X, n_tok);
  return cg_launch_status();
}

extern "C" int cg_tok_bn_bwd_stats(double* stat_out, const float* dY, const float* T2, const float* mean,
                                   const float* invstd, int n_tok, void* stream) {
  if (!stat_out || !dY || !T2 || !mean || !invstd || n_tok <= 0 || !aligned16(dY) || !aligned16(T2))
    return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_bn_bwd_stats, dim3(cg_tok_blocks(n_tok)), dim3(TB), 0, ST, stat_out, dY, T2, mean, invstd,
                     n_tok);
  return cg_launch_status();
}

extern "C" int cg_tok_bwd_mid(float* G1, double* stat_out, float* wpart, float* dgamma2, float* dbeta2, const float* dY,
                              const float* T2, const float* R, const float* T1, const double* stat_in,
                              const float* mean2, const float* invstd2, const float* gamma2, const float* W2,
                              const float* mean1, const float* invstd1, int n_tok, void* stream) {
  if (!G1 || !stat_out || !wpart || !dY || !T2 || !R || !T1 || !stat_in || !mean2 || !invstd2 || !gamma2 || !W2 ||
      !mean1 || !invstd1 || n_tok <= 0)
    return CG_ERR_BAD_ARG;
  if (!aligned16(G1) || !aligned16(dY) || !aligned16(T2) || !aligned16(R) || !aligned16(T1)) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_bwd_mid, dim3(cg_tok_blocks(n_tok)), dim3(TB), 0, ST, G1, stat_out, wpart, dgamma2, dbeta2, dY,
                     T2, R, T1, stat_in, mean2, invstd2, gamma2, W2, mean1, invstd1, n_tok, cg_tok_blocks(n_tok));
  return cg_launch_status();
}

extern "C" int cg_tok_bwd_mid_from(float* G1, double* stat_out, float* wpart, float* dgamma2, float* dbeta2,
                                   const float* dY, const float* T2, const float* R, const float* T1,
                                   const double* stat_in, int n_stat_in, const float* mean2, const float* invstd2,
                                   const float* gamma2, const float* W2, const float* mean1, const float* invstd1,
                                   int n_tok, void* stream) {
  if (!G1 || !stat_out || !wpart || !dY || !T2 || !R || !T1 || !stat_in || !mean2 || !invstd2 || !gamma2 || !W2 ||
      !mean1 || !invstd1 || n_tok <= 0 || n_stat_in <= 0)
    return CG_ERR_BAD_ARG;
  if (!aligned16(G1) || !aligned16(dY) || !aligned16(T2) || !aligned16(R) || !aligned16(T1)) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_bwd_mid, dim3(cg_tok_blocks(n_tok)), dim3(TB), 0, ST, G1, stat_out, wpart, dgamma2, dbeta2, dY,
                     T2, R, T1, stat_in, mean2, invstd2, gamma2, W2, mean1, invstd1, n_tok, n_stat_in);
  return cg_launch_status();
}

extern "C" int cg_tok_bwd_in(float* dX, float* wpart, float* dgamma1, float* dbeta1, const float* G1, const float* T1,
                             const float* X, const float* dY, const double* stat_in, const float* mean1,
                             const float* invstd1, const float* gamma1, const float* W1, int n_tok, void* stream) {
  if (!dX || !wpart || !G1 || !T1 || !X || !dY || !stat_in || !mean1 || !invstd1 || !gamma1 || !W1 || n_tok <= 0)
    return CG_ERR_BAD_ARG;
  if (!aligned16(dX) || !aligned16(G1) || !aligned16(T1) || !aligned16(X) || !aligned16(dY)) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_bwd_in, dim3(cg_tok_blocks(n_tok)), dim3(TB), 0, ST, dX, wpart, dgamma1, dbeta1, G1, T1, X, dY,
                     stat_in, mean1, invstd1, gamma1, W1, n_tok);
  return cg_launch_status();
}

extern "C" int cg_tok_wgrad_reduce(float* dW2, float* dW1, const float* wpart2, const float* wpart1, int n_tok,
                                   void* stream) {
  if (!dW2 || !dW1 || !wpart2 || !wpart1 || n_tok <= 0) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_wgrad_reduce, dim3(2), dim3(1024), 0, ST, dW2, dW1, wpart2, wpart1, cg_tok_blocks(n_tok), 0LL,
                     0LL);
  return cg_launch_status();
}

extern "C" int cg_tok_wgrad_reduce_blocks(float* dW2, float* dW1, long long dstride, const float* wpart2,
                                          const float* wpart1, long long wstride, int n_res_blocks, int n_tok,
                                          void* stream) {
  if (!dW2 || !dW1 || !wpart2 || !wpart1 || n_res_blocks <= 0 || n_tok <= 0) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(tok_wgrad_reduce, dim3(2, n_res_blocks), dim3(1024), 0, ST, dW2, dW1, wpart2, wpart1,
                     cg_tok_blocks(n_tok), dstride, wstride);
  return cg_launch_status();
}
