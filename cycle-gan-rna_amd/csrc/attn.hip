// Fused token self-attention + LayerNorm of the generator bottleneck for
// 16 tokens x 16 features x 4 heads (head_dim 4), forward and backward.
//
// One wave handles one cell at a time: lane l <-> (token i = l & 15, head
// h = l >> 4).  A lane keeps its token's q/k/v head slices, its 16 attention
// probabilities and its 4 output features in registers; the few exchanges
// between lanes (all keys/values of a head, the concatenated heads for the
// output projection, transposed probability tiles in the backward) go through
// a private per-wave LDS area.  The backward recomputes the forward from X, so
// nothing but X and the dropout keep-mask is kept between the two.
//
// Parameter block layout (floats), identical for params and dparams:
//   Win[48][16] | bin[48] | Wout[16][16] | bout[16] | ln_w[16] | ln_b[16]
#include "common.h"

namespace {

constexpr int S = 16, E = 16, H = 4, HD = 4;
constexpr int OFF_WIN = 0, OFF_BIN = 768, OFF_WOUT = 816, OFF_BOUT = 1072, OFF_LNW = 1088, OFF_LNB = 1104;
constexpr int NPARAM = CG_ATTN_NPARAM;   // 1120
constexpr float LN_EPS = 1e-5f;
constexpr int TR = 20;                   // row stride of the transposed tiles [h][j][i] (bank spread for b128 reads)
constexpr int HS = 16 * TR + 16;         // their head stride (heads 0/1 of a 32-lane store group on different banks)
constexpr int CATW = 113;                // odd row stride of the per-token gradient rows

__device__ __forceinline__ float quad_sum(float v) {   // over the 4 head-lanes of a token
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

struct Fwd {
  float x[E];        // token row
  float q[HD], k[HD], v[HD];
  float a[S];        // softmax probabilities (before dropout)
  float p[S];        // after dropout
  float o[HD];
  float s[HD];       // x + attention output, features 4h..4h+3
  float xhat[HD], rstd;
};

// Per-wave LDS scratch for the forward pieces.
struct WaveLdsFwd {
  float ks[H][S][HD];
  float vs[H][S][HD];
  float os[S][E];
};

// Token i of `cell` into f.x, and this lane's own four features (4h .. 4h+3) into xown - fetched by address: indexing
// f.x with the lane's head would put the whole struct into scratch memory.
__device__ __forceinline__ void load_token(Fwd& f, f32x4& xown, const float* __restrict__ X, int cell, int i, int h) {
  const f32x4* xr = reinterpret_cast<const f32x4*>(X + ((size_t)cell * S + i) * E);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 t = xr[c];
    f.x[4 * c] = t[0]; f.x[4 * c + 1] = t[1]; f.x[4 * c + 2] = t[2]; f.x[4 * c + 3] = t[3];
  }
  xown = xr[h];
}

// The forward of one cell from f.x / xown (load_token, or the fused block kernel's own prologue).
__device__ __forceinline__ void forward_core(Fwd& f, const f32x4 xown, const float* __restrict__ keep,
                                             const float* W /* LDS params */, WaveLdsFwd& L, int cell, int i, int h) {
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    float q = W[OFF_BIN + 4 * h + c], k = W[OFF_BIN + 16 + 4 * h + c], v = W[OFF_BIN + 32 + 4 * h + c];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      q += f.x[e] * W[OFF_WIN + (4 * h + c) * E + e];
      k += f.x[e] * W[OFF_WIN + (16 + 4 * h + c) * E + e];
      v += f.x[e] * W[OFF_WIN + (32 + 4 * h + c) * E + e];
    }
    f.q[c] = q; f.k[c] = k; f.v[c] = v;
    L.ks[h][i][c] = k;
    L.vs[h][i][c] = v;
  }
  __syncthreads();
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < S; ++j) {
    float sc = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) sc += (f.q[c] * 0.5f) * L.ks[h][j][c];
    f.a[j] = sc;
    mx = fmaxf(mx, sc);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < S; ++j) {
    f.a[j] = expf(f.a[j] - mx);
    sum += f.a[j];
  }
  const float inv = 1.f / sum;
  if (keep) {
    const f32x4* kr = reinterpret_cast<const f32x4*>(keep + (((size_t)cell * H + h) * S + i) * S);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 t = kr[c];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f.a[4 * c + u] *= inv;
        f.p[4 * c + u] = f.a[4 * c + u] * t[u];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < S; ++j) {
      f.a[j] *= inv;
      f.p[j] = f.a[j];
    }
  }
#pragma unroll
  for (int c = 0; c < HD; ++c) f.o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < S; ++j)
#pragma unroll
    for (int c = 0; c < HD; ++c) f.o[c] += f.p[j] * L.vs[h][j][c];
#pragma unroll
  for (int c = 0; c < HD; ++c) L.os[i][4 * h + c] = f.o[c];
  __syncthreads();
  float m = 0.f;
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    const int e = 4 * h + c;
    float t = W[OFF_BOUT + e];
#pragma unroll
    for (int kk = 0; kk < E; ++kk) t += L.os[i][kk] * W[OFF_WOUT + e * E + kk];
    f.s[c] = xown[c] + t;
    m += f.s[c];
  }
  const float mean = quad_sum(m) * (1.f / E);
  float vs = 0.f;
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    const float d = f.s[c] - mean;
    vs += d * d;
  }
  f.rstd = 1.f / sqrtf(quad_sum(vs) * (1.f / E) + LN_EPS);
#pragma unroll
  for (int c = 0; c < HD; ++c) f.xhat[c] = (f.s[c] - mean) * f.rstd;
}

__global__ void __launch_bounds__(256) attn_fwd_kernel(float* __restrict__ Y, const float* __restrict__ X,
                                                       const float* __restrict__ keep,
                                                       const float* __restrict__ params, int n_cells, int iters) {
  __shared__ __align__(16) float W[NPARAM];
  __shared__ __align__(16) WaveLdsFwd lds[4];
  for (int t = threadIdx.x; t < NPARAM; t += 256) W[t] = params[t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, h = lane >> 4;
  for (int it = 0; it < iters; ++it) {
    const int cell_raw = (it * gridDim.x + blockIdx.x) * 4 + wave;
    const bool valid = cell_raw < n_cells;
    const int cell = valid ? cell_raw : n_cells - 1;
    Fwd f;
    f32x4 xown;
    load_token(f, xown, X, cell, i, h);
    forward_core(f, xown, keep, W, lds[wave], cell, i, h);
    if (valid) {
      f32x4 y;
#pragma unroll
      for (int c = 0; c < HD; ++c) y[c] = f.xhat[c] * W[OFF_LNW + 4 * h + c] + W[OFF_LNB + 4 * h + c];
      *reinterpret_cast<f32x4*>(Y + ((size_t)cell * S + i) * E + 4 * h) = y;
    }
    __syncthreads();
  }
}

// The attention of a residual-attention block together with the steps of the token ResidualBlocks on either side of
// it (GAN_functions.py:54-68, :28-36, :142-145), which need no statistic of their own:
//   before  x = xin + bn2(t2)                - the block's last step (tok_bn_res_fwd in tok.hip): the column totals of
//           t2 are re-added from the per-workgroup partials exactly as there, x goes to Xres (the backward pass reads it)
//   after   t1 = y W1^T + b1 of the NEXT block - its first step (tok_lin_fwd) - with the column sums of t1 and t1^2
//           per workgroup of this launch (POST; stat_out[gridDim.x][32], float64, fixed order of additions)
// Two launches per block instead of four.
struct BlockFuse {
  const float* T2; const float* Xin; const double* stat_in; int n_stat_in;
  const float* gamma; const float* beta; float eps;
  float* mean_out; float* invstd_out; float* rmean; float* rvar; float momentum;
  float* Xres; int n_tok;
  float* T1; double* stat_out; const float* W1; const float* b1;
};

template <bool POST>
__global__ void __launch_bounds__(256) attn_block_fwd_kernel(float* __restrict__ Y, const float* __restrict__ keep,
                                                             const float* __restrict__ params, int n_cells, int iters,
                                                             BlockFuse bf) {
  __shared__ __align__(16) float W[NPARAM];
  __shared__ __align__(16) WaveLdsFwd lds[4];
  __shared__ double red[4 * 32], tot[32], wsum[4][32];
  __shared__ float cmean[E], cinv[E], cgam[E], cbet[E];
  __shared__ float w1[E * E], b1s[E];
  for (int t = threadIdx.x; t < NPARAM; t += 256) W[t] = params[t];
  if (POST) {
    w1[threadIdx.x] = bf.W1[threadIdx.x];
    if (threadIdx.x < E) b1s[threadIdx.x] = bf.b1[threadIdx.x];
    if (threadIdx.x < 128) wsum[threadIdx.x >> 5][threadIdx.x & 31] = 0.0;
  }
  // column totals of t2 from the partials: the additions of tok.hip's total_stats, in its order
  if (threadIdx.x < 128) {
    const int s = threadIdx.x & 31, q = threadIdx.x >> 5, nblk = bf.n_stat_in;
    const int per = (nblk + 3) >> 2;
    const int b0 = q * per, b1 = min(nblk, b0 + per);
    double a = 0.0;
    for (int b = b0; b < b1; b += 16) {
      double v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = bf.stat_in[(size_t)min(b + k, nblk - 1) * 32 + s];
#pragma unroll
      for (int k = 0; k < 16; ++k) a += b + k < b1 ? v[k] : 0.0;
    }
    red[q * 32 + s] = a;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int s = threadIdx.x;
    tot[s] = ((red[s] + red[32 + s]) + red[64 + s]) + red[96 + s];
  }
  __syncthreads();
  if (threadIdx.x < E) {      // as bn_fwd_cols in tok.hip
    const int c = threadIdx.x, n_tok = bf.n_tok;
    const double mean_d = tot[c] / n_tok;
    double var_d = tot[16 + c] / n_tok - mean_d * mean_d;
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d, invstd = (float)(1.0 / sqrt(var_d + (double)bf.eps));
    cmean[c] = mean;
    cinv[c] = invstd;
    cgam[c] = bf.gamma[c];
    cbet[c] = bf.beta[c];
    if (blockIdx.x == 0) {
      bf.mean_out[c] = mean;
      bf.invstd_out[c] = invstd;
      if (bf.rmean) bf.rmean[c] = (1.f - bf.momentum) * bf.rmean[c] + bf.momentum * mean;
      if (bf.rvar) {
        const float unb = (float)(var_d * ((double)n_tok / (double)(n_tok > 1 ? n_tok - 1 : 1)));
        bf.rvar[c] = (1.f - bf.momentum) * bf.rvar[c] + bf.momentum * unb;
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, h = lane >> 4;
  for (int it = 0; it < iters; ++it) {
    const int cell_raw = (it * gridDim.x + blockIdx.x) * 4 + wave;
    const bool valid = cell_raw < n_cells;
    const int cell = valid ? cell_raw : n_cells - 1;
    Fwd f;
    f32x4 xown;
    {
      const size_t row = ((size_t)cell * S + i) * E;
      const f32x4* tr = reinterpret_cast<const f32x4*>(bf.T2 + row);
      const f32x4* xr = reinterpret_cast<const f32x4*>(bf.Xin + row);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 t = tr[c], x = xr[c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = 4 * c + u;
          const float bn = (t[u] - cmean[e]) * cinv[e] * cgam[e] + cbet[e];
          f.x[e] = 1.0f * x[u] + 1.0f * bn;
        }
      }
      const f32x4 t = tr[h], x = xr[h];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = 4 * h + u;
        const float bn = (t[u] - cmean[e]) * cinv[e] * cgam[e] + cbet[e];
        xown[u] = 1.0f * x[u] + 1.0f * bn;
      }
      if (valid) *reinterpret_cast<f32x4*>(bf.Xres + row + 4 * h) = xown;
    }
    WaveLdsFwd& L = lds[wave];
    forward_core(f, xown, keep, W, L, cell, i, h);
    f32x4 y;
#pragma unroll
    for (int c = 0; c < HD; ++c) y[c] = f.xhat[c] * W[OFF_LNW + 4 * h + c] + W[OFF_LNB + 4 * h + c];
    if (valid) *reinterpret_cast<f32x4*>(Y + ((size_t)cell * S + i) * E + 4 * h) = y;
    if (POST) {
      __syncthreads();      // every lane has read the concatenated heads from L.os
#pragma unroll
      for (int c = 0; c < HD; ++c) L.os[i][4 * h + c] = y[c];
      __syncthreads();
      f32x4 t1;
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        const int o = 4 * h + c;
        float a = b1s[o];
#pragma unroll
        for (int e = 0; e < E; ++e) a += L.os[i][e] * w1[o * E + e];
        t1[c] = valid ? a : 0.f;
      }
      if (valid) *reinterpret_cast<f32x4*>(bf.T1 + ((size_t)cell * S + i) * E + 4 * h) = t1;
      // sums over the cell's 16 tokens (the lanes of one head group), then over the iterations of this wave
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        double s1 = (double)t1[c], s2 = (double)t1[c] * (double)t1[c];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          s1 += __shfl_xor(s1, off, 64);
          s2 += __shfl_xor(s2, off, 64);
        }
        if (i == 0) {
          wsum[wave][4 * h + c] += s1;
          wsum[wave][16 + 4 * h + c] += s2;
        }
      }
    }
    __syncthreads();
  }
  if (POST && threadIdx.x < 32) {
    const int s = threadIdx.x;
    bf.stat_out[(size_t)blockIdx.x * 32 + s] = ((wsum[0][s] + wsum[1][s]) + wsum[2][s]) + wsum[3][s];
  }
}

// Extra per-wave LDS of the backward.
struct WaveLdsBwd {
  float qs[H][S][HD];
  float dos[H][S][HD];
  float t1[H * HS];          // dscore, stored [h][j][i]
  float t2[H * HS];          // p,      stored [h][j][i]
  float xs[S][E];
  float cat[S][CATW];        // per token: dqkv(48) | ds(16) | dy*xhat(16) | dy(16)
};

// STATS: the first step of the token ResidualBlock's backward (tok_bn_bwd_stats in tok.hip) in the same launch: with
// g = dX, the gradient this kernel hands to that block's second BatchNorm, the per-workgroup sums of g and
// g * xhat2, xhat2 = (T2 - mean2) * invstd2, in float64 -> stat_out[gridDim.x][32] (fixed order of additions).
struct BwdStats {
  const float* T2; const float* mean2; const float* invstd2; double* stat_out;
};

template <bool STATS>
__global__ void __launch_bounds__(256) attn_bwd_kernel(float* __restrict__ dX, float* __restrict__ wpart,
                                                       const float* __restrict__ dY, const float* __restrict__ X,
                                                       const float* __restrict__ keep,
                                                       const float* __restrict__ params, int n_cells, int iters,
                                                       BwdStats bs) {
  __shared__ __align__(16) float W[NPARAM];
  __shared__ __align__(16) WaveLdsFwd ldsf[4];
  __shared__ __align__(16) WaveLdsBwd ldsb[4];
  __shared__ double wsum[4][32];
  __shared__ float smean[E], sinv[E];
  for (int t = threadIdx.x; t < NPARAM; t += 256) W[t] = params[t];
  if (STATS) {
    if (threadIdx.x < 128) wsum[threadIdx.x >> 5][threadIdx.x & 31] = 0.0;
    if (threadIdx.x < E) {
      smean[threadIdx.x] = bs.mean2[threadIdx.x];
      sinv[threadIdx.x] = bs.invstd2[threadIdx.x];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, h = lane >> 4;
  WaveLdsFwd& LF = ldsf[wave];
  WaveLdsBwd& LB = ldsb[wave];

  // parameter-gradient accumulators of this lane (see the maps below)
  float acc_win[12], acc_wout[4], acc_b[2];
#pragma unroll
  for (int u = 0; u < 12; ++u) acc_win[u] = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) acc_wout[u] = 0.f;
  acc_b[0] = acc_b[1] = 0.f;

  for (int it = 0; it < iters; ++it) {
    const int cell_raw = (it * gridDim.x + blockIdx.x) * 4 + wave;
    const bool valid = cell_raw < n_cells;
    const int cell = valid ? cell_raw : n_cells - 1;
    const float vscale = valid ? 1.f : 0.f;
    Fwd f;
    f32x4 xown;
    load_token(f, xown, X, cell, i, h);
    forward_core(f, xown, keep, W, LF, cell, i, h);

    // ---- LayerNorm backward (features 4h..4h+3 of token i) ----
    const f32x4 dy4 = *reinterpret_cast<const f32x4*>(dY + ((size_t)cell * S + i) * E + 4 * h);
    float dy[HD], dxh[HD], ds[HD];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      dy[c] = dy4[c] * vscale;
      dxh[c] = dy[c] * W[OFF_LNW + 4 * h + c];
      m1 += dxh[c];
      m2 += dxh[c] * f.xhat[c];
    }
    m1 = quad_sum(m1) * (1.f / E);
    m2 = quad_sum(m2) * (1.f / E);
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      ds[c] = f.rstd * (dxh[c] - m1 - f.xhat[c] * m2);
      LB.cat[i][48 + 4 * h + c] = ds[c];
      LB.cat[i][64 + 4 * h + c] = dy[c] * f.xhat[c];
      LB.cat[i][80 + 4 * h + c] = dy[c];
      LB.qs[h][i][c] = f.q[c];
    }
    if (h == 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) LB.xs[i][e] = f.x[e];
    }
    __syncthreads();
    // ---- output projection backward: do[i, 4h+c] = sum_e ds[i,e] * Wout[e, 4h+c] ----
    float dout[HD] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float dse = LB.cat[i][48 + e];
#pragma unroll
      for (int c = 0; c < HD; ++c) dout[c] += dse * W[OFF_WOUT + e * E + 4 * h + c];
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) LB.dos[h][i][c] = dout[c];
    // ---- attention backward, row i of head h ----
    float dsc[S];
    float tsum = 0.f;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      float dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) dp += dout[c] * LF.vs[h][j][c];
      // p = a * keep, so dA = dP * keep
      const float kp = keep ? keep[(((size_t)cell * H + h) * S + i) * S + j] : 1.f;
      dsc[j] = dp * kp;
      tsum += dsc[j] * f.a[j];
    }
    float dq[HD] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < S; ++j) {
      dsc[j] = f.a[j] * (dsc[j] - tsum);
      LB.t1[h * HS + j * TR + i] = dsc[j];
      LB.t2[h * HS + j * TR + i] = f.p[j];
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] += dsc[j] * LF.ks[h][j][c];
    }
    __syncthreads();
    // transposed sums: this lane now plays key/value index j = i
    float dk[HD] = {0.f, 0.f, 0.f, 0.f}, dv[HD] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(&LB.t1[h * HS + i * TR + 4 * u]);
      const f32x4 p4 = *reinterpret_cast<const f32x4*>(&LB.t2[h * HS + i * TR + 4 * u]);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int ii = 4 * u + w;
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          dk[c] += d4[w] * LB.qs[h][ii][c];
          dv[c] += p4[w] * LB.dos[h][ii][c];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      LB.cat[i][4 * h + c] = dq[c] * 0.5f;
      LB.cat[i][16 + 4 * h + c] = dk[c] * 0.5f;
      LB.cat[i][32 + 4 * h + c] = dv[c];
    }
    __syncthreads();
    // ---- input gradient: dx[i, 4h+c] = ds + sum_r dqkv[i, r] * Win[r, 4h+c] ----
    {
      f32x4 dx;
#pragma unroll
      for (int c = 0; c < HD; ++c) dx[c] = ds[c];
#pragma unroll
      for (int r = 0; r < 48; ++r) {
        const float g = LB.cat[i][r];
#pragma unroll
        for (int c = 0; c < HD; ++c) dx[c] += g * W[OFF_WIN + r * E + 4 * h + c];
      }
      if (valid) *reinterpret_cast<f32x4*>(dX + ((size_t)cell * S + i) * E + 4 * h) = dx;
      if (STATS) {
        const f32x4 t2 = *reinterpret_cast<const f32x4*>(bs.T2 + ((size_t)cell * S + i) * E + 4 * h);
#pragma unroll
        for (int c = 0; c < HD; ++c) {
          const float g = valid ? dx[c] : 0.f;
          const float xh = valid ? (t2[c] - smean[4 * h + c]) * sinv[4 * h + c] : 0.f;
          double s1 = (double)g, s2 = (double)g * (double)xh;
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) {      // over the cell's 16 tokens (the lanes of one head group)
            s1 += __shfl_xor(s1, off, 64);
            s2 += __shfl_xor(s2, off, 64);
          }
          if (i == 0) {
            wsum[wave][4 * h + c] += s1;
            wsum[wave][16 + 4 * h + c] += s2;
          }
        }
      }
    }
    // ---- parameter gradients (invalid cells contribute zeros: dy was zeroed) ----
    {
      const int c = lane & 15, rg = lane >> 4;            // dWin[rg*12 + u][c]
#pragma unroll
      for (int ii = 0; ii < S; ++ii) {
        const float xv = LB.xs[ii][c];
#pragma unroll
        for (int u = 0; u < 12; ++u) acc_win[u] += LB.cat[ii][rg * 12 + u] * xv;
      }
      // dWout[eg*4 + u][k], k = lane & 15, eg = lane >> 4
#pragma unroll
      for (int ii = 0; ii < S; ++ii) {
        const float ov = LF.os[ii][c];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc_wout[u] += LB.cat[ii][48 + rg * 4 + u] * ov;
      }
      // bias-like sums: column `lane` (dbin 0..47, dbout 48..63) and column 64+lane for lane < 32 (ln_w, ln_b)
#pragma unroll
      for (int ii = 0; ii < S; ++ii) {
        acc_b[0] += LB.cat[ii][lane];
        if (lane < 32) acc_b[1] += LB.cat[ii][64 + lane];
      }
    }
    __syncthreads();
  }

  // ---- combine the 4 waves and write this block's partial ----
  float* red = reinterpret_cast<float*>(ldsb);   // 4 * NPARAM floats needed; WaveLdsBwd is larger
  {
    const int c = lane & 15, rg = lane >> 4;
    float* mine = red + wave * NPARAM;
#pragma unroll
    for (int u = 0; u < 12; ++u) mine[OFF_WIN + (rg * 12 + u) * E + c] = acc_win[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) mine[OFF_WOUT + (rg * 4 + u) * E + c] = acc_wout[u];
    // column sums of `cat`: 0..47 dbin, 48..63 dbout, 64..79 dln_w, 80..95 dln_b
    mine[lane < 48 ? OFF_BIN + lane : OFF_BOUT + lane - 48] = acc_b[0];
    if (lane < 32) mine[lane < 16 ? OFF_LNW + lane : OFF_LNB + lane - 16] = acc_b[1];
  }
  __syncthreads();
  float* outp = wpart + (size_t)blockIdx.x * NPARAM;
  for (int t = threadIdx.x; t < NPARAM; t += 256) {
    outp[t] = red[t] + red[NPARAM + t] + red[2 * NPARAM + t] + red[3 * NPARAM + t];
  }
  if (STATS && threadIdx.x < 32) {
    const int s = threadIdx.x;
    bs.stat_out[(size_t)blockIdx.x * 32 + s] = ((wsum[0][s] + wsum[1][s]) + wsum[2][s]) + wsum[3][s];
  }
}

// 32 parameters per block; 8 lanes per parameter walk the per-block partials, then a fixed-order combine.
__global__ void __launch_bounds__(256) attn_param_reduce_kernel(float* dparams, const float* wpart, int n_blocks,
                                                                int accumulate, int np, long long dstride = 0,
                                                                long long wstride = 0) {
  __shared__ float lds[8][32];
  dparams += (size_t)blockIdx.y * dstride;      // blockIdx.y: one of several parameter blocks reduced in one launch
  wpart += (size_t)blockIdx.y * wstride;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + c;
  float s = 0.f;
  if (t < np)
    for (int b = r; b < n_blocks; b += 8) s += wpart[(size_t)b * np + t];
  lds[r][c] = s;
  __syncthreads();
  if (r == 0 && t < np) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += lds[i][c];
    dparams[t] = (accumulate ? dparams[t] : 0.f) + a;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Any other geometry: 16 tokens x EE features (EE = 4, 8, 16 or 32) and any head count that divides EE
// (GAN_functions.py:82,:103-113 take both from the config).  One wave per workgroup, one cell at a time;
// lane l <-> (token i = l & 15, part p = l >> 4): a lane owns EE/4 features of its token and the heads
// p, p+4, ... .  Everything the lanes exchange goes through LDS.  Parameter block layout as above with 16 -> EE:
//   Win[3EE][EE] | bin[3EE] | Wout[EE][EE] | bout[EE] | ln_w[EE] | ln_b[EE]
template <int EE>
struct Geo {
  static constexpr int F = EE / 4;
  static constexpr int O_WIN = 0, O_BIN = 3 * EE * EE, O_WOUT = O_BIN + 3 * EE, O_BOUT = O_WOUT + EE * EE,
                       O_LNW = O_BOUT + EE, O_LNB = O_LNW + EE, NP = O_LNB + EE;
  // one LDS row per token: dq | dk | dv | ds | dy*xhat | dy | x | o | 1  -- every parameter gradient is the sum
  // over tokens of a product of two columns of this row
  static constexpr int C_DS = 3 * EE, C_DYX = 4 * EE, C_DY = 5 * EE, C_X = 6 * EE, C_O = 7 * EE, C_ONE = 8 * EE,
                       RS = 8 * EE + 1;
  static constexpr int QS = 3 * EE + 1;
  static constexpr int MAXU = (NP + 63) / 64;
};

template <int EE>
struct GenLds {
  float W[Geo<EE>::NP];
  float row[S][Geo<EE>::RS];
  float qkv[S][Geo<EE>::QS];
  float dos[S][EE + 1];
  float t1[4][S][S + 1];     // dscore of the 4 heads of a round, [i][j]
  float t2[4][S][S + 1];     // p
};

// softmax row i of head h: a = probabilities, kp = keep-mask row (1 without dropout), pk = a * kp
template <int EE>
__device__ __forceinline__ void gen_attn_row(const GenLds<EE>& L, const float* __restrict__ keep, int cell, int H, int HD,
                                             float scale, int h, int i, float (&a)[S], float (&kp)[S], float (&pk)[S]) {
  const int base = h * HD;
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < S; ++j) {
    float sc = 0.f;
    for (int c = 0; c < HD; ++c) sc += (L.qkv[i][base + c] * scale) * L.qkv[j][EE + base + c];
    a[j] = sc;
    mx = fmaxf(mx, sc);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < S; ++j) {
    a[j] = expf(a[j] - mx);
    sum += a[j];
  }
  const float inv = 1.f / sum;
  if (keep) {
    const f32x4* kr = reinterpret_cast<const f32x4*>(keep + (((size_t)cell * H + h) * S + i) * S);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 t = kr[c];
#pragma unroll
      for (int u = 0; u < 4; ++u) kp[4 * c + u] = t[u];
    }
  } else {
#pragma unroll
    for (int j = 0; j < S; ++j) kp[j] = 1.f;
  }
#pragma unroll
  for (int j = 0; j < S; ++j) {
    a[j] *= inv;
    pk[j] = a[j] * kp[j];
  }
}

// Forward of one cell; leaves x, o in L.row, q|k|v in L.qkv and returns this lane's normalised features.
template <int EE>
__device__ __forceinline__ void gen_forward(GenLds<EE>& L, const float* __restrict__ X, const float* __restrict__ keep,
                                            int cell, int H, int HD, float scale, int lane,
                                            float (&xhat)[Geo<EE>::F], float& rstd) {
  using G = Geo<EE>;
  constexpr int F = G::F;
  const int i = lane & 15, p = lane >> 4;
  const float* xr = X + ((size_t)cell * S + i) * EE + p * F;
#pragma unroll
  for (int f = 0; f < F; ++f) L.row[i][G::C_X + p * F + f] = xr[f];
  __syncthreads();
  for (int u = 0; u < 3 * F; ++u) {
    const int r = p * 3 * F + u;
    float acc = L.W[G::O_BIN + r];
#pragma unroll
    for (int e = 0; e < EE; ++e) acc += L.row[i][G::C_X + e] * L.W[G::O_WIN + r * EE + e];
    L.qkv[i][r] = acc;
  }
  __syncthreads();
  for (int h = p; h < H; h += 4) {
    float a[S], kp[S], pk[S];
    gen_attn_row<EE>(L, keep, cell, H, HD, scale, h, i, a, kp, pk);
    const int base = h * HD;
    for (int c = 0; c < HD; ++c) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < S; ++j) o += pk[j] * L.qkv[j][2 * EE + base + c];
      L.row[i][G::C_O + base + c] = o;
    }
  }
  __syncthreads();
  float s[F];
  float m = 0.f;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const int e = p * F + f;
    float t = L.W[G::O_BOUT + e];
#pragma unroll
    for (int k = 0; k < EE; ++k) t += L.row[i][G::C_O + k] * L.W[G::O_WOUT + e * EE + k];
    s[f] = L.row[i][G::C_X + e] + t;
    m += s[f];
  }
  const float mean = quad_sum(m) * (1.f / EE);
  float vs = 0.f;
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const float d = s[f] - mean;
    vs += d * d;
  }
  rstd = 1.f / sqrtf(quad_sum(vs) * (1.f / EE) + LN_EPS);
#pragma unroll
  for (int f = 0; f < F; ++f) xhat[f] = (s[f] - mean) * rstd;
}

template <int EE>
__global__ void __launch_bounds__(64) attn_gen_fwd_kernel(float* __restrict__ Y, const float* __restrict__ X,
                                                          const float* __restrict__ keep,
                                                          const float* __restrict__ params, int n_cells, int iters,
                                                          int H, float scale) {
  using G = Geo<EE>;
  __shared__ __align__(16) GenLds<EE> L;
  for (int t = threadIdx.x; t < G::NP; t += 64) L.W[t] = params[t];
  __syncthreads();
  const int lane = threadIdx.x, i = lane & 15, p = lane >> 4;
  const int HD = EE / H;
  for (int it = 0; it < iters; ++it) {
    const int cell_raw = it * gridDim.x + blockIdx.x;
    const bool valid = cell_raw < n_cells;
    const int cell = valid ? cell_raw : n_cells - 1;
    float xhat[G::F], rstd;
    gen_forward<EE>(L, X, keep, cell, H, HD, scale, lane, xhat, rstd);
    if (valid) {
      float* yr = Y + ((size_t)cell * S + i) * EE + p * G::F;
#pragma unroll
      for (int f = 0; f < G::F; ++f) yr[f] = xhat[f] * L.W[G::O_LNW + p * G::F + f] + L.W[G::O_LNB + p * G::F + f];
    }
    __syncthreads();
  }
}

template <int EE>
__global__ void __launch_bounds__(64) attn_gen_bwd_kernel(float* __restrict__ dX, float* __restrict__ wpart,
                                                          const float* __restrict__ dY, const float* __restrict__ X,
                                                          const float* __restrict__ keep,
                                                          const float* __restrict__ params, int n_cells, int iters,
                                                          int H, float scale) {
  using G = Geo<EE>;
  constexpr int F = G::F;
  __shared__ __align__(16) GenLds<EE> L;
  for (int t = threadIdx.x; t < G::NP; t += 64) L.W[t] = params[t];
  const int lane = threadIdx.x, i = lane & 15, p = lane >> 4;
  const int HD = EE / H;
  if (p == 0) L.row[i][G::C_ONE] = 1.f;
  // the two row columns whose product, summed over tokens, is the gradient of parameter lane + 64u
  int colA[G::MAXU], colB[G::MAXU];
  float acc[G::MAXU];
#pragma unroll
  for (int u = 0; u < G::MAXU; ++u) {
    const int q = lane + 64 * u;
    int a = 0, b = G::C_ONE;
    if (q < G::O_BIN) { a = q / EE; b = G::C_X + q % EE; }
    else if (q < G::O_WOUT) { a = q - G::O_BIN; }
    else if (q < G::O_BOUT) { a = G::C_DS + (q - G::O_WOUT) / EE; b = G::C_O + (q - G::O_WOUT) % EE; }
    else if (q < G::O_LNW) { a = G::C_DS + q - G::O_BOUT; }
    else if (q < G::O_LNB) { a = G::C_DYX + q - G::O_LNW; }
    else if (q < G::NP) { a = G::C_DY + q - G::O_LNB; }
    colA[u] = a; colB[u] = b; acc[u] = 0.f;
  }
  __syncthreads();

  for (int it = 0; it < iters; ++it) {
    const int cell_raw = it * gridDim.x + blockIdx.x;
    const bool valid = cell_raw < n_cells;
    const int cell = valid ? cell_raw : n_cells - 1;
    const float vscale = valid ? 1.f : 0.f;
    float xhat[F], rstd;
    gen_forward<EE>(L, X, keep, cell, H, HD, scale, lane, xhat, rstd);

    // ---- LayerNorm backward (this lane's features of token i) ----
    const float* dyr = dY + ((size_t)cell * S + i) * EE + p * F;
    float ds[F], dy[F], dxh[F];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      dy[f] = dyr[f] * vscale;
      dxh[f] = dy[f] * L.W[G::O_LNW + p * F + f];
      m1 += dxh[f];
      m2 += dxh[f] * xhat[f];
    }
    m1 = quad_sum(m1) * (1.f / EE);
    m2 = quad_sum(m2) * (1.f / EE);
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const int e = p * F + f;
      ds[f] = rstd * (dxh[f] - m1 - xhat[f] * m2);
      L.row[i][G::C_DS + e] = ds[f];
      L.row[i][G::C_DYX + e] = dy[f] * xhat[f];
      L.row[i][G::C_DY + e] = dy[f];
    }
    __syncthreads();
    // ---- output projection backward: do[i][k] = sum_e ds[i][e] * Wout[e][k] ----
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const int k = p * F + f;
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < EE; ++e) t += L.row[i][G::C_DS + e] * L.W[G::O_WOUT + e * EE + k];
      L.dos[i][k] = t;
    }
    __syncthreads();
    // ---- attention backward, four heads per round ----
    for (int h0 = 0; h0 < H; h0 += 4) {
      const int h = h0 + p;
      const bool act = h < H;
      const int base = h * HD;
      if (act) {
        float a[S], kp[S], pk[S];
        gen_attn_row<EE>(L, keep, cell, H, HD, scale, h, i, a, kp, pk);
        float dsc[S];
        float tsum = 0.f;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          float dp = 0.f;
          for (int c = 0; c < HD; ++c) dp += L.dos[i][base + c] * L.qkv[j][2 * EE + base + c];
          dsc[j] = dp * kp[j];
          tsum += dsc[j] * a[j];
        }
#pragma unroll
        for (int j = 0; j < S; ++j) {
          dsc[j] = a[j] * (dsc[j] - tsum);
          L.t1[p][i][j] = dsc[j];
          L.t2[p][i][j] = pk[j];
        }
        for (int c = 0; c < HD; ++c) {
          float dq = 0.f;
#pragma unroll
          for (int j = 0; j < S; ++j) dq += dsc[j] * L.qkv[j][EE + base + c];
          L.row[i][base + c] = dq * scale;
        }
      }
      __syncthreads();
      if (act) {      // this lane now plays key/value index j = i
        for (int c = 0; c < HD; ++c) {
          float dk = 0.f, dv = 0.f;
#pragma unroll
          for (int ii = 0; ii < S; ++ii) {
            dk += L.t1[p][ii][i] * L.qkv[ii][base + c];
            dv += L.t2[p][ii][i] * L.dos[ii][base + c];
          }
          L.row[i][EE + base + c] = dk * scale;
          L.row[i][2 * EE + base + c] = dv;
        }
      }
      __syncthreads();
    }
    // ---- input gradient ----
    {
      float* dxr = dX + ((size_t)cell * S + i) * EE + p * F;
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const int e = p * F + f;
        float t = ds[f];
        for (int r = 0; r < 3 * EE; ++r) t += L.row[i][r] * L.W[G::O_WIN + r * EE + e];
        if (valid) dxr[f] = t;
      }
    }
    // ---- parameter gradients (a cell beyond the end contributes zeros: its dy was zeroed) ----
#pragma unroll
    for (int u = 0; u < G::MAXU; ++u) {
      float t = 0.f;
#pragma unroll
      for (int ii = 0; ii < S; ++ii) t += L.row[ii][colA[u]] * L.row[ii][colB[u]];
      acc[u] += t;
    }
    __syncthreads();
  }
  float* outp = wpart + (size_t)blockIdx.x * G::NP;
#pragma unroll
  for (int u = 0; u < G::MAXU; ++u) {
    const int q = lane + 64 * u;
    if (q < G::NP) outp[q] = acc[u];
  }
}

inline bool geometry_ok(int E_, int heads) {
  return (E_ == 4 || E_ == 8 || E_ == 16 || E_ == 32) && heads >= 1 && heads <= E_ && E_ % heads == 0;
}

inline bool fused_16x4(int E_, int heads) {
  static const bool force_generic = getenv("CGRNA_ATTN_GENERIC") != nullptr;
  return E_ == E && heads == H && !force_generic;
}

inline int gen_blocks(int n_cells) { return n_cells < 512 ? (n_cells < 1 ? 1 : n_cells) : 512; }

template <int EE>
void launch_gen_fwd(float* Y, const float* X, const float* keep, const float* params, int n_cells, int heads,
                    hipStream_t st) {
  const int nb = gen_blocks(n_cells), iters = (n_cells + nb - 1) / nb;
  const float scale = (float)sqrt(1.0 / (double)(EE / heads));
  hipLaunchKernelGGL(attn_gen_fwd_kernel<EE>, dim3(nb), dim3(64), 0, st, Y, X, keep, params, n_cells, iters, heads,
                     scale);
}

template <int EE>
void launch_gen_bwd(float* dX, float* dparams, float* wpart, const float* dY, const float* X, const float* keep,
                    const float* params, int n_cells, int heads, int accumulate, hipStream_t st) {
  const int nb = gen_blocks(n_cells), iters = (n_cells + nb - 1) / nb;
  const float scale = (float)sqrt(1.0 / (double)(EE / heads));
  hipLaunchKernelGGL(attn_gen_bwd_kernel<EE>, dim3(nb), dim3(64), 0, st, dX, wpart, dY, X, keep, params, n_cells,
                     iters, heads, scale);
  hipLaunchKernelGGL(attn_param_reduce_kernel, dim3((Geo<EE>::NP + 31) / 32), dim3(256), 0, st, dparams, wpart, nb,
                     accumulate, Geo<EE>::NP);
}

}  // namespace

extern "C" int cg_attn_bwd_blocks(int n_cells) {
  int nb = (n_cells + 3) / 4;
  if (nb > 256) nb = 256;
  if (nb < 1) nb = 1;
  return nb;
}

extern "C" long long cg_attn_bwd_ws(int n_cells, int E_, int heads) {
  if (!geometry_ok(E_, heads)) return -1;
  if (fused_16x4(E_, heads)) return (long long)cg_attn_bwd_blocks(n_cells) * NPARAM;
  return (long long)gen_blocks(n_cells) * (4 * E_ * E_ + 6 * E_);
}

extern "C" int cg_attn_ln_fwd(float* Y, const float* X, const float* keep, const float* params, int n_cells, int E_,
                              int heads, void* stream) {
  if (!Y || !X || !params || n_cells <= 0) return CG_ERR_BAD_ARG;
  if (!geometry_ok(E_, heads)) return CG_ERR_UNSUPPORTED;
  if ((((uintptr_t)Y | (uintptr_t)X | (uintptr_t)keep) % 16) != 0) return CG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (fused_16x4(E_, heads)) {
    int nb = (n_cells + 3) / 4;
    if (nb > 1024) nb = 1024;
    const int iters = (n_cells + nb * 4 - 1) / (nb * 4);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(nb), dim3(256), 0, st, Y, X, keep, params, n_cells, iters);
  } else if (E_ == 4) {
    launch_gen_fwd<4>(Y, X, keep, params, n_cells, heads, st);
  } else if (E_ == 8) {
    launch_gen_fwd<8>(Y, X, keep, params, n_cells, heads, st);
  } else if (E_ == 16) {
    launch_gen_fwd<16>(Y, X, keep, params, n_cells, heads, st);
  } else {
    launch_gen_fwd<32>(Y, X, keep, params, n_cells, heads, st);
  }
  return cg_launch_status();
}

extern "C" int cg_attn_block_bwd(float* dX, float* dparams, float* wpart, const float* dY, const float* X,
                                 const float* keep, const float* params, int n_cells, int E_, int heads, int accumulate,
                                 const float* T2, const float* mean2, const float* invstd2, double* stat_out,
                                 int reduce_now, void* stream) {
  if (!dX || !wpart || !dY || !X || !params || !T2 || !mean2 || !invstd2 || !stat_out || n_cells <= 0)
    return CG_ERR_BAD_ARG;
  if (reduce_now && !dparams) return CG_ERR_BAD_ARG;
  if (!fused_16x4(E_, heads)) return CG_ERR_UNSUPPORTED;
  if ((((uintptr_t)dX | (uintptr_t)dY | (uintptr_t)X | (uintptr_t)keep | (uintptr_t)T2) % 16) != 0) return CG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nb = cg_attn_bwd_blocks(n_cells);
  const int iters = (n_cells + nb * 4 - 1) / (nb * 4);
  hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(nb), dim3(256), 0, st, dX, wpart, dY, X, keep, params, n_cells, iters,
                     BwdStats{T2, mean2, invstd2, stat_out});
  if (reduce_now)
    hipLaunchKernelGGL(attn_param_reduce_kernel, dim3((NPARAM + 31) / 32), dim3(256), 0, st, dparams, wpart, nb,
                       accumulate, NPARAM, 0LL, 0LL);
  return cg_launch_status();
}

extern "C" int cg_attn_param_reduce_blocks(float* dparams, long long dstride, const float* wpart, long long wstride,
                                           int n_param_blocks, int n_cells, int accumulate, void* stream) {
  if (!dparams || !wpart || n_param_blocks <= 0 || n_cells <= 0) return CG_ERR_BAD_ARG;
  hipLaunchKernelGGL(attn_param_reduce_kernel, dim3((NPARAM + 31) / 32, n_param_blocks), dim3(256), 0, (hipStream_t)stream,
                     dparams, wpart, cg_attn_bwd_blocks(n_cells), accumulate, NPARAM, dstride, wstride);
  return cg_launch_status();
}

extern "C" int cg_attn_block_blocks(int n_cells) {
  int nb = (n_cells + 3) / 4;
  if (nb > 1024) nb = 1024;
  return nb < 1 ? 1 : nb;
}

extern "C" int cg_attn_block_fwd(float* Y, float* Xres, const float* T2, const float* Xin, const double* stat_in,
                                 int n_stat_in, const float* gamma, const float* beta, float eps, float* mean_out,
                                 float* invstd_out, float* running_mean, float* running_var, float momentum,
                                 const float* keep, const float* params, int n_cells, int E_, int heads,
                                 float* T1_next, double* stat_next, const float* W1_next, const float* b1_next,
                                 void* stream) {
  if (!Y || !Xres || !T2 || !Xin || !stat_in || !gamma || !beta || !mean_out || !invstd_out || !params ||
      n_cells <= 0 || n_stat_in <= 0)
    return CG_ERR_BAD_ARG;
  if (!fused_16x4(E_, heads)) return CG_ERR_UNSUPPORTED;
  if ((((uintptr_t)Y | (uintptr_t)Xres | (uintptr_t)T2 | (uintptr_t)Xin | (uintptr_t)keep | (uintptr_t)T1_next) % 16) != 0)
    return CG_ERR_BAD_ARG;
  if (T1_next && (!stat_next || !W1_next || !b1_next)) return CG_ERR_BAD_ARG;
  const int nb = cg_attn_block_blocks(n_cells);
  const int iters = (n_cells + nb * 4 - 1) / (nb * 4);
  BlockFuse bf{T2, Xin, stat_in, n_stat_in, gamma, beta, eps, mean_out, invstd_out, running_mean, running_var, momentum,
               Xres, n_cells * S, T1_next, stat_next, W1_next, b1_next};
  hipStream_t st = (hipStream_t)stream;
  if (T1_next) hipLaunchKernelGGL(attn_block_fwd_kernel<true>, dim3(nb), dim3(256), 0, st, Y, keep, params, n_cells, iters, bf);
  else hipLaunchKernelGGL(attn_block_fwd_kernel<false>, dim3(nb), dim3(256), 0, st, Y, keep, params, n_cells, iters, bf);
  return cg_launch_status();
}

extern "C" int cg_attn_ln_bwd(float* dX, float* dparams, float* wpart, const float* dY, const float* X,
                              const float* keep, const float* params, int n_cells, int E_, int heads, int accumulate,
                              void* stream) {
  if (!dX || !dparams || !wpart || !dY || !X || !params || n_cells <= 0) return CG_ERR_BAD_ARG;
  if (!geometry_ok(E_, heads)) return CG_ERR_UNSUPPORTED;
  if ((((uintptr_t)dX | (uintptr_t)dY | (uintptr_t)X | (uintptr_t)keep) % 16) != 0) return CG_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (fused_16x4(E_, heads)) {
    const int nb = cg_attn_bwd_blocks(n_cells);
    const int iters = (n_cells + nb * 4 - 1) / (nb * 4);
    hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(nb), dim3(256), 0, st, dX, wpart, dY, X, keep, params, n_cells, iters,
                       BwdStats{nullptr, nullptr, nullptr, nullptr});
    hipLaunchKernelGGL(attn_param_reduce_kernel, dim3((NPARAM + 31) / 32), dim3(256), 0, st, dparams, wpart, nb,
                       accumulate, NPARAM);
  } else if (E_ == 4) {
    launch_gen_bwd<4>(dX, dparams, wpart, dY, X, keep, params, n_cells, heads, accumulate, st);
  } else if (E_ == 8) {
    launch_gen_bwd<8>(dX, dparams, wpart, dY, X, keep, params, n_cells, heads, accumulate, st);
  } else if (E_ == 16) {
    launch_gen_bwd<16>(dX, dparams, wpart, dY, X, keep, params, n_cells, heads, accumulate, st);
  } else {
    launch_gen_bwd<32>(dX, dparams, wpart, dY, X, keep, params, n_cells, heads, accumulate, st);
  }
  return cg_launch_status();
}
