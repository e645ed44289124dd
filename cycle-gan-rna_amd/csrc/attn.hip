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

__device__ __forceinline__ void forward_core(Fwd& f, const float* __restrict__ X, const float* __restrict__ keep,
                                             const float* W /* LDS params */, WaveLdsFwd& L, int cell, int i, int h) {
  const f32x4* xr = reinterpret_cast<const f32x4*>(X + ((size_t)cell * S + i) * E);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 t = xr[c];
    f.x[4 * c] = t[0]; f.x[4 * c + 1] = t[1]; f.x[4 * c + 2] = t[2]; f.x[4 * c + 3] = t[3];
  }
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
    f.s[c] = f.x[e] + t;
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
    forward_core(f, X, keep, W, lds[wave], cell, i, h);
    if (valid) {
      f32x4 y;
#pragma unroll
      for (int c = 0; c < HD; ++c) y[c] = f.xhat[c] * W[OFF_LNW + 4 * h + c] + W[OFF_LNB + 4 * h + c];
      *reinterpret_cast<f32x4*>(Y + ((size_t)cell * S + i) * E + 4 * h) = y;
    }
    __syncthreads();
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

__global__ void __launch_bounds__(256) attn_bwd_kernel(float* __restrict__ dX, float* __restrict__ wpart,
                                                       const float* __restrict__ dY, const float* __restrict__ X,
                                                       const float* __restrict__ keep,
                                                       const float* __restrict__ params, int n_cells, int iters) {
  __shared__ __align__(16) float W[NPARAM];
  __shared__ __align__(16) WaveLdsFwd ldsf[4];
  __shared__ __align__(16) WaveLdsBwd ldsb[4];
  for (int t = threadIdx.x; t < NPARAM; t += 256) W[t] = params[t];
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
    forward_core(f, X, keep, W, LF, cell, i, h);

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
}

// 32 parameters per block; 8 lanes per parameter walk the per-block partials, then a fixed-order combine.
__global__ void __launch_bounds__(256) attn_param_reduce_kernel(float* dparams, const float* wpart, int n_blocks,
                                                                int accumulate) {
  __shared__ float lds[8][32];
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + c;
  float s = 0.f;
  if (t < NPARAM)
    for (int b = r; b < n_blocks; b += 8) s += wpart[(size_t)b * NPARAM + t];
  lds[r][c] = s;
  __syncthreads();
  if (r == 0 && t < NPARAM) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a += lds[i][c];
    dparams[t] = (accumulate ? dparams[t] : 0.f) + a;
  }
}

}  // namespace

extern "C" int cg_attn_bwd_blocks(int n_cells) {
  int nb = (n_cells + 3) / 4;
  if (nb > 256) nb = 256;
  if (nb < 1) nb = 1;
  return nb;
}

extern "C" int cg_attn_ln_fwd(float* Y, const float* X, const float* keep, const float* params, int n_cells, int E_,
                              int heads, void* stream) {
  if (!Y || !X || !params || n_cells <= 0) return CG_ERR_BAD_ARG;
  if (E_ != E || heads != H) return CG_ERR_UNSUPPORTED;
  if ((((uintptr_t)Y | (uintptr_t)X | (uintptr_t)keep) % 16) != 0) return CG_ERR_BAD_ARG;
  int nb = (n_cells + 3) / 4;
  if (nb > 1024) nb = 1024;
  const int iters = (n_cells + nb * 4 - 1) / (nb * 4);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, Y, X, keep, params, n_cells, iters);
  return cg_launch_status();
}

extern "C" int cg_attn_ln_bwd(float* dX, float* dparams, float* wpart, const float* dY, const float* X,
                              const float* keep, const float* params, int n_cells, int E_, int heads, int accumulate,
                              void* stream) {
  if (!dX || !dparams || !wpart || !dY || !X || !params || n_cells <= 0) return CG_ERR_BAD_ARG;
  if (E_ != E || heads != H) return CG_ERR_UNSUPPORTED;
  if ((((uintptr_t)dX | (uintptr_t)dY | (uintptr_t)X | (uintptr_t)keep) % 16) != 0) return CG_ERR_BAD_ARG;
  const int nb = cg_attn_bwd_blocks(n_cells);
  const int iters = (n_cells + nb * 4 - 1) / (nb * 4);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dX, wpart, dY, X, keep, params,
                     n_cells, iters);
  hipLaunchKernelGGL(attn_param_reduce_kernel, dim3((NPARAM + 31) / 32), dim3(256), 0, (hipStream_t)stream, dparams,
                     wpart, nb, accumulate);
  return cg_launch_status();
}
