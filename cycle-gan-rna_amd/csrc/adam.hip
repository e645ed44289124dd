// Fused multi-tensor Adam (amsgrad) over one flat parameter arena.
// HBM-bound: reads p, g, m, v, vmax and writes p, m, v, vmax = 36 B per
// parameter (+2 B when the bf16 shadow is written); 16 B per lane per access.
#include "common.h"

namespace {

struct AdamScalars {
  float step_size;   // lr / (1 - b1^t)
  float bc2_sqrt;    // sqrt(1 - b2^t)
};

__device__ __forceinline__ AdamScalars adam_scalars(const double* lr_and_step, float b1, float b2) {
  const double lr = lr_and_step[0], t = lr_and_step[1];
  AdamScalars s;
  s.step_size = (float)(lr / (1.0 - pow((double)b1, t)));
  s.bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  return s;
}

template <bool AMS>
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float& vm,
                                         const AdamScalars& s, float b1, float b2, float eps) {
  // exp_avg.lerp_(grad, 1-b1): torch's lerp switches formula at weight 0.5
  const float w = 1.f - b1;
  m = (w < 0.5f) ? m + w * (g - m) : g - (g - m) * (1.f - w);
  v = v * b2 + (1.f - b2) * g * g;
  if (AMS) vm = fmaxf(vm, v);
  const float denom = sqrtf(AMS ? vm : v) / s.bc2_sqrt + eps;
  p = p + (-s.step_size * m) / denom;
}

template <bool AMS>
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ vmax, size_t n,
                                                   const double* __restrict__ lr_and_step, float b1, float b2,
                                                   float eps, u16* __restrict__ pbf) {
  const AdamScalars s = adam_scalars(lr_and_step, b1, b2);
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    f32x4 xx = {0.f, 0.f, 0.f, 0.f};
    if (AMS) xx = reinterpret_cast<f32x4*>(vmax)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], me = mm[e], ve = vv[e], xe = xx[e];
      adam_one<AMS>(pe, gg[e], me, ve, xe, s, b1, b2, eps);
      pp[e] = pe; mm[e] = me; vv[e] = ve; xx[e] = xe;
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (AMS) reinterpret_cast<f32x4*>(vmax)[i] = xx;
    if (pbf) {
      ushort4 o;
      o.x = f2bf(pp[0]); o.y = f2bf(pp[1]); o.z = f2bf(pp[2]); o.w = f2bf(pp[3]);
      reinterpret_cast<ushort4*>(pbf)[i] = o;
    }
  }
  // tail (n not a multiple of 4)
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pp = p[i], mm = m[i], vv = v[i], xx = AMS ? vmax[i] : 0.f;
    adam_one<AMS>(pp, g[i], mm, vv, xx, s, b1, b2, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (AMS) vmax[i] = xx;
    if (pbf) pbf[i] = f2bf(pp);
  }
}

}  // namespace

extern "C" int cg_adam_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, size_t n,
                               const double* lr_and_step, float beta1, float beta2, float eps,
                               void* p_bf16, void* stream) {
  if (!p || !g || !m || !v || !lr_and_step) return CG_ERR_BAD_ARG;
  if (n == 0) return CG_OK;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax) % 16) != 0) return CG_ERR_BAD_ARG;
  if (p_bf16 && ((uintptr_t)p_bf16 % 8) != 0) return CG_ERR_BAD_ARG;
  const size_t n4 = (n + 3) / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks == 0) blocks = 1;
  if (vmax)
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax,
                       n, lr_and_step, beta1, beta2, eps, (u16*)p_bf16);
  else      // torch.optim.Adam's default: no running maximum
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax,
                       n, lr_and_step, beta1, beta2, eps, (u16*)p_bf16);
  return cg_launch_status();
}
