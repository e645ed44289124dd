// Shared helpers for the gfx950 kernels of libcgrna.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cgrna.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define CG_WAVE 64

static inline int cg_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CG_OK : (int)e;
}

// float32 -> bf16 bits, round to nearest even; NaN stays NaN.
__device__ __forceinline__ u16 f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ float bf2f(u16 h) {
  return __builtin_bit_cast(float, ((uint32_t)h) << 16);
}

// Sum over the 64 lanes of a wave; every lane gets the result.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blocks of up to 1024 threads; result valid in thread 0
// (and broadcast to all through LDS when `bcast`).
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* lds /* >= 17 entries */, bool bcast = false) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) lds[w] = v;
  __syncthreads();
  if (w == 0) {
    T s = lane < nw ? lds[lane] : (T)0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) lds[16] = s;
  }
  if (bcast) {
    __syncthreads();
    v = lds[16];
    return v;
  }
  return lds[16];
}
