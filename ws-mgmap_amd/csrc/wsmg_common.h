// Shared helpers for the gfx950 kernels of libwsmgmap.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wsmgmap.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WSMG_WAVE 64

// after a kernel launch: report launch-time errors through the C ABI's int return
#define WSMG_RETURN_LAUNCH()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    return e__ == hipSuccess ? 0 : (int)e__;      \
  } while (0)

static inline hipStream_t wsmg_s(wsmg_stream_t s) { return (hipStream_t)s; }

static inline int64_t wsmg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
