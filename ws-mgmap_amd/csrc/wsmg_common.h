// Shared helpers for the gfx950 kernels of libwsmgmap.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "wsmgmap.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16_t;
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#define WSMG_WAVE 64

// storage-type helpers: activations are float32 or bf16 in HBM, arithmetic is float32
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = (bf16_t)v; }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
  u16x4 u = *reinterpret_cast<const u16x4*>(p);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = __uint_as_float(((unsigned)u[j]) << 16);
  return o;
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
  u16x4 u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf16_t b = (bf16_t)v[j];
    u[j] = __builtin_bit_cast(unsigned short, b);
  }
  *reinterpret_cast<u16x4*>(p) = u;
}

// after a kernel launch: report launch-time errors through the C ABI's int return
#define WSMG_RETURN_LAUNCH()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    return e__ == hipSuccess ? 0 : (int)e__;      \
  } while (0)

static inline hipStream_t wsmg_s(wsmg_stream_t s) { return (hipStream_t)s; }

// Tuning / A-B switches of the library (WSMG_* environment variables of the launchers: tile choices, kernel variants).  Every one
// is read ONCE per process, at the first launch that asks — no launch after that touches the environment.  WSMG_TUNE is the
// only way the library reads it.
static inline int wsmg_tune_read(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#define WSMG_TUNE(name, dflt) ([]() -> int { static const int v_ = wsmg_tune_read(name, dflt); return v_; }())

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

// wsmg_conv_win.hip: direct convolution with an LDS-resident input window (64 -> 64 channels, k8 s2 p3); WSMG_EINVAL for
// any other shape
int wsmg_conv_win_fwd_bf16(const void* x, const void* w_ohwi, const float* bias, void* y, int relu, double* stats, int nslab, int B,
                           int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, hipStream_t s);
// wsmg_conv_win_wgrad.hip: weight gradient of the same layer out of an LDS-resident input window; WSMG_EINVAL for any other shape
int wsmg_conv_win_wgrad_bf16(const void* x, const void* dy, float* dw_ohwi, long long slab_floats, int B, int H, int W, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, int OH, int OW, hipStream_t s);
int wsmg_conv_win_wgrad_splits(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW);
// wsmg_conv_s2_wgrad.hip: weight gradient of the k5 s2 (64 -> 128 at 50 x 50) and k7 s2 (256 -> 64 at 24 x 24) layers out of an
// LDS-resident input window; WSMG_EINVAL / 0 splits for any other shape
int wsmg_conv_s2_wgrad_bf16(const void* x, const void* dy, float* dw_ohwi, long long slab_floats, int B, int H, int W, int Cin, int Cout,
                            int KH, int KW, int stride, int pad, int OH, int OW, hipStream_t s);
int wsmg_conv_s2_wgrad_splits(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW);
// wsmg_conv_win3.hip: 3 x 3 / stride 1 / pad 1 out of a zero-padded LDS pixel window (forward / backward-data), N % 128 == 0,
// Kc % 32 == 0, mt = 512 or 256 pixels per workgroup, mixed != 0: the last partial round as half-size tiles; WSMG_EINVAL otherwise
// (round 6) relu_z [pixels][N] or null: ReLU outputs the output tile is masked with (wsmg_relu_mask.h); dst_ld: pixel pitch of dst in
// elements (0 = N); dst2 / split_c: output channels >= split_c go to the second tensor (null: none)
int wsmg_conv_win3_bf16(int bwd, const void* src, const void* wt, const float* bias, void* dst, int relu, double* stats, int nslab,
                        int B, int H, int W, int Kc, int N, int mt, int mixed, const void* relu_z, int dst_ld, void* dst2, int split_c,
                        hipStream_t s);
// wsmg_conv_win3_k32.hip (round 6): the same convolution over a 32-channel reduction axis (Kc == 32, N % 32 == 0, plain output): weights
// resident in LDS, one barrier per 256-pixel tile, three workgroups per CU; WSMG_EINVAL otherwise
int wsmg_conv_win3_k32_bf16(int bwd, const void* src, const void* wt, const float* bias, void* dst, int relu, double* stats, int nslab,
                            int B, int H, int W, int N, int dst_ld, hipStream_t s);
// wsmg_convt_k4s2.hip (round 6): ConvTranspose2d(64 -> 32, k4, s2, p1) forward — both column parities of a row parity per workgroup,
// weights resident in LDS, full-line stores; WSMG_EINVAL when the window does not fit
int wsmg_convt_k4s2_bf16(const void* src, const void* w_ihwo, void* dst, double* stats, int nslab, int B, int H, int W, hipStream_t s);
// wsmg_conv_win3_wgrad.hip: weight gradient of a 3 x 3 / stride 1 / pad 1 layer out of a zero-padded LDS window (W <= 24, channel
// multiples of 64 / 128); WSMG_EINVAL otherwise
int wsmg_conv_win3_wgrad_bf16(const void* x, const void* dy, float* dw_ohwi, long long slab_floats, int B, int H, int W, int Cin, int Cout,
                              hipStream_t s);
int wsmg_conv_win3_wgrad_splits(int B, int H, int W, int Cin, int Cout);
// wsmg_rnn.hip: device pointer of the process-wide persistent-kernel status word (host-mapped; bits 1 gru_fwd, 2 gru_bwd, 4 lstm_fwd,
// 8 lstm_bwd), for kernels of other files that wait on a chain counter (wsmg_rows_gemm.hip)
unsigned* wsmgi_rnn_status_dev();
