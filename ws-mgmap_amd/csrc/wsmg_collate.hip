// Time-major collate of cached trajectories on the device (reference dagger_trainer.py:40-113
// followed by the trainer's `v.float().to(device)` at :614-617).
//
// The reference pads and stacks on the host in the on-disk dtypes, converts every observation tensor to float32
// on the CPU (1.3 GB per update for the ego map alone) and ships float32 over PCIe.  Here the episodes travel in
// their compact on-disk dtypes (float16 / uint8 / int64 / float32) and ONE kernel per sensor writes the padded,
// episode-interleaved float32 tensor [T][N][elems] the policy consumes: element (t, n, e) = src_n[t][e] for
// t < length_n, else the pad value (1.0 for observations, 0 for actions / weights).
#include "wsmg_common.h"

namespace {

enum { DT_F16 = 0, DT_U8 = 1, DT_I64 = 2, DT_F32 = 3 };

template <int DT>
__device__ __forceinline__ float load_as_float(const void* p, int64_t i) {
  if constexpr (DT == DT_F16) return (float)reinterpret_cast<const _Float16*>(p)[i];
  else if constexpr (DT == DT_U8) return (float)reinterpret_cast<const uint8_t*>(p)[i];
  else if constexpr (DT == DT_I64) return (float)reinterpret_cast<const int64_t*>(p)[i];
  else return reinterpret_cast<const float*>(p)[i];
}

// one thread = 4 consecutive elements of one (t, n) row when elems % 4 == 0 (V = 4), else one element (V = 1)
template <int DT, int V>
__global__ __launch_bounds__(256) void collate_pad_kernel(const void* const* __restrict__ src, const int* __restrict__ lengths,
                                                          int N, int T, int64_t elems, float pad, float* __restrict__ dst) {
  const int64_t per_row = elems / V;
  const int64_t total = (int64_t)T * N * per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / per_row;          // t * N + n
    const int64_t e = (i - row * per_row) * V;
    const int t = (int)(row / N), n = (int)(row - (int64_t)t * N);
    float v[V];
    if (t < lengths[n]) {
      const void* s = src[n];
      const int64_t o = (int64_t)t * elems + e;
      if constexpr (V == 4 && DT == DT_F16) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 q = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(s) + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)q[j];
      } else if constexpr (V == 4 && DT == DT_F32) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(s) + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = q[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = load_as_float<DT>(s, o + j);
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = pad;
    }
    if constexpr (V == 4) {
      *reinterpret_cast<f32x4*>(dst + row * elems + e) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
      dst[row * elems + e] = v[0];
    }
  }
}

// The cached ego map for the bf16 map stack (round 4): padded, episode-interleaved AND channels-last bf16 in one pass —
// dst[t][n][p][c] = bf16(src_n[t][c][p]) (float16 on disk -> float32 -> bf16, the roundings of wsmg_collate_pad followed by the
// policy's NCHW float32 -> NHWC bf16 conversion, so the values are bit-identical to that route) or bf16(pad) for t >= length_n.
// 64-channel x 64-pixel tiles through LDS: 8-byte loads along the pixel axis, 16-byte stores along the channel axis.
__global__ __launch_bounds__(256) void collate_pad_nhwc_bf16_kernel(const void* const* __restrict__ src, const int* __restrict__ lengths,
                                                                    int N, int C, int HW, float pad, bf16_t* __restrict__ dst) {
  __shared__ float tile[64][65];
  const int row = blockIdx.z, t = row / N, n = row - t * N;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  const bool live = t < lengths[n];
  if (live) {
    const _Float16* s = reinterpret_cast<const _Float16*>(src[n]) + (size_t)t * C * HW;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int id = tid + 256 * j;         // 1024 pieces of 4 pixels: 16 per channel row
      const int c = id >> 4, pq = (id & 15) * 4;
      h4 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
      if (p0 + pq < HW) v = *reinterpret_cast<const h4*>(s + (size_t)(c0 + c) * HW + p0 + pq);
#pragma unroll
      for (int k = 0; k < 4; ++k) tile[c][pq + k] = (float)v[k];
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int id = tid + 256 * j;           // 512 pieces of 8 channels: 8 per pixel
    const int p = id >> 3, cs = (id & 7) * 8;
    if (p0 + p < HW) {
      f32x4 lo, hi;
#pragma unroll
      for (int k = 0; k < 4; ++k) { lo[k] = live ? tile[cs + k][p] : pad; hi[k] = live ? tile[cs + 4 + k][p] : pad; }
      bf16_t* o = dst + ((size_t)row * HW + p0 + p) * C + c0 + cs;
      st4(o, lo);
      st4(o + 4, hi);
    }
  }
}

// The same tensor from the SPARSE record form (round 5; wsmgmap/data/codec.py): per episode n the presence bits [T_n][HW] (one 8-byte
// word per pixel: bit c = channel c), the per-pixel exclusive prefix of non-zeros inside its step off[T_n][HW], the per-step prefix
// base[T_n + 1], and the packed non-zero float16 values.  One wave per pixel, lane = channel: the word is a broadcast load, lane c's
// value sits at base + off + popcount(word & lanes below c) — 64 lanes read <= 64 consecutive values — and the 128-byte channel run
// of the pixel is one coalesced store.  float16 -> float32 -> bf16 as in the dense kernel: bit-identical output.
__global__ __launch_bounds__(256) void collate_ego_sparse_nhwc_bf16_kernel(const void* const* __restrict__ bits, const void* const* __restrict__ off,
                                                                           const void* const* __restrict__ base, const void* const* __restrict__ vals,
                                                                           const int* __restrict__ lengths, int N, int HW, float pad,
                                                                           bf16_t* __restrict__ dst) {
  const int row = blockIdx.y, t = row / N, n = row - t * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = t < lengths[n];
  const unsigned long long* const bw = reinterpret_cast<const unsigned long long*>(bits[n]) + (size_t)t * HW;
  const unsigned* const ow = reinterpret_cast<const unsigned*>(off[n]) + (size_t)t * HW;
  const _Float16* const vv = reinterpret_cast<const _Float16*>(vals[n]);
  const long long b0 = live ? reinterpret_cast<const long long*>(base[n])[t] : 0;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  constexpr int PPW = 16, U = 4;                       // pixels per wave and workgroup trip; loads of U pixels in flight
  const int p0 = (blockIdx.x * 4 + wave) * PPW;
  for (int i = 0; i < PPW; i += U) {
    unsigned long long w[U];
    unsigned o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + i + u;
      const bool ok = live && p < HW;
      w[u] = ok ? bw[p] : 0ull;
      o[u] = ok ? ow[p] : 0u;
    }
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool bit = (w[u] >> lane) & 1ull;
      v[u] = bit ? (float)vv[b0 + (long long)o[u] + __popcll(w[u] & below)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + i + u;
      if (p < HW) dst[((size_t)row * HW + p) * 64 + lane] = (bf16_t)(live ? v[u] : pad);
    }
  }
}

template <int DT>
int launch(const void* const* src, const int* lengths, int N, int T, int64_t elems, float pad, float* dst, hipStream_t s) {
  const bool v4 = (elems % 4) == 0;
  const int64_t total = (int64_t)T * N * (v4 ? elems / 4 : elems);
  int64_t g = wsmg_cdiv(total, 256);
  if (g > 16384) g = 16384;
  if (v4)
    hipLaunchKernelGGL((collate_pad_kernel<DT, 4>), dim3((unsigned)g), dim3(256), 0, s, src, lengths, N, T, elems, pad, dst);
  else
    hipLaunchKernelGGL((collate_pad_kernel<DT, 1>), dim3((unsigned)g), dim3(256), 0, s, src, lengths, N, T, elems, pad, dst);
  WSMG_RETURN_LAUNCH();
}

}  // namespace

extern "C" int wsmg_collate_pad(const void* const* src, const int* lengths, int N, int T, int64_t elems, int src_dtype,
                                float pad, float* dst, wsmg_stream_t stream) {
  if (N <= 0 || T <= 0 || elems <= 0) return WSMG_EINVAL;
  hipStream_t s = wsmg_s(stream);
  switch (src_dtype) {
    case DT_F16: return launch<DT_F16>(src, lengths, N, T, elems, pad, dst, s);
    case DT_U8: return launch<DT_U8>(src, lengths, N, T, elems, pad, dst, s);
    case DT_I64: return launch<DT_I64>(src, lengths, N, T, elems, pad, dst, s);
    case DT_F32: return launch<DT_F32>(src, lengths, N, T, elems, pad, dst, s);
    default: return WSMG_EINVAL;
  }
}

extern "C" int wsmg_collate_pad_nhwc_bf16(const void* const* src, const int* lengths, int N, int T, int C, int HW, float pad, void* dst,
                                          wsmg_stream_t stream) {
  if (N <= 0 || T <= 0 || C <= 0 || HW <= 0 || (C % 64) || (HW % 4) || (int64_t)T * N > 65535) return WSMG_EINVAL;
  hipLaunchKernelGGL(collate_pad_nhwc_bf16_kernel, dim3((unsigned)wsmg_cdiv(HW, 64), (unsigned)(C / 64), (unsigned)(T * N)), dim3(256), 0,
                     wsmg_s(stream), src, lengths, N, C, HW, pad, (bf16_t*)dst);
  WSMG_RETURN_LAUNCH();
}

/* The padded, episode-interleaved channels-last bf16 ego map [T][N][HW][64] from the SPARSE record form (wsmgmap/data/codec.py:
 * bits / off / base / vals per episode); values bit-identical to wsmg_collate_pad_nhwc_bf16 on the dense map. */
extern "C" int wsmg_collate_ego_sparse_nhwc_bf16(const void* const* bits, const void* const* off, const void* const* base,
                                                 const void* const* vals, const int* lengths, int N, int T, int C, int HW, float pad,
                                                 void* dst, wsmg_stream_t stream) {
  if (N <= 0 || T <= 0 || C != 64 || HW <= 0 || (int64_t)T * N > 65535) return WSMG_EINVAL;
  hipLaunchKernelGGL(collate_ego_sparse_nhwc_bf16_kernel, dim3((unsigned)wsmg_cdiv(HW, 64), (unsigned)(T * N)), dim3(256), 0, wsmg_s(stream),
                     bits, off, base, vals, lengths, N, HW, pad, (bf16_t*)dst);
  WSMG_RETURN_LAUNCH();
}
