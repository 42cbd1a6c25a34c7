// Small NHWC float32 helpers of the UNet-style decoder: ReLU, MaxPool2d(3,2,1),
// bilinear 2x upsample (align_corners=True), avg_pool2d(2), and the NCHW<->NHWC boundary
// transposes.  Reference call sites: map_encoder.py:80,84,102,108; mg_map_policy.py:89-100,197.
// All are streaming, HBM-bound kernels; backward passes are written as gathers so results
// are deterministic (no atomics).
#include <type_traits>

#include "wsmg_common.h"

namespace {

int sgrid(int64_t n) {
  int64_t g = wsmg_cdiv(n, 256);
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

template <class T>
__global__ void relu_fwd_kernel(const T* x, T* y, int64_t n4) {
  GRID_STRIDE(i, n4) {
    f32x4 v = ld4(x + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    st4(y + i * 4, v);
  }
}
template <class T>
__global__ void relu_bwd_kernel(const T* dy, const T* y, T* dx, int64_t n4) {
  GRID_STRIDE(i, n4) {
    f32x4 g = ld4(dy + i * 4), v = ld4(y + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = v[j] > 0.f ? g[j] : 0.f;
    st4(dx + i * 4, g);
  }
}

// ---- MaxPool2d(kernel 3, stride 2, pad 1): first maximum in (ky,kx) scan order wins ties (ATen)
// All pool / upsample kernels below: one thread = 4 consecutive channels of one pixel (16-byte f32 / 8-byte bf16
// accesses; C % 4 == 0 is guaranteed by the engine's 32-channel padding), pixel geometry computed once per 4 values.
#define PIX_DECODE(i, C4, WW, HH, c, px, py, b)   \
  const int c = (int)((i) % (C4)) * 4;             \
  const int64_t p_ = (i) / (C4);                   \
  const int px = (int)(p_ % (WW));                 \
  const int py = (int)((p_ / (WW)) % (HH));        \
  const int b = (int)(p_ / ((int64_t)(WW) * (HH)));

template <class T>
__global__ void maxpool_fwd_kernel(const T* x, T* y, int B, int H, int W, int C, int OH, int OW) {
  const int C4 = C / 4;
  int64_t n = (int64_t)B * OH * OW * C4;
  GRID_STRIDE(i, n) {
    PIX_DECODE(i, C4, OW, OH, c, ox, oy, b)
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int ky = 0; ky < 3; ++ky) {
      int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        f32x4 v = ld4(x + (((size_t)b * H + iy) * W + ix) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] > best[j] || v[j] != v[j]) best[j] = v[j];
      }
    }
    st4(y + i * 4, best);
  }
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ inline i32x4 maxpool_argmax4(const T* x, int b, int oy, int ox, int c, int H, int W, int C) {
  f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  i32x4 arg = {-1, -1, -1, -1};
  for (int ky = 0; ky < 3; ++ky) {
    int iy = oy * 2 - 1 + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      int ix = ox * 2 - 1 + kx;
      if (ix < 0 || ix >= W) continue;
      f32x4 v = ld4(x + (((size_t)b * H + iy) * W + ix) * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (v[j] > best[j] || v[j] != v[j] || arg[j] < 0) { best[j] = v[j]; arg[j] = iy * W + ix; }
    }
  }
  return arg;
}

template <class T>
__global__ void maxpool_bwd_kernel(const T* dy, const T* x, T* dx, int B, int H, int W, int C, int OH,
                                   int OW) {
  const int C4 = C / 4;
  int64_t n = (int64_t)B * H * W * C4;
  GRID_STRIDE(i, n) {
    PIX_DECODE(i, C4, W, H, c, ix, iy, b)
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    int oy0 = iy / 2, oy1 = (iy + 1) / 2;  // windows [2oy-1, 2oy+1] containing iy
    int ox0 = ix / 2, ox1 = (ix + 1) / 2;
    for (int oy = oy0; oy <= oy1; ++oy) {
      if (oy >= OH) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        if (ox >= OW) continue;
        const i32x4 arg = maxpool_argmax4(x, b, oy, ox, c, H, W, C);
        const f32x4 d = ld4(dy + (((size_t)b * OH + oy) * OW + ox) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (arg[j] == iy * W + ix) g[j] += d[j];
      }
    }
    st4(dx + i * 4, g);
  }
}

// ---- the same pool with the winning tap of every output element kept (round 3): one byte per element, tap = 3 ky + kx in the
// scan order whose first maximum wins.  The backward above recomputes the arg-max of up to four windows per input element —
// 36 loads and their compares for one store: 50 us for the decoder's 12 x 12 map at B = 512, a 9 MB tensor, on the critical
// chain of small launches; with the taps it is four index words and four gradient loads.
template <class T>
__global__ void maxpool_fwd_idx_kernel(const T* x, T* y, unsigned* idx, int B, int H, int W, int C, int OH, int OW) {
  const int C4 = C / 4;
  int64_t n = (int64_t)B * OH * OW * C4;
  GRID_STRIDE(i, n) {
    PIX_DECODE(i, C4, OW, OH, c, ox, oy, b)
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {-1, -1, -1, -1};
    for (int ky = 0; ky < 3; ++ky) {
      int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        f32x4 v = ld4(x + (((size_t)b * H + iy) * W + ix) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] > best[j] || v[j] != v[j] || arg[j] < 0) { best[j] = v[j]; arg[j] = 3 * ky + kx; }
      }
    }
    st4(y + i * 4, best);
    idx[i] = (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
  }
}
template <class T>
__global__ void maxpool_bwd_idx_kernel(const T* dy, const unsigned* idx, T* dx, int B, int H, int W, int C, int OH, int OW) {
  const int C4 = C / 4;
  int64_t n = (int64_t)B * H * W * C4;
  GRID_STRIDE(i, n) {
    PIX_DECODE(i, C4, W, H, c, ix, iy, b)
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    const int oy0 = iy / 2, oy1 = (iy + 1) / 2, ox0 = ix / 2, ox1 = (ix + 1) / 2;   // windows [2o - 1, 2o + 1] containing (iy, ix)
    for (int oy = oy0; oy <= oy1; ++oy) {
      if (oy >= OH) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        if (ox >= OW) continue;
        const unsigned tap = (unsigned)(3 * (iy - (2 * oy - 1)) + (ix - (2 * ox - 1)));
        const size_t o = (((size_t)b * OH + oy) * OW + ox) * C4 + c / 4;
        const unsigned w = idx[o];
        const f32x4 d = ld4(dy + o * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (((w >> (8 * j)) & 0xffu) == tap) g[j] += d[j];
      }
    }
    st4(dx + i * 4, g);
  }
}

// ---- bilinear x2, align_corners=True (ATen upsample_bilinear2d: src = dst * (in-1)/(out-1))
__device__ inline void up_src(int o, int in, float scale, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * (float)o;
  i0 = (int)s;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

template <class T>
__global__ void upsample_fwd_kernel(const T* x, T* y, int B, int H, int W, int C) {
  const int OH = 2 * H, OW = 2 * W, C4 = C / 4;
  const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  int64_t n = (int64_t)B * OH * OW * C4;
  GRID_STRIDE(i, n) {
    PIX_DECODE(i, C4, OW, OH, c, ox, oy, b)
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    up_src(oy, H, sh, y0, y1, hy0, hy1);
    up_src(ox, W, sw, x0, x1, wx0, wx1);
    const T* xb = x + (size_t)b * H * W * C + c;
    const f32x4 v00 = ld4(xb + ((size_t)y0 * W + x0) * C), v01 = ld4(xb + ((size_t)y0 * W + x1) * C);
    const f32x4 v10 = ld4(xb + ((size_t)y1 * W + x0) * C), v11 = ld4(xb + ((size_t)y1 * W + x1) * C);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = hy0 * (wx0 * v00[j] + wx1 * v01[j]) + hy1 * (wx0 * v10[j] + wx1 * v11[j]);
    st4(y + i * 4, o);
  }
}

template <class T>
__global__ void upsample_bwd_kernel(const T* dy, T* dx, int B, int H, int W, int C, int64_t ld_dy) {   // ld_dy: pixel stride of dy
  const int OH = 2 * H, OW = 2 * W, C4 = C / 4;
  const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  int64_t n = (int64_t)B * H * W * C4;
  GRID_STRIDE(i, n) {
    PIX_DECODE(i, C4, W, H, c, ix, iy, b)
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    int oy_lo = 2 * iy - 2 < 0 ? 0 : 2 * iy - 2, oy_hi = 2 * iy + 3 > OH - 1 ? OH - 1 : 2 * iy + 3;
    int ox_lo = 2 * ix - 2 < 0 ? 0 : 2 * ix - 2, ox_hi = 2 * ix + 3 > OW - 1 ? OW - 1 : 2 * ix + 3;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float hy0, hy1;
      up_src(oy, H, sh, y0, y1, hy0, hy1);
      float wy = (y0 == iy ? hy0 : 0.f) + (y1 == iy ? hy1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float wx0, wx1;
        up_src(ox, W, sw, x0, x1, wx0, wx1);
        float wx = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
        if (wx == 0.f) continue;
        const f32x4 d = ld4(dy + (((size_t)b * OH + oy) * OW + ox) * ld_dy + c);
        const float wgt = wy * wx;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] += wgt * d[j];
      }
    }
    st4(dx + i * 4, g);
  }
}

// ---- avg_pool2d(2,2), H and W even
template <class T>
__global__ void avgpool_fwd_kernel(const T* x, T* y, int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, C4 = C / 4;
  int64_t n = (int64_t)B * OH * OW * C4;
  GRID_STRIDE(i, n) {
    PIX_DECODE(i, C4, OW, OH, c, ox, oy, b)
    const T* xb = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
    const f32x4 a0 = ld4(xb), a1 = ld4(xb + C), a2 = ld4(xb + (size_t)W * C), a3 = ld4(xb + (size_t)W * C + C);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (a0[j] + a1[j] + a2[j] + a3[j]) * 0.25f;
    st4(y + i * 4, o);
  }
}
template <class T>
__global__ void avgpool_bwd_kernel(const T* dy, T* dx, int B, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2, C4 = C / 4;
  int64_t n = (int64_t)B * H * W * C4;
  GRID_STRIDE(i, n) {
    PIX_DECODE(i, C4, W, H, c, ix, iy, b)
    f32x4 d = ld4(dy + (((size_t)b * OH + iy / 2) * OW + ix / 2) * C + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] *= 0.25f;
    st4(dx + i * 4, d);
  }
}

// ---- [B][R][S] -> [B][S][R'] transposes through a 32x33 LDS tile (coalesced both sides)
// TO_NHWC: src rows = channels (R = C_src), cols = pixels;  dst [pixel][C_dst], zero-filled past C_src.
template <bool TO_NHWC, class TI, class TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* x, TO* y, int C_src, int HW, int C_dst) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (TO_NHWC) {
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int c = c0 + ty + 8 * j, p = p0 + tx;
      tile[ty + 8 * j][tx] = (c < C_src && p < HW) ? ldf(x + ((size_t)b * C_src + c) * HW + p) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int p = p0 + ty + 8 * j, c = c0 + tx;
      if (p < HW && c < C_dst) stf(y + ((size_t)b * HW + p) * C_dst + c, tile[tx][ty + 8 * j]);
    }
  } else {
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int p = p0 + ty + 8 * j, c = c0 + tx;
      tile[ty + 8 * j][tx] = (p < HW && c < C_src) ? ldf(x + ((size_t)b * HW + p) * C_src + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int c = c0 + ty + 8 * j, p = p0 + tx;
      if (c < C_dst && p < HW) stf(y + ((size_t)b * C_dst + c) * HW + p, tile[tx][ty + 8 * j]);
    }
  }
}

// NCHW float32 -> NHWC (float32 / bf16) for C_src % 64 == 0, HW % 4 == 0: 64 channels x 64 pixels per workgroup,
// 16-byte loads along the pixel axis and 16-byte (bf16: 8 channels) / 2 x 16-byte (f32) stores along the
// channel axis.  The cached ego map of the update path (1.3 GB per update) goes through this once.
template <class TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc64_kernel(const float* __restrict__ x, TO* __restrict__ y, int C_src, int HW,
                                                             int C_dst) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int id = tid + 256 * j;         // 1024 float4 pieces: 16 per channel row
    const int c = id >> 4, pq = (id & 15) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p0 + pq < HW) v = *reinterpret_cast<const f32x4*>(x + ((size_t)b * C_src + c0 + c) * HW + p0 + pq);
    tile[c][pq] = v[0]; tile[c][pq + 1] = v[1]; tile[c][pq + 2] = v[2]; tile[c][pq + 3] = v[3];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int id = tid + 256 * j;         // 512 pieces of 8 channels: 8 per pixel
    const int p = id >> 3, cs = (id & 7) * 8;
    if (p0 + p < HW) {
      f32x4 lo = {tile[cs][p], tile[cs + 1][p], tile[cs + 2][p], tile[cs + 3][p]};
      f32x4 hi = {tile[cs + 4][p], tile[cs + 5][p], tile[cs + 6][p], tile[cs + 7][p]};
      TO* dst = y + ((size_t)b * HW + p0 + p) * C_dst + c0 + cs;
      st4(dst, lo);
      st4(dst + 4, hi);
    }
  }
}

// ---- fused per-pixel cross-entropy over NHWC logits (policy.py:61-66: F.cross_entropy(pred_sem_map, target,
// reduction='none')): the 27 classes of a pixel are one 64/128-byte run of the 32-channel-padded conv output, so the
// loss needs no NHWC->NCHW transpose and no materialised log-softmax.  One thread per pixel row.
template <class T>
__global__ __launch_bounds__(256) void ce_nhwc_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                          int64_t rows, int classes, float* __restrict__ loss) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float v[32];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x4 q = ld4(logits + r * 32 + 4 * j);
    v[4 * j] = q[0]; v[4 * j + 1] = q[1]; v[4 * j + 2] = q[2]; v[4 * j + 3] = q[3];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 32; ++c) if (c < classes) mx = fmaxf(mx, v[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) if (c < classes) sum += expf(v[c] - mx);
  const int t = (int)target[r];
  float vt = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) if (c == t) vt = v[c];
  loss[r] = (mx + logf(sum)) - vt;
}

// dlogits[r][c] = (softmax(logits[r])[c] - [c == target[r]]) * gscale[r / rows_per_sample]; padded channels get 0
template <class T>
__global__ __launch_bounds__(256) void ce_nhwc_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target,
                                                          const float* __restrict__ gloss, int64_t rows, int classes,
                                                          T* __restrict__ dlogits) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float v[32];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x4 q = ld4(logits + r * 32 + 4 * j);
    v[4 * j] = q[0]; v[4 * j + 1] = q[1]; v[4 * j + 2] = q[2]; v[4 * j + 3] = q[3];
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < 32; ++c) if (c < classes) mx = fmaxf(mx, v[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) { v[c] = c < classes ? expf(v[c] - mx) : 0.f; sum += v[c]; }
  const float g = gloss[r], inv = 1.f / sum;
  const int t = (int)target[r];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x4 q;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int c = 4 * j + e; q[e] = c < classes ? (v[c] * inv - (c == t ? 1.f : 0.f)) * g : 0.f; }
    st4(dlogits + r * 32 + 4 * j, q);
  }
}

// ---- conv weight layouts in one launch: OIHW float32 parameter -> OHWI (forward / weight-gradient operand) and IHWO
// (backward-data operand) in the compute dtype, input channels zero-padded to I_pad.  Replaces per layer and update a
// permute copy, a dtype cast, a second permute copy (and an F.pad when the engine pads channels).
template <class T>
__global__ __launch_bounds__(256) void weight_relayout_kernel(const float* __restrict__ w, int O, int I, int KH, int KW, int I_pad,
                                                              T* __restrict__ ohwi, T* __restrict__ ihwo) {
  const int64_t n = (int64_t)O * KH * KW * I_pad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx % I_pad);
    int64_t r = idx / I_pad;
    const int kw = (int)(r % KW); r /= KW;
    const int kh = (int)(r % KH);
    const int o = (int)(r / KH);
    const float v = i < I ? w[(((int64_t)o * I + i) * KH + kh) * KW + kw] : 0.f;
    stf(ohwi + idx, v);
    if (ihwo) stf(ihwo + (((int64_t)i * KH + kh) * KW + kw) * O + o, v);
  }
}
// the same for up to WSMG_RELAYOUT_MAX parameters in ONE launch (blockIdx.y = parameter): the map stack has 20 convolutions,
// and 20 five-microsecond launches in front of them were 0.1 ms of the forward pass's critical path
struct RelayoutBatch { WsmgRelayoutDesc d[WSMG_RELAYOUT_MAX]; };
template <class T>
__global__ __launch_bounds__(256) void weight_relayout_multi_kernel(RelayoutBatch b) {
  const WsmgRelayoutDesc& d = b.d[blockIdx.y];
  const int O = d.O, I = d.I, KH = d.KH, KW = d.KW, I_pad = d.I_pad;
  const float* __restrict__ w = d.w_oihw;
  T* __restrict__ ohwi = (T*)d.w_ohwi;
  T* __restrict__ ihwo = (T*)d.w_ihwo;
  const int64_t n = (int64_t)O * KH * KW * I_pad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int i = (int)(idx % I_pad);
    int64_t r = idx / I_pad;
    const int kw = (int)(r % KW); r /= KW;
    const int kh = (int)(r % KH);
    const int o = (int)(r / KH);
    const float v = i < I ? w[(((int64_t)o * I + i) * KH + kh) * KW + kw] : 0.f;
    stf(ohwi + idx, v);
    if (ihwo) stf(ihwo + (((int64_t)i * KH + kh) * KW + kw) * O + o, v);
  }
}
// dW [O][KH][KW][I_pad] float32 -> the parameter's OIHW gradient (padded channels dropped)
__global__ __launch_bounds__(256) void weight_grad_to_oihw_kernel(const float* __restrict__ dw, int O, int I, int KH, int KW,
                                                                  int I_pad, float* __restrict__ out) {
  const int64_t n = (int64_t)O * I * KH * KW;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
    const int kw = (int)(idx % KW);
    int64_t r = idx / KW;
    const int kh = (int)(r % KH); r /= KH;
    const int i = (int)(r % I);
    const int o = (int)(r / I);
    out[idx] = dw[(((int64_t)o * KH + kh) * KW + kw) * I_pad + i];
  }
}

// The deterministic weight gradient's last step: dW[o][i][kh][kw] (OIHW, i < I) = sum over z < nsplit of ws[z][o][kh][kw][i]
// (OHWI slabs of I_pad channels, written whole by the weight-gradient kernels' workgroups — wsmg_conv2d_bwd_weight*_slabs).
// The order of the additions depends on nsplit only — four z ranges summed in z order by four threads, their results added
// in range order — so dW is bit-identical from run to run (float atomics add in arrival order).  A thread owns 4 consecutive
// input channels (one 16-byte load per slab); the OIHW stores are 4-byte, KH*KW apart: dW is 10-40x smaller than the slabs.
__global__ __launch_bounds__(256) void weight_grad_reduce_oihw_kernel(const float* __restrict__ ws, int nsplit, int64_t slab4, int I,
                                                                      int KH, int KW, int I_pad, float* __restrict__ out) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  __shared__ f32x4v part[3][64];
  const int ql = threadIdx.x & 63, zp = threadIdx.x >> 6;
  const int64_t q = (int64_t)blockIdx.x * 64 + ql;
  const int per = (nsplit + 3) >> 2;
  const int z0 = zp * per, z1 = z0 + per < nsplit ? z0 + per : nsplit;
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  if (q < slab4) {
    const f32x4v* p = reinterpret_cast<const f32x4v*>(ws) + (int64_t)z0 * slab4 + q;
    int z = z0;
    for (; z + 4 <= z1; z += 4, p += 4 * slab4) {     // four loads in flight, added in z order
      const f32x4v a = p[0], b = p[slab4], c = p[2 * slab4], d = p[3 * slab4];
      acc += a; acc += b; acc += c; acc += d;
    }
    for (; z < z1; ++z, p += slab4) acc += p[0];
  }
  if (zp) part[zp - 1][ql] = acc;
  __syncthreads();
  if (zp || q >= slab4) return;
  acc += part[0][ql]; acc += part[1][ql]; acc += part[2][ql];
  const int64_t e = q * 4;                           // flat OHWI index of the first of the 4 channels
  const int i = (int)(e % I_pad);
  int64_t r = e / I_pad;
  const int kw = (int)(r % KW); r /= KW;
  const int kh = (int)(r % KH);
  const int64_t o = r / KH;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (i + j < I) out[((o * I + i + j) * KH + kh) * KW + kw] = acc[j];
}

// ---- mean over a short innermost axis: out[r] = mean(x[r][0..n)), x [R][n] float32 contiguous (rgb_linear's AdaptiveAvgPool1d(1)
// over the 7 x 7 positions of the cached RGB feature, mg_map_policy.py:90-96: R = B * 512 rows of n = 49).  A thread per row would
// read 196-byte rows 196 bytes apart; here a workgroup loads its 256 rows as one contiguous, coalesced run into LDS and every
// thread adds its row from there (row pitch n floats: conflict-free for odd n).
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ x, int64_t R, int n, float* __restrict__ out) {
  extern __shared__ float rows_lds[];
  const int64_t r0 = (int64_t)blockIdx.x * 256;
  const int nr = R - r0 < 256 ? (int)(R - r0) : 256;
  const float* src = x + r0 * n;
  const int total = nr * n;
  for (int i = threadIdx.x; i < total; i += 256) rows_lds[i] = src[i];
  __syncthreads();
  if ((int)threadIdx.x < nr) {
    float s = 0.f;
    const float* p = rows_lds + threadIdx.x * n;
    for (int j = 0; j < n; ++j) s += p[j];
    out[r0 + threadIdx.x] = s / (float)n;
  }
}

// ---- channel concatenation of two NHWC tensors (the UNet skip connections, map_encoder.py:104,110, and
// mg_map_policy.py:99): out[p] = a[p] ++ b[p], one 16-byte chunk per thread, the pixel index computed once per chunk
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ void cat_channels_kernel(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b, u32x4_t* __restrict__ y,
                                    int64_t rows, int ca, int cb) {
  const int cy = ca + cb;
  const int64_t n = rows * cy;
  GRID_STRIDE(i, n) {
    const int64_t p = i / cy;
    const int c = (int)(i - p * cy);
    y[i] = c < ca ? a[p * ca + c] : b[p * cb + (c - ca)];
  }
}

// Rollout route of the UNet decoders (unet_encoder.py:95-109, map_encoder.py:103-110): cat([upsample2x(a), b], channels) in one
// pass, bf16 — the upsampled tensor is never written (and read back by the concatenation); 8 channels per thread, the
// interpolation arithmetic of upsample_fwd_kernel.
__global__ void upsample_cat_bf16_kernel(const bf16_t* __restrict__ a, const u32x4_t* __restrict__ b, u32x4_t* __restrict__ y, int B,
                                         int H, int W, int ca8, int cb8) {
  const int OH = 2 * H, OW = 2 * W, cy = ca8 + cb8, C = ca8 * 8;
  const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const int64_t n = (int64_t)B * OH * OW * cy;
  GRID_STRIDE(i, n) {
    const int64_t p = i / cy;
    const int c = (int)(i - p * cy);
    if (c >= ca8) {
      y[i] = b[p * cb8 + (c - ca8)];
      continue;
    }
    const int ox = (int)(p % OW);
    const int64_t q = p / OW;
    const int oy = (int)(q % OH), bb = (int)(q / OH);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    up_src(oy, H, sh, y0, y1, hy0, hy1);
    up_src(ox, W, sw, x0, x1, wx0, wx1);
    const bf16_t* xb = a + (size_t)bb * H * W * C + c * 8;
    u32x4_t o;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      const f32x4 v00 = ld4(xb + ((size_t)y0 * W + x0) * C + 4 * hlf), v01 = ld4(xb + ((size_t)y0 * W + x1) * C + 4 * hlf);
      const f32x4 v10 = ld4(xb + ((size_t)y1 * W + x0) * C + 4 * hlf), v11 = ld4(xb + ((size_t)y1 * W + x1) * C + 4 * hlf);
      unsigned short r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16_t v = (bf16_t)(hy0 * (wx0 * v00[j] + wx1 * v01[j]) + hy1 * (wx0 * v10[j] + wx1 * v11[j]));
        r[j] = __builtin_bit_cast(unsigned short, v);
      }
      o[2 * hlf] = (unsigned)r[0] | ((unsigned)r[1] << 16);
      o[2 * hlf + 1] = (unsigned)r[2] | ((unsigned)r[3] << 16);
    }
    y[i] = o;
  }
}

// bf16: 8 values (one 16-byte access) per thread
__device__ __forceinline__ unsigned relu2(unsigned v) {   // two packed bf16: x > 0 ? x : 0 (NaN -> 0, like the float compare)
  const float lo = __uint_as_float(v << 16), hi = __uint_as_float(v & 0xffff0000u);
  return (lo > 0.f ? (v & 0xffffu) : 0u) | (hi > 0.f ? (v & 0xffff0000u) : 0u);
}
__device__ __forceinline__ unsigned mask2(unsigned g, unsigned y) {   // g where y > 0, else 0
  const float lo = __uint_as_float(y << 16), hi = __uint_as_float(y & 0xffff0000u);
  return (lo > 0.f ? (g & 0xffffu) : 0u) | (hi > 0.f ? (g & 0xffff0000u) : 0u);
}
// a + b + c of three bf16 tensors in one pass, float32 sums, one rounding (round 3: the gradient of a tensor with three
// consumers — the encoded map feeds the token projection, the decoder's stem and its full-resolution branch — instead of two
// add launches of six passes over 151 MB each way)
__device__ __forceinline__ unsigned add3_2(unsigned a, unsigned b, unsigned c) {
  const float lo = __uint_as_float(a << 16) + __uint_as_float(b << 16) + __uint_as_float(c << 16);
  const float hi = __uint_as_float(a & 0xffff0000u) + __uint_as_float(b & 0xffff0000u) + __uint_as_float(c & 0xffff0000u);
  const bf16_t l = (bf16_t)lo, h = (bf16_t)hi;
  return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}
__global__ void add3_bf16_kernel(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b, const u32x4_t* __restrict__ c,
                                 u32x4_t* __restrict__ y, int64_t n8) {
  GRID_STRIDE(i, n8) {
    const u32x4_t u = a[i], v = b[i], w = c[i];
    u32x4_t o = {add3_2(u[0], v[0], w[0]), add3_2(u[1], v[1], w[1]), add3_2(u[2], v[2], w[2]), add3_2(u[3], v[3], w[3])};
    y[i] = o;
  }
}
__global__ void relu_fwd8_kernel(const u32x4_t* __restrict__ x, u32x4_t* __restrict__ y, int64_t n8) {
  GRID_STRIDE(i, n8) {
    const u32x4_t v = x[i];
    u32x4_t o = {relu2(v[0]), relu2(v[1]), relu2(v[2]), relu2(v[3])};
    y[i] = o;
  }
}
__global__ void relu_bwd8_kernel(const u32x4_t* __restrict__ dy, const u32x4_t* __restrict__ y, u32x4_t* __restrict__ dx, int64_t n8) {
  GRID_STRIDE(i, n8) {
    const u32x4_t g = dy[i], v = y[i];
    u32x4_t o = {mask2(g[0], v[0]), mask2(g[1], v[1]), mask2(g[2], v[2]), mask2(g[3], v[3])};
    dx[i] = o;
  }
}

// the same with a row stride on dy (rows of C8 16-byte pieces, `ld8` pieces apart): a channel slice of a wider gradient read in place
__global__ void relu_bwd8_rows_kernel(const u32x4_t* __restrict__ dy, int64_t ld8, const u32x4_t* __restrict__ y, u32x4_t* __restrict__ dx,
                                      int C8, int64_t n8) {
  GRID_STRIDE(i, n8) {
    const int64_t r = i / C8;
    const u32x4_t g = dy[r * ld8 + (i - r * C8)], v = y[i];
    u32x4_t o = {mask2(g[0], v[0]), mask2(g[1], v[1]), mask2(g[2], v[2]), mask2(g[3], v[3])};
    dx[i] = o;
  }
}

// dx[b][i][c] = (dx[b][i][c] + g[b][c] * inv_I) * (x[b][i][c] > 0), in place: the gradient of the map tokens from the attention
// (already in dx) merged with the broadcast gradient of their token mean (mg_map_policy.py:217: AdaptiveAvgPool1d over the 576
// tokens) and masked by the fused ReLU of the convolution that produced the tokens (map_cated_linear, :99-100) — one pass
// instead of an add launch over a materialised broadcast plus a ReLU-mask launch.
template <class T>
__global__ void token_grad_merge_kernel(T* __restrict__ dx, const T* __restrict__ x, const float* __restrict__ g, int I, int C,
                                        float inv_I, int relu, int64_t n4) {
  const int C4 = C / 4;
  GRID_STRIDE(i, n4) {
    const int c = (int)(i % C4) * 4;
    const int64_t b = i / ((int64_t)C4 * I);
    f32x4 d = ld4(dx + i * 4);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + b * C + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = fmaf(gv[j], inv_I, d[j]);
    if (relu) {
      const f32x4 xv = ld4(x + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = xv[j] > 0.f ? d[j] : 0.f;
    }
    st4(dx + i * 4, d);
  }
}
template <class T>
int token_grad_merge_t(T* dx, const T* x, const float* g, int B, int I, int C, int relu, wsmg_stream_t stream) {
  if (B <= 0 || I <= 0 || C <= 0 || (C & 3)) return WSMG_EINVAL;
  const int64_t n4 = (int64_t)B * I * C / 4;
  hipLaunchKernelGGL(token_grad_merge_kernel<T>, dim3(sgrid(n4)), dim3(256), 0, wsmg_s(stream), dx, x, g, I, C, 1.0f / (float)I, relu, n4);
  WSMG_RETURN_LAUNCH();
}

template <class T>
int relu_fwd_t(const T* x, T* y, int64_t n, wsmg_stream_t stream) {
  if (n <= 0 || n % 4) return WSMG_EINVAL;
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (n % 8 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
      hipLaunchKernelGGL(relu_fwd8_kernel, dim3(sgrid(n / 8)), dim3(256), 0, wsmg_s(stream), (const u32x4_t*)x, (u32x4_t*)y, n / 8);
      WSMG_RETURN_LAUNCH();
    }
  }
  hipLaunchKernelGGL(relu_fwd_kernel<T>, dim3(sgrid(n / 4)), dim3(256), 0, wsmg_s(stream), x, y, n / 4);
  WSMG_RETURN_LAUNCH();
}
template <class T>
int relu_bwd_t(const T* dy, const T* y, T* dx, int64_t n, wsmg_stream_t stream) {
  if (n <= 0 || n % 4) return WSMG_EINVAL;
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (n % 8 == 0 && (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15) == 0) {
      hipLaunchKernelGGL(relu_bwd8_kernel, dim3(sgrid(n / 8)), dim3(256), 0, wsmg_s(stream), (const u32x4_t*)dy, (const u32x4_t*)y,
                         (u32x4_t*)dx, n / 8);
      WSMG_RETURN_LAUNCH();
    }
  }
  hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(sgrid(n / 4)), dim3(256), 0, wsmg_s(stream), dy, y, dx, n / 4);
  WSMG_RETURN_LAUNCH();
}
template <class T>
int maxpool_fwd_t(const T* x, T* y, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t stream) {
  if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1 || B <= 0 || C <= 0 || (C & 3)) return WSMG_EINVAL;
  hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(sgrid((int64_t)B * OH * OW * C / 4)), dim3(256), 0, wsmg_s(stream), x, y, B,
                     H, W, C, OH, OW);
  WSMG_RETURN_LAUNCH();
}
template <class T>
int maxpool_bwd_t(const T* dy, const T* x, T* dx, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t stream) {
  if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1 || B <= 0 || C <= 0 || (C & 3)) return WSMG_EINVAL;
  hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(sgrid((int64_t)B * H * W * C / 4)), dim3(256), 0, wsmg_s(stream), dy, x, dx,
                     B, H, W, C, OH, OW);
  WSMG_RETURN_LAUNCH();
}
// upsample_bwd_kernel for bf16 with 8 channels per thread (round 6).  The same sum in the same order — output rows ascending, output
// columns ascending, g += (wy wx) dy — so the result is bit-identical; but the row and the column weights of the (at most 6 + 6)
// candidate positions are worked out once per thread instead of once per (row, column) pair (36 evaluations of the source-index
// arithmetic per thread were most of the kernel's time: 61 us for the 151 MB layer), the candidates with weight 0 are dropped before any
// load, and the loads of a row (4-5 of 16 bytes) are issued together.
__global__ __launch_bounds__(256) void upsample_bwd8_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B, int H, int W, int C,
                                                            int64_t ld_dy) {
  typedef unsigned int u32x4u __attribute__((ext_vector_type(4)));
  const int OH = 2 * H, OW = 2 * W, C8 = C / 8;
  const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  const int64_t n = (int64_t)B * H * W * C8;
  GRID_STRIDE(i, n) {
    const int c = (int)(i % C8) * 8;
    const int64_t p_ = i / C8;
    const int ix = (int)(p_ % W), iy = (int)((p_ / W) % H), b = (int)(p_ / ((int64_t)W * H));
    // the output rows / columns that read this input row / column (4, sometimes 5; 6 at H = 2), ascending, with their weights
    int oys[6], oxs[6], ny = 0, nx = 0;
    float wys[6], wxs[6];
    const int oy_lo = 2 * iy - 2 < 0 ? 0 : 2 * iy - 2, oy_hi = 2 * iy + 3 > OH - 1 ? OH - 1 : 2 * iy + 3;
    const int ox_lo = 2 * ix - 2 < 0 ? 0 : 2 * ix - 2, ox_hi = 2 * ix + 3 > OW - 1 ? OW - 1 : 2 * ix + 3;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int oy = oy_lo + t;
      if (oy <= oy_hi) {
        int y0, y1;
        float hy0, hy1;
        up_src(oy, H, sh, y0, y1, hy0, hy1);
        const float wy = (y0 == iy ? hy0 : 0.f) + (y1 == iy ? hy1 : 0.f);
        if (wy != 0.f) { oys[ny] = oy; wys[ny] = wy; ++ny; }
      }
      const int ox = ox_lo + t;
      if (ox <= ox_hi) {
        int x0, x1;
        float wx0, wx1;
        up_src(ox, W, sw, x0, x1, wx0, wx1);
        const float wx = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
        if (wx != 0.f) { oxs[nx] = ox; wxs[nx] = wx; ++nx; }
      }
    }
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    const bf16_t* const base = dy + (size_t)b * OH * OW * ld_dy + c;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (a >= ny) break;
      u32x4u d[6];
#pragma unroll
      for (int e = 0; e < 6; ++e)
        if (e < nx) d[e] = *reinterpret_cast<const u32x4u*>(base + ((size_t)oys[a] * OW + oxs[e]) * ld_dy);
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        if (e >= nx) break;
        const float wgt = wys[a] * wxs[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          g[2 * j] += wgt * __uint_as_float(d[e][j] << 16);
          g[2 * j + 1] += wgt * __uint_as_float(d[e][j] & 0xffff0000u);
        }
      }
    }
    u32x4u o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16_t lo = (bf16_t)g[2 * j], hi = (bf16_t)g[2 * j + 1];
      o[j] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    *reinterpret_cast<u32x4u*>(dx + i * 8) = o;
  }
}

template <class T>
int upsample_fwd_t(const T* x, T* y, int B, int H, int W, int C, wsmg_stream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return WSMG_EINVAL;
  hipLaunchKernelGGL(upsample_fwd_kernel<T>, dim3(sgrid((int64_t)B * H * W * C)), dim3(256), 0, wsmg_s(stream), x,
                     y, B, H, W, C);
  WSMG_RETURN_LAUNCH();
}
template <class T>
int upsample_bwd_t(const T* dy, T* dx, int B, int H, int W, int C, wsmg_stream_t stream, int64_t ld_dy = 0) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return WSMG_EINVAL;
  if (ld_dy == 0) ld_dy = C;
  if (ld_dy < C || (ld_dy & 7) || ((uintptr_t)dy & 15)) return WSMG_EINVAL;
  if constexpr (std::is_same<T, bf16_t>::value) {
    if ((C & 7) == 0 && ((uintptr_t)dx & 15) == 0) {
      hipLaunchKernelGGL(upsample_bwd8_kernel, dim3(sgrid((int64_t)B * H * W * C / 8)), dim3(256), 0, wsmg_s(stream), dy, dx, B, H, W, C, ld_dy);
      WSMG_RETURN_LAUNCH();
    }
  }
  hipLaunchKernelGGL(upsample_bwd_kernel<T>, dim3(sgrid((int64_t)B * H * W * C / 4)), dim3(256), 0, wsmg_s(stream), dy, dx,
                     B, H, W, C, ld_dy);
  WSMG_RETURN_LAUNCH();
}
template <class T>
int avgpool_fwd_t(const T* x, T* y, int B, int H, int W, int C, wsmg_stream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (H & 1) || (W & 1)) return WSMG_EINVAL;
  hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(sgrid((int64_t)B * H * W * C / 16)), dim3(256), 0, wsmg_s(stream), x, y,
                     B, H, W, C);
  WSMG_RETURN_LAUNCH();
}
template <class T>
int avgpool_bwd_t(const T* dy, T* dx, int B, int H, int W, int C, wsmg_stream_t stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (H & 1) || (W & 1)) return WSMG_EINVAL;
  hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(sgrid((int64_t)B * H * W * C / 4)), dim3(256), 0, wsmg_s(stream), dy, dx, B,
                     H, W, C);
  WSMG_RETURN_LAUNCH();
}
template <bool TO_NHWC, class TI, class TO>
int transpose_t(const TI* x, TO* y, int B, int C_src, int H, int W, int C_dst, wsmg_stream_t stream) {
  if (B <= 0 || C_src <= 0 || C_dst <= 0 || H <= 0 || W <= 0 || B > 65535) return WSMG_EINVAL;
  int HW = H * W;
  if constexpr (TO_NHWC && std::is_same<TI, float>::value) {
    if (C_src == C_dst && C_src % 64 == 0 && HW % 4 == 0) {
      dim3 g64((unsigned)wsmg_cdiv(HW, 64), (unsigned)(C_src / 64), (unsigned)B);
      hipLaunchKernelGGL((nchw_to_nhwc64_kernel<TO>), g64, dim3(256), 0, wsmg_s(stream), x, y, C_src, HW, C_dst);
      WSMG_RETURN_LAUNCH();
    }
  }
  dim3 grid((unsigned)wsmg_cdiv(HW, 32), (unsigned)wsmg_cdiv(TO_NHWC ? C_dst : (C_dst > C_src ? C_dst : C_src), 32), (unsigned)B);
  hipLaunchKernelGGL((transpose_kernel<TO_NHWC, TI, TO>), grid, dim3(256), 0, wsmg_s(stream), x, y, C_src, HW, C_dst);
  WSMG_RETURN_LAUNCH();
}

}  // namespace

#define B16(p) ((bf16_t*)(p))
#define CB16(p) ((const bf16_t*)(p))
extern "C" int wsmg_cat_channels(const void* a, const void* b, void* y, int64_t rows, int bytes_a, int bytes_b, wsmg_stream_t s) {
  if (rows <= 0 || bytes_a <= 0 || bytes_b <= 0 || (bytes_a & 15) || (bytes_b & 15)) return WSMG_EINVAL;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15) return WSMG_EINVAL;
  const int ca = bytes_a / 16, cb = bytes_b / 16;
  hipLaunchKernelGGL(cat_channels_kernel, dim3(sgrid(rows * (ca + cb))), dim3(256), 0, wsmg_s(s), (const u32x4_t*)a, (const u32x4_t*)b,
                     (u32x4_t*)y, rows, ca, cb);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_upsample2x_cat_bf16(const void* a, const void* b, void* y, int B, int H, int W, int Ca, int Cb, wsmg_stream_t s) {
  if (B <= 0 || H <= 0 || W <= 0 || Ca <= 0 || Cb <= 0 || (Ca & 7) || (Cb & 7)) return WSMG_EINVAL;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15) return WSMG_EINVAL;
  hipLaunchKernelGGL(upsample_cat_bf16_kernel, dim3(sgrid((int64_t)B * 4 * H * W * (Ca + Cb) / 8)), dim3(256), 0, wsmg_s(s), CB16(a),
                     (const u32x4_t*)b, (u32x4_t*)y, B, H, W, Ca / 8, Cb / 8);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_add3_bf16(const void* a, const void* b, const void* c, void* y, int64_t n, wsmg_stream_t s) {
  if (!a || !b || !c || !y || n <= 0 || (n & 7) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)y) & 15)) return WSMG_EINVAL;
  hipLaunchKernelGGL(add3_bf16_kernel, dim3(sgrid(n / 8)), dim3(256), 0, wsmg_s(s), (const u32x4_t*)a, (const u32x4_t*)b, (const u32x4_t*)c,
                     (u32x4_t*)y, n / 8);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_relu_fwd(const float* x, float* y, int64_t n, wsmg_stream_t s) { return relu_fwd_t<float>(x, y, n, s); }
extern "C" int wsmg_relu_fwd_bf16(const void* x, void* y, int64_t n, wsmg_stream_t s) { return relu_fwd_t<bf16_t>(CB16(x), B16(y), n, s); }
extern "C" int wsmg_token_grad_merge(float* dx, const float* x, const float* g, int B, int I, int C, int relu, wsmg_stream_t s) {
  return token_grad_merge_t<float>(dx, x, g, B, I, C, relu, s);
}
extern "C" int wsmg_token_grad_merge_bf16(void* dx, const void* x, const float* g, int B, int I, int C, int relu, wsmg_stream_t s) {
  return token_grad_merge_t<bf16_t>(B16(dx), CB16(x), g, B, I, C, relu, s);
}
extern "C" int wsmg_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, wsmg_stream_t s) { return relu_bwd_t<float>(dy, y, dx, n, s); }
extern "C" int wsmg_relu_bwd_bf16(const void* dy, const void* y, void* dx, int64_t n, wsmg_stream_t s) { return relu_bwd_t<bf16_t>(CB16(dy), CB16(y), B16(dx), n, s); }
extern "C" int wsmg_relu_bwd_rows_bf16(const void* dy, int64_t ld_dy, const void* y, void* dx, int64_t rows, int C, wsmg_stream_t s) {
  if (rows <= 0 || C <= 0 || (C & 7) || ld_dy < C || (ld_dy & 7) || (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx) & 15)) return WSMG_EINVAL;
  hipLaunchKernelGGL(relu_bwd8_rows_kernel, dim3(sgrid(rows * C / 8)), dim3(256), 0, wsmg_s(s), (const u32x4_t*)dy, ld_dy / 8,
                     (const u32x4_t*)y, (u32x4_t*)dx, C / 8, rows * C / 8);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_maxpool3x3s2_fwd(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t s) { return maxpool_fwd_t<float>(x, y, B, H, W, C, OH, OW, s); }
extern "C" int wsmg_maxpool3x3s2_fwd_bf16(const void* x, void* y, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t s) { return maxpool_fwd_t<bf16_t>(CB16(x), B16(y), B, H, W, C, OH, OW, s); }
extern "C" int wsmg_maxpool3x3s2_fwd_idx_bf16(const void* x, void* y, uint32_t* idx, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t s) {
  if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1 || B <= 0 || C <= 0 || (C & 3) || !idx) return WSMG_EINVAL;
  hipLaunchKernelGGL(maxpool_fwd_idx_kernel<bf16_t>, dim3(sgrid((int64_t)B * OH * OW * C / 4)), dim3(256), 0, wsmg_s(s), CB16(x), B16(y), idx,
                     B, H, W, C, OH, OW);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_maxpool3x3s2_bwd_idx_bf16(const void* dy, const uint32_t* idx, void* dx, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t s) {
  if (OH != (H + 2 - 3) / 2 + 1 || OW != (W + 2 - 3) / 2 + 1 || B <= 0 || C <= 0 || (C & 3) || !idx) return WSMG_EINVAL;
  hipLaunchKernelGGL(maxpool_bwd_idx_kernel<bf16_t>, dim3(sgrid((int64_t)B * H * W * C / 4)), dim3(256), 0, wsmg_s(s), CB16(dy), idx, B16(dx),
                     B, H, W, C, OH, OW);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_maxpool3x3s2_bwd(const float* dy, const float* x, float* dx, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t s) { return maxpool_bwd_t<float>(dy, x, dx, B, H, W, C, OH, OW, s); }
extern "C" int wsmg_maxpool3x3s2_bwd_bf16(const void* dy, const void* x, void* dx, int B, int H, int W, int C, int OH, int OW, wsmg_stream_t s) { return maxpool_bwd_t<bf16_t>(CB16(dy), CB16(x), B16(dx), B, H, W, C, OH, OW, s); }
extern "C" int wsmg_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, wsmg_stream_t s) { return upsample_fwd_t<float>(x, y, B, H, W, C, s); }
extern "C" int wsmg_upsample2x_fwd_bf16(const void* x, void* y, int B, int H, int W, int C, wsmg_stream_t s) { return upsample_fwd_t<bf16_t>(CB16(x), B16(y), B, H, W, C, s); }
extern "C" int wsmg_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, wsmg_stream_t s) { return upsample_bwd_t<float>(dy, dx, B, H, W, C, s); }
extern "C" int wsmg_upsample2x_bwd_bf16(const void* dy, void* dx, int B, int H, int W, int C, wsmg_stream_t s) { return upsample_bwd_t<bf16_t>(CB16(dy), B16(dx), B, H, W, C, s); }
extern "C" int wsmg_upsample2x_bwd_ld(const float* dy, int64_t ld_dy, float* dx, int B, int H, int W, int C, wsmg_stream_t s) { return upsample_bwd_t<float>(dy, dx, B, H, W, C, s, ld_dy); }
extern "C" int wsmg_upsample2x_bwd_ld_bf16(const void* dy, int64_t ld_dy, void* dx, int B, int H, int W, int C, wsmg_stream_t s) { return upsample_bwd_t<bf16_t>(CB16(dy), B16(dx), B, H, W, C, s, ld_dy); }
extern "C" int wsmg_avgpool2_fwd(const float* x, float* y, int B, int H, int W, int C, wsmg_stream_t s) { return avgpool_fwd_t<float>(x, y, B, H, W, C, s); }
extern "C" int wsmg_avgpool2_fwd_bf16(const void* x, void* y, int B, int H, int W, int C, wsmg_stream_t s) { return avgpool_fwd_t<bf16_t>(CB16(x), B16(y), B, H, W, C, s); }
extern "C" int wsmg_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, wsmg_stream_t s) { return avgpool_bwd_t<float>(dy, dx, B, H, W, C, s); }
extern "C" int wsmg_avgpool2_bwd_bf16(const void* dy, void* dx, int B, int H, int W, int C, wsmg_stream_t s) { return avgpool_bwd_t<bf16_t>(CB16(dy), B16(dx), B, H, W, C, s); }
extern "C" int wsmg_nchw_to_nhwc(const float* x, float* y, int B, int C_src, int H, int W, int C_dst, wsmg_stream_t s) { return transpose_t<true, float, float>(x, y, B, C_src, H, W, C_dst, s); }
extern "C" int wsmg_nchw_to_nhwc_bf16(const float* x, void* y, int B, int C_src, int H, int W, int C_dst, wsmg_stream_t s) { return transpose_t<true, float, bf16_t>(x, B16(y), B, C_src, H, W, C_dst, s); }
extern "C" int wsmg_nhwc_to_nchw(const float* x, float* y, int B, int C_src, int H, int W, int C_dst, wsmg_stream_t s) { return transpose_t<false, float, float>(x, y, B, C_src, H, W, C_dst, s); }
extern "C" int wsmg_nhwc_to_nchw_bf16(const void* x, float* y, int B, int C_src, int H, int W, int C_dst, wsmg_stream_t s) { return transpose_t<false, bf16_t, float>(CB16(x), y, B, C_src, H, W, C_dst, s); }

extern "C" int wsmg_ce_nhwc_fwd(const float* logits, const int64_t* target, int64_t rows, int classes, float* loss, wsmg_stream_t s) {
  if (rows <= 0 || classes <= 0 || classes > 32) return WSMG_EINVAL;
  hipLaunchKernelGGL(ce_nhwc_fwd_kernel<float>, dim3((unsigned)wsmg_cdiv(rows, 256)), dim3(256), 0, wsmg_s(s), logits, target, rows, classes, loss);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_ce_nhwc_fwd_bf16(const void* logits, const int64_t* target, int64_t rows, int classes, float* loss, wsmg_stream_t s) {
  if (rows <= 0 || classes <= 0 || classes > 32) return WSMG_EINVAL;
  hipLaunchKernelGGL(ce_nhwc_fwd_kernel<bf16_t>, dim3((unsigned)wsmg_cdiv(rows, 256)), dim3(256), 0, wsmg_s(s), CB16(logits), target, rows, classes, loss);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_ce_nhwc_bwd(const float* logits, const int64_t* target, const float* gloss, int64_t rows, int classes, float* dlogits, wsmg_stream_t s) {
  if (rows <= 0 || classes <= 0 || classes > 32) return WSMG_EINVAL;
  hipLaunchKernelGGL(ce_nhwc_bwd_kernel<float>, dim3((unsigned)wsmg_cdiv(rows, 256)), dim3(256), 0, wsmg_s(s), logits, target, gloss, rows, classes, dlogits);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_ce_nhwc_bwd_bf16(const void* logits, const int64_t* target, const float* gloss, int64_t rows, int classes, void* dlogits, wsmg_stream_t s) {
  if (rows <= 0 || classes <= 0 || classes > 32) return WSMG_EINVAL;
  hipLaunchKernelGGL(ce_nhwc_bwd_kernel<bf16_t>, dim3((unsigned)wsmg_cdiv(rows, 256)), dim3(256), 0, wsmg_s(s), CB16(logits), target, gloss, rows, classes, B16(dlogits));
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_weight_relayout(const float* w_oihw, int O, int I, int KH, int KW, int I_pad, float* w_ohwi, float* w_ihwo,
                                    wsmg_stream_t s) {
  if (O <= 0 || I <= 0 || KH <= 0 || KW <= 0 || I_pad < I) return WSMG_EINVAL;
  hipLaunchKernelGGL(weight_relayout_kernel<float>, dim3(sgrid((int64_t)O * KH * KW * I_pad)), dim3(256), 0, wsmg_s(s), w_oihw, O, I, KH,
                     KW, I_pad, w_ohwi, w_ihwo);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_weight_relayout_bf16(const float* w_oihw, int O, int I, int KH, int KW, int I_pad, void* w_ohwi, void* w_ihwo,
                                         wsmg_stream_t s) {
  if (O <= 0 || I <= 0 || KH <= 0 || KW <= 0 || I_pad < I) return WSMG_EINVAL;
  hipLaunchKernelGGL(weight_relayout_kernel<bf16_t>, dim3(sgrid((int64_t)O * KH * KW * I_pad)), dim3(256), 0, wsmg_s(s), w_oihw, O, I,
                     KH, KW, I_pad, B16(w_ohwi), B16(w_ihwo));
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_weight_relayout_multi(const WsmgRelayoutDesc* descs, int n, int bf16, wsmg_stream_t s) {
  if (!descs || n <= 0) return WSMG_EINVAL;
  for (int i0 = 0; i0 < n; i0 += WSMG_RELAYOUT_MAX) {
    RelayoutBatch b;
    const int m = n - i0 < WSMG_RELAYOUT_MAX ? n - i0 : WSMG_RELAYOUT_MAX;
    for (int i = 0; i < m; ++i) {
      b.d[i] = descs[i0 + i];
      if (b.d[i].O <= 0 || b.d[i].I <= 0 || b.d[i].KH <= 0 || b.d[i].KW <= 0 || b.d[i].I_pad < b.d[i].I || !b.d[i].w_oihw || !b.d[i].w_ohwi)
        return WSMG_EINVAL;
    }
    if (bf16) hipLaunchKernelGGL(weight_relayout_multi_kernel<bf16_t>, dim3(48, (unsigned)m), dim3(256), 0, wsmg_s(s), b);
    else hipLaunchKernelGGL(weight_relayout_multi_kernel<float>, dim3(48, (unsigned)m), dim3(256), 0, wsmg_s(s), b);
  }
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_mean_rows(const float* x, int64_t R, int n, float* out, wsmg_stream_t s) {
  if (!x || !out || R <= 0 || n <= 0 || n > 160) return WSMG_EINVAL;       // 256 rows x n floats of LDS
  hipLaunchKernelGGL(mean_rows_kernel, dim3((unsigned)wsmg_cdiv(R, 256)), dim3(256), (size_t)256 * n * sizeof(float), wsmg_s(s), x, R, n, out);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_weight_grad_reduce_oihw(const float* ws, int nsplit, int O, int I, int KH, int KW, int I_pad, float* dw_oihw,
                                            wsmg_stream_t s) {
  if (!ws || !dw_oihw || nsplit <= 0 || O <= 0 || I <= 0 || KH <= 0 || KW <= 0 || I_pad < I || (I_pad & 3)) return WSMG_EINVAL;
  const int64_t slab4 = (int64_t)O * KH * KW * I_pad / 4;
  hipLaunchKernelGGL(weight_grad_reduce_oihw_kernel, dim3((unsigned)wsmg_cdiv(slab4, 64)), dim3(256), 0, wsmg_s(s), ws, nsplit, slab4, I,
                     KH, KW, I_pad, dw_oihw);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_weight_grad_to_oihw(const float* dw_ohwi, int O, int I, int KH, int KW, int I_pad, float* dw_oihw, wsmg_stream_t s) {
  if (O <= 0 || I <= 0 || KH <= 0 || KW <= 0 || I_pad < I) return WSMG_EINVAL;
  hipLaunchKernelGGL(weight_grad_to_oihw_kernel, dim3(sgrid((int64_t)O * I * KH * KW)), dim3(256), 0, wsmg_s(s), dw_ohwi, O, I, KH, KW,
                     I_pad, dw_oihw);
  WSMG_RETURN_LAUNCH();
}
