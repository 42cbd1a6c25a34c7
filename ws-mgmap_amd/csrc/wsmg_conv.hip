// Map conv engine: implicit-GEMM convolution forward / backward-data / backward-weight for
// NHWC float32 tensors on the gfx950 f32 matrix core (v_mfma_f32_32x32x2_f32: exact f32
// products, k-ordered f32 accumulation — bitwise an fmaf chain).
//
// Replaces the cuDNN conv2d calls behind nn.Conv2d / nn.ConvTranspose2d / nn.Conv1d(k=1) at
// map_encoder.py:19-29,94-112 and mg_map_policy.py:78-100,127,130 of the reference.
//
// GEMM view (forward):   Y[m][n] = sum_k A[m][k] * Wt[n][k]
//     m = (b, oy, ox)   n = cout   k = (ky, kx, cin)   A[m][k] = X[b][oy*s-p+ky][ox*s-p+kx][cin]
// The A tile is gathered straight from the NHWC tensor (one 128-B channel run per pixel and
// tap); weights are OHWI so a Wt row is contiguous in k.  Backward-data is the same kernel
// with the roles of the two pixel grids swapped and IHWO weights.  Backward-weight reduces
// over pixels:  dW[co][(tap,ci)] = sum_px dY[px][co] * X[px@tap][ci], split over pixel
// ranges (grid.z) and combined with float atomics.
#include <stdlib.h>

#include "wsmg_common.h"

namespace {

constexpr int BM = 128;   // pixels per workgroup tile
constexpr int BN = 64;    // output channels per workgroup tile
constexpr int BK = 32;    // channels per k-step
constexpr int LDT = BK + 4;  // LDS row stride (floats): 144 B keeps ds_read_b128 conflict-free

struct ConvArgs {
  const float* src;   // [B][SH][SW][Kc]
  const float* wt;    // [N][KH][KW][Kc]
  const float* bias;  // [N] or null
  float* dst;         // [B][TH][TW][N]
  int B, SH, SW, Kc, TH, TW, N, KH, KW, stride, pad;
  int mtiles;         // pixel tiles per parity class (grid.x = mtiles * ntiles)
  int ntiles;
};

// XCD-aware, bijective remap of a 1-D block id: the dispatcher deals blocks round-robin over the
// 8 XCDs, so give each XCD one contiguous run of logical tiles (tiles that share input windows /
// the same A rows then share one L2).  Speed only — any placement is correct.
__device__ __forceinline__ int xcd_swizzle(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}

// BWD=false: target pixel t reads source t*s - p + k.
// BWD=true : source (t + p - k)/s where divisible.  For s == 2 the target pixels are split into
//            the 4 parity classes (blockIdx.y) so that every tap a class visits is valid:
//            class (cy,cx) owns targets with (t+p)&1 == c and taps k = c + 2a, source = (t+p-c)/2 - a.
// DB=true : two LDS buffers, one barrier per k-step (2 workgroups per CU by LDS);
// DB=false: one LDS buffer, two barriers per k-step (3 workgroups per CU).
template <bool BWD, bool DB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float As[DB ? 2 : 1][BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[DB ? 2 : 1][BN * LDT];
  __shared__ int dpix[BM];  // destination pixel index of each tile row (-1 = none)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int logical = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (logical / a.ntiles) * BM, n0 = (logical % a.ntiles) * BN;
  const int seg = tid & 7;          // 16-B segment inside a 128-B channel run
  const int lrow = tid >> 3;        // 0..31

  // parity class geometry (identity unless BWD && stride == 2)
  int cy = 0, cx = 0, ty0 = 0, tx0 = 0, tstep = 1, THc = a.TH, TWc = a.TW, KHc = a.KH, KWc = a.KW;
  if (BWD && a.stride == 2) {
    cy = blockIdx.y >> 1; cx = blockIdx.y & 1;
    ty0 = (cy + a.pad) & 1; tx0 = (cx + a.pad) & 1;   // first target row/col with (t + pad) & 1 == c
    tstep = 2;
    THc = (a.TH - ty0 + 1) >> 1; TWc = (a.TW - tx0 + 1) >> 1;
    KHc = (a.KH - cy + 1) >> 1; KWc = (a.KW - cx + 1) >> 1;
  }
  const int Mc = a.B * THc * TWc;
  if (m0 >= Mc) return;  // uniform per workgroup (classes differ in size)

  // the 4 A rows this thread stages (fixed over the k loop)
  int pb[4], py[4], px[4];
  bool pv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + lrow + 32 * i;
    pv[i] = m < Mc;
    int mm = pv[i] ? m : 0;
    int b = mm / (THc * TWc);
    int r = mm - b * (THc * TWc);
    int iy = r / TWc, ix = r - iy * TWc;
    int ty = ty0 + iy * tstep, tx = tx0 + ix * tstep;
    pb[i] = b * a.SH * a.SW;
    if (BWD) {
      py[i] = (a.stride == 2) ? ((ty + a.pad - cy) >> 1) : ty + a.pad;
      px[i] = (a.stride == 2) ? ((tx + a.pad - cx) >> 1) : tx + a.pad;
    } else {
      py[i] = ty * a.stride - a.pad;
      px[i] = tx * a.stride - a.pad;
    }
    if (seg == 0) dpix[lrow + 32 * i] = pv[i] ? (b * a.TH + ty) * a.TW + tx : -1;
  }
  const int kchunks = a.Kc / BK;
  const int taps_c = KHc * KWc;
  const int steps = taps_c * kchunks;
  const int taps = a.KH * a.KW;

  f32x4 ra[4], rb[2];
  auto gload = [&](int step) {
    int tc = step / kchunks;
    int c0 = (step - tc * kchunks) * BK + seg * 4;
    int ay = tc / KWc, ax = tc - ay * KWc;
    int ky = BWD ? cy + ay * tstep : ay, kx = BWD ? cx + ax * tstep : ax;
    int tap = ky * a.KW + kx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int sy = BWD ? py[i] - ((a.stride == 2) ? ay : ky) : py[i] + ky;
      int sx = BWD ? px[i] - ((a.stride == 2) ? ax : kx) : px[i] + kx;
      bool ok = pv[i] && sy >= 0 && sy < a.SH && sx >= 0 && sx < a.SW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(a.src + ((size_t)(pb[i] + sy * a.SW + sx)) * a.Kc + c0);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int n = n0 + lrow + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < a.N) v = *reinterpret_cast<const f32x4*>(a.wt + ((size_t)n * taps + tap) * a.Kc + c0);
      rb[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * i) * LDT + seg * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * i) * LDT + seg * 4]) = rb[i];
  };

  // wave tile: 64 pixels x 32 channels = two 32x32 MFMA tiles
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
  const int r = lane & 31, h = lane >> 5;
  const bool wave_live = (n0 + wn) < a.N;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  // two LDS buffers, ONE barrier per k-step: while buffer `cur` feeds the MFMAs, the registers
  // prefetched during the previous step are written to the other buffer and the step after
  // that is requested from memory.
  gload(0);
  lstore(0);
  if (DB && steps > 1) gload(1);
  __syncthreads();
  for (int step = 0; step < steps; ++step) {
    const int cur = DB ? (step & 1) : 0;
    if (DB) {
      if (step + 1 < steps) lstore(cur ^ 1);
      if (step + 2 < steps) gload(step + 2);
    } else {
      if (step + 1 < steps) gload(step + 1);
    }
    if (wave_live) {
      const float* Ab = As[cur];
      const float* Bb = Bs[cur];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // lane half h supplies k = 8j + 4h + e of this k-step (same map for A and B)
        f32x4 a0 = *reinterpret_cast<const f32x4*>(&Ab[(wm + r) * LDT + 8 * j + 4 * h]);
        f32x4 a1 = *reinterpret_cast<const f32x4*>(&Ab[(wm + 32 + r) * LDT + 8 * j + 4 * h]);
        f32x4 bb = *reinterpret_cast<const f32x4*>(&Bb[(wn + r) * LDT + 8 * j + 4 * h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bb[e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bb[e], acc1, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (!DB && step + 1 < steps) {
      lstore(0);
      __syncthreads();
    }
  }

  if (!wave_live) return;
  const int n = n0 + wn + r;
  if (n >= a.N) return;
  const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      int row = wm + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
      int dp = dpix[row];
      if (dp >= 0) a.dst[(size_t)dp * a.N + n] = (t == 0 ? acc0[g] : acc1[g]) + bv;
    }
  }
}

// ----------------------------------------------------------------------------- backward weight
constexpr int WPX = 32;   // pixels per k-step
constexpr int WCO = 64;   // output channels per workgroup
constexpr int WUN = 4;    // (tap, 32-channel chunk) units per workgroup = one per wave

struct WgradArgs {
  const float* x;    // [B][H][W][Cin]
  const float* dy;   // [B][OH][OW][Cout]
  float* dw;         // [Cout][KH][KW][Cin]
  int B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW;
  int units;         // KH*KW*(Cin/32)
  int64_t npix;      // B*OH*OW
  int64_t chunk;     // pixels per pixel-range slice (multiple of WPX)
  int gx, gy;        // unit tiles, co tiles (1-D grid of gx*gy*gz blocks, XCD-swizzled)
  int64_t slab;      // > 0: floats per slab — `dw` is a workspace [gz][Cout][KH][KW][Cin]; the workgroup of pixel chunk z STORES its
                     // partial tile into slab z (no atomics, nothing to zero); wsmg_weight_grad_reduce_oihw adds the slabs in order
};

template <bool DB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float Ds[DB ? 2 : 1][WPX * WCO];
  __shared__ __attribute__((aligned(16))) float Xs[DB ? 2 : 1][WUN * WPX * 32];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // all tiles of one pixel slice are consecutive logical blocks => one XCD / one L2 per slice
  const int logical = xcd_swizzle(blockIdx.x, gridDim.x);
  const int gxy = a.gx * a.gy;
  const int bz = logical / gxy, bxy = logical - bz * gxy;
  const int u0 = (bxy % a.gx) * WUN;
  const int co0 = (bxy / a.gx) * WCO;
  const int64_t p_begin = (int64_t)bz * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > a.npix) p_end = a.npix;
  const int cchunks = a.Cin / 32;
  const int ohw = a.OH * a.OW;

  // unit of each X staging slot (every thread stages one 16-B piece per unit and step)
  int uky[WUN], ukx[WUN], uci[WUN];
  bool uok[WUN];
#pragma unroll
  for (int u = 0; u < WUN; ++u) {
    int uu = u0 + u;
    uok[u] = uu < a.units;
    int t = uok[u] ? uu / cchunks : 0;
    uci[u] = uok[u] ? (uu - t * cchunks) * 32 : 0;
    uky[u] = t / a.KW;
    ukx[u] = t - uky[u] * a.KW;
  }
  const int xpx = tid >> 3, xseg = tid & 7;     // X staging: pixel 0..31, 16-B segment 0..7
  const int dseg = tid & 15;                     // dY staging: 16 segments per 256-B row

  const int r = lane & 31, h = lane >> 5;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  f32x4 rd[2], rx[WUN];
  auto gload = [&](int64_t p0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t p = p0 + (tid >> 4) + 16 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      int co = co0 + dseg * 4;
      if (p < p_end && co < a.Cout) v = *reinterpret_cast<const f32x4*>(a.dy + (size_t)p * a.Cout + co);
      rd[i] = v;
    }
    int64_t p = p0 + xpx;
    bool pok = p < p_end;
    int64_t pp = pok ? p : 0;
    int b = (int)(pp / ohw);
    int rem = (int)(pp - (int64_t)b * ohw);
    int oy = rem / a.OW, ox = rem - oy * a.OW;
#pragma unroll
    for (int u = 0; u < WUN; ++u) {
      int iy = oy * a.stride - a.pad + uky[u], ix = ox * a.stride - a.pad + ukx[u];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pok && uok[u] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
        v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)(b * a.H + iy) * a.W + ix) * a.Cin + uci[u] + xseg * 4);
      rx[u] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&Ds[buf][((tid >> 4) + 16 * i) * WCO + dseg * 4]) = rd[i];
#pragma unroll
    for (int u = 0; u < WUN; ++u) *reinterpret_cast<f32x4*>(&Xs[buf][(u * WPX + xpx) * 32 + xseg * 4]) = rx[u];
  };

  // same pipeline choice as conv_igemm_kernel (DB: two LDS buffers, one barrier per step)
  if (p_begin < p_end) {
    gload(p_begin);
    lstore(0);
    if (DB && p_begin + WPX < p_end) gload(p_begin + WPX);
  }
  __syncthreads();
  const bool wave_live = (u0 + wave) < a.units;
  int cur = 0;
  for (int64_t p0 = p_begin; p0 < p_end; p0 += WPX, cur ^= (DB ? 1 : 0)) {
    if (DB) {
      if (p0 + WPX < p_end) lstore(cur ^ 1);
      if (p0 + 2 * WPX < p_end) gload(p0 + 2 * WPX);
    } else {
      if (p0 + WPX < p_end) gload(p0 + WPX);
    }
    if (wave_live) {
      const float* ds = Ds[cur];
      const float* xs = &Xs[cur][wave * WPX * 32];
#pragma unroll
      for (int s = 0; s < WPX / 2; ++s) {
        float a0 = ds[(2 * s + h) * WCO + r];
        float a1 = ds[(2 * s + h) * WCO + 32 + r];
        float bb = xs[(2 * s + h) * 32 + r];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb, acc1, 0, 0, 0);
      }
    }
    __syncthreads();
    if (!DB && p0 + WPX < p_end) {
      lstore(0);
      __syncthreads();
    }
  }
  if (!wave_live) return;
  const int uu = u0 + wave;
  const int tap = uu / cchunks;
  const int ci = (uu - tap * cchunks) * 32 + r;
  const int taps = a.KH * a.KW;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      int co = co0 + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
      if (co < a.Cout) {
        float* const q = a.dw + (size_t)bz * a.slab + ((size_t)co * taps + tap) * a.Cin + ci;
        if (a.slab) *q = t == 0 ? acc0[g] : acc1[g];
        else atomicAdd(q, t == 0 ? acc0[g] : acc1[g]);
      }
    }
  }
}

// pipeline variant: WSMG_CONV_DB=0/1 (tuning knob; default set from measurements)
bool conv_double_buffer() { return (0) != 0; }

int check_conv(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return WSMG_EINVAL;
  if (Cin % 32 || Cout % 32) return WSMG_EINVAL;
  if (stride != 1 && stride != 2) return WSMG_EINVAL;
  if (OH != (H + 2 * pad - KH) / stride + 1 || OW != (W + 2 * pad - KW) / stride + 1) return WSMG_EINVAL;
  if ((int64_t)B * H * W >= (1ll << 31) || (int64_t)B * OH * OW >= (1ll << 31)) return WSMG_EINVAL;
  return 0;
}

}  // namespace

extern "C" int wsmg_conv2d_fwd(const float* x, const float* w_ohwi, const float* bias, float* y, int B, int H,
                               int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                               wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  ConvArgs a{x, w_ohwi, bias, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, 0, 0};
  a.mtiles = (int)wsmg_cdiv((int64_t)B * OH * OW, BM);
  a.ntiles = (int)wsmg_cdiv(Cout, BN);
  dim3 grid((unsigned)(a.mtiles * a.ntiles));
  if (conv_double_buffer())
    hipLaunchKernelGGL((conv_igemm_kernel<false, true>), grid, dim3(256), 0, wsmg_s(stream), a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<false, false>), grid, dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_conv2d_bwd_data(const float* dy, const float* w_ihwo, float* dx, int B, int H, int W, int Cin,
                                    int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                    wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  ConvArgs a{dy, w_ihwo, nullptr, dx, B, OH, OW, Cout, H, W, Cin, KH, KW, stride, pad, 0, 0};
  a.ntiles = (int)wsmg_cdiv(Cin, BN);
  int classes = 1;
  int64_t mmax = (int64_t)B * H * W;
  if (stride == 2) {  // largest parity class: ceil(H/2) x ceil(W/2) target pixels per image
    classes = 4;
    mmax = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  }
  a.mtiles = (int)wsmg_cdiv(mmax, BM);
  dim3 grid((unsigned)(a.mtiles * a.ntiles), (unsigned)classes);
  if (conv_double_buffer())
    hipLaunchKernelGGL((conv_igemm_kernel<true, true>), grid, dim3(256), 0, wsmg_s(stream), a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<true, false>), grid, dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

namespace {
// split of the pixel reduction: ~2048 workgroups in flight, >= 8 k-steps each
void wgrad_plan_f32(int B, int OH, int OW, int Cin, int Cout, int KH, int KW, int& gx, int& gy, int64_t& chunk, int& gz_out) {
  const int units = KH * KW * (Cin / 32);
  const int64_t npix = (int64_t)B * OH * OW;
  gx = (int)wsmg_cdiv(units, WUN);
  gy = (int)wsmg_cdiv(Cout, WCO);
  int64_t want = wsmg_cdiv(2048, (int64_t)gx * gy);
  int64_t maxz = wsmg_cdiv(npix, WPX * 8);
  int64_t gz = want < 1 ? 1 : (want > maxz ? maxz : want);
  if (gz < 1) gz = 1;
  if (gz > 65535) gz = 65535;
  chunk = wsmg_cdiv(wsmg_cdiv(npix, gz), WPX) * WPX;
  gz_out = (int)wsmg_cdiv(npix, chunk);
}

int launch_wgrad_f32(const float* x, const float* dy, float* dw, long long slab, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                     int stride, int pad, int OH, int OW, hipStream_t stream) {
  WgradArgs a{x, dy, dw, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, 0, 0, 0, 0, 0, (int64_t)slab};
  a.units = KH * KW * (Cin / 32);
  a.npix = (int64_t)B * OH * OW;
  int gz = 1;
  wgrad_plan_f32(B, OH, OW, Cin, Cout, KH, KW, a.gx, a.gy, a.chunk, gz);
  dim3 grid((unsigned)((int64_t)a.gx * a.gy * gz));
  if (conv_double_buffer())
    hipLaunchKernelGGL(conv_wgrad_kernel<true>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<false>, grid, dim3(256), 0, stream, a);
  WSMG_RETURN_LAUNCH();
}
}  // namespace

extern "C" int wsmg_conv2d_bwd_weight(const float* x, const float* dy, float* dw_ohwi, int B, int H, int W, int Cin,
                                      int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                      wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  return launch_wgrad_f32(x, dy, dw_ohwi, 0, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, wsmg_s(stream));
}

// Deterministic weight gradient of the float32 engine: plan / slabs, as the bf16 pair (wsmg_conv2d_bwd_weight_bf16_plan / _slabs)
extern "C" int wsmg_conv2d_bwd_weight_plan(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                           int* nsplit, long long* ws_floats) {
  if (!nsplit || !ws_floats) return WSMG_EINVAL;
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  int gx, gy, gz;
  int64_t chunk;
  wgrad_plan_f32(B, OH, OW, Cin, Cout, KH, KW, gx, gy, chunk, gz);
  *nsplit = gz;
  *ws_floats = (long long)gz * Cout * KH * KW * Cin;
  return 0;
}

extern "C" int wsmg_conv2d_bwd_weight_slabs(const float* x, const float* dy, float* ws, int nsplit, long long ws_floats, int B, int H, int W,
                                            int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  int gx, gy, gz;
  int64_t chunk;
  wgrad_plan_f32(B, OH, OW, Cin, Cout, KH, KW, gx, gy, chunk, gz);
  const long long slab = (long long)Cout * KH * KW * Cin;
  if (!ws || nsplit != gz || ws_floats < slab * gz) return WSMG_EINVAL;
  return launch_wgrad_f32(x, dy, ws, slab, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, wsmg_s(stream));
}
