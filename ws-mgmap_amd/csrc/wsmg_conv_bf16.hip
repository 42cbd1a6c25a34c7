// bf16 map conv engine (BASELINE configs[1]: "Teacher-forcing fwd+bwd ... bf16"): the same
// implicit-GEMM convolution as wsmg_conv.hip with bf16 activations / weights in HBM and LDS,
// v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate) and float32 accumulation.  Master weights
// and weight gradients stay float32 (dW is accumulated with f32 atomics).
//
//   forward / backward-data : 128-pixel x BN-channel tile (BN = 64 or 128), BK = 64 (or 32)
//       channels per k-step, wave tile 64 x BN/2; A rows gathered per tap from NHWC (one 128-B
//       run of 64 bf16 per pixel), LDS rows padded to 144 B / 80 B (conflict-free ds_read_b128);
//       stride-2 backward-data uses the 4 parity classes of wsmg_conv.hip.
//   backward-weight : reduction over pixels needs both operands k(=pixel)-strided; tiles are
//       staged pixel-major exactly as they sit in HBM and read with the gfx950 transposing LDS
//       read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-per-
//       lane) — no transposed copy is ever materialised.
#include <stdlib.h>

#include "wsmg_common.h"

namespace {

typedef __bf16 bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;

__device__ __forceinline__ int xcd_swizzle(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}

__device__ __forceinline__ unsigned short f2bf_bits(float f) {
  bf16_t b = (bf16_t)f;
  return __builtin_bit_cast(unsigned short, b);
}

struct ConvArgsB {
  const bf16_t* src;  // [B][SH][SW][Kc]
  const bf16_t* wt;   // [N][KH][KW][Kc]
  const float* bias;  // [N] or null
  void* dst;          // [B][TH][TW][N] bf16 or f32
  int B, SH, SW, Kc, TH, TW, N, KH, KW, stride, pad;
  int mtiles, ntiles;
  int out_f32;
  int tap_inner;   // 1: k-steps walk the taps of one channel chunk back to back (shifted re-reads hit L1)
};

// PF = k-steps of global loads kept in flight in registers (a bf16 k-step is only 256-512 MFMA
// cycles, far less than the L2/HBM latency, so one step of look-ahead leaves the loads exposed).
template <bool BWD, int BN, int BK, int PF>
__global__ __launch_bounds__(256) void conv_igemm_bf16_kernel(ConvArgsB a) {
  constexpr int LDB = BK * 2 + 16;           // LDS row stride in bytes
  constexpr int SEGS = BK / 8;               // 16-B segments per row (8 bf16 each)
  constexpr int RPP = 256 / SEGS;            // rows staged per pass
  constexpr int APASS = BM / RPP;            // A passes per thread
  constexpr int BPASS = BN / RPP;            // B passes per thread
  constexpr int NT = BN / 64;                // 32-wide n-tiles per wave
  __shared__ __attribute__((aligned(16))) unsigned char As[BM * LDB];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * LDB];
  __shared__ int dpix[BM];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int logical = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (logical / a.ntiles) * BM, n0 = (logical % a.ntiles) * BN;
  const int seg = tid % SEGS;
  const int lrow = tid / SEGS;

  int cy = 0, cx = 0, ty0 = 0, tx0 = 0, tstep = 1, THc = a.TH, TWc = a.TW, KHc = a.KH, KWc = a.KW;
  if (BWD && a.stride == 2) {
    cy = blockIdx.y >> 1; cx = blockIdx.y & 1;
    ty0 = (cy + a.pad) & 1; tx0 = (cx + a.pad) & 1;
    tstep = 2;
    THc = (a.TH - ty0 + 1) >> 1; TWc = (a.TW - tx0 + 1) >> 1;
    KHc = (a.KH - cy + 1) >> 1; KWc = (a.KW - cx + 1) >> 1;
  }
  const int Mc = a.B * THc * TWc;
  if (m0 >= Mc) return;

  int pb[APASS], py[APASS], px[APASS];
  bool pv[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int m = m0 + lrow + RPP * i;
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
    if (seg == 0) dpix[lrow + RPP * i] = pv[i] ? (b * a.TH + ty) * a.TW + tx : -1;
  }
  const int kchunks = a.Kc / BK;
  const int steps = KHc * KWc * kchunks;
  const int taps = a.KH * a.KW;

  u32x4 ra[PF][APASS], rb[PF][BPASS];
  const int ntap_c = KHc * KWc;
  auto gload = [&](int step, u32x4 (&ra)[APASS], u32x4 (&rb)[BPASS]) {
    int tc, ch;
    if (a.tap_inner) { ch = step / ntap_c; tc = step - ch * ntap_c; }
    else { tc = step / kchunks; ch = step - tc * kchunks; }
    int c0 = ch * BK + seg * 8;
    int ay = tc / KWc, ax = tc - ay * KWc;
    int ky = BWD ? cy + ay * tstep : ay, kx = BWD ? cx + ax * tstep : ax;
    int tap = ky * a.KW + kx;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      int sy = BWD ? py[i] - ((a.stride == 2) ? ay : ky) : py[i] + ky;
      int sx = BWD ? px[i] - ((a.stride == 2) ? ax : kx) : px[i] + kx;
      bool ok = pv[i] && sy >= 0 && sy < a.SH && sx >= 0 && sx < a.SW;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(a.src + ((size_t)(pb[i] + sy * a.SW + sx)) * a.Kc + c0);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      int n = n0 + lrow + RPP * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (n < a.N) v = *reinterpret_cast<const u32x4*>(a.wt + ((size_t)n * taps + tap) * a.Kc + c0);
      rb[i] = v;
    }
  };
  auto lstore = [&](const u32x4 (&ra)[APASS], const u32x4 (&rb)[BPASS]) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) *reinterpret_cast<u32x4*>(&As[(lrow + RPP * i) * LDB + seg * 16]) = ra[i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *reinterpret_cast<u32x4*>(&Bs[(lrow + RPP * i) * LDB + seg * 16]) = rb[i];
  };

  // wave tile: 64 pixels x BN/2 channels
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * (BN / 2);
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

  // register set j holds the k-step with (step % PF == j); LDS holds the current step
  gload(0, ra[0], rb[0]);
  lstore(ra[0], rb[0]);
  __syncthreads();
#pragma unroll
  for (int j = 1; j < PF; ++j)
    if (j < steps) gload(j, ra[j], rb[j]);
  for (int step0 = 0; step0 < steps; step0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int step = step0 + j;
      if (step < steps) {
        if (step + PF < steps) gload(step + PF, ra[j], rb[j]);  // set j is free: its step is in LDS
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
          // lane (r, h) holds k = 16*ks + 8h .. +7 of its row: one 16-B LDS read per operand
          bf16x8 af[2], bf[NT];
#pragma unroll
          for (int t = 0; t < 2; ++t)
            af[t] = *reinterpret_cast<const bf16x8*>(&As[(wm + 32 * t + r) * LDB + ks * 32 + h * 16]);
#pragma unroll
          for (int u = 0; u < NT; ++u)
            bf[u] = *reinterpret_cast<const bf16x8*>(&Bs[(wn + 32 * u + r) * LDB + ks * 32 + h * 16]);
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
              acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], bf[u], acc[t][u], 0, 0, 0);
        }
        __syncthreads();
        if (step + 1 < steps) {
          lstore(ra[(j + 1) % PF], rb[(j + 1) % PF]);
          __syncthreads();
        }
      }
    }
  }

#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int n = n0 + wn + 32 * u + r;
    if (n >= a.N) continue;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int row = wm + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
        int dp = dpix[row];
        if (dp < 0) continue;
        float v = acc[t][u][g] + bv;
        if (a.out_f32)
          reinterpret_cast<float*>(a.dst)[(size_t)dp * a.N + n] = v;
        else
          reinterpret_cast<unsigned short*>(a.dst)[(size_t)dp * a.N + n] = f2bf_bits(v);
      }
    }
  }
}

// ----------------------------------------------------------------------------- backward weight
constexpr int WKP = 64;          // pixels per k-step
constexpr int WCO = 64;          // output channels per workgroup
constexpr int WUW = 1;           // (tap, 32-channel) units per wave (dY fragments reused across them)
constexpr int WUN = 4 * WUW;     // units per workgroup
constexpr int D_LD = 192;        // dY tile row stride (bytes): 128 B data + 64 B pad -> tr reads conflict-free
constexpr int X_LD = 64;         // X tile row stride (bytes): 32 bf16, rows land 16 banks apart

struct WgradArgsB {
  const bf16_t* x;   // [B][H][W][Cin]
  const bf16_t* dy;  // [B][OH][OW][Cout]
  float* dw;         // [Cout][KH][KW][Cin] f32
  int B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW;
  int units;
  int64_t npix, chunk;
};

__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)(p));
}

template <int PF>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(WgradArgsB a) {
  __shared__ __attribute__((aligned(16))) unsigned char Ds[WKP * D_LD];
  __shared__ __attribute__((aligned(16))) unsigned char Xs[WUN * WKP * X_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * WUN;
  const int co0 = blockIdx.y * WCO;
  const int64_t p_begin = (int64_t)blockIdx.z * a.chunk;
  int64_t p_end = p_begin + a.chunk;
  if (p_end > a.npix) p_end = a.npix;
  const int cchunks = a.Cin / 32;
  const int ohw = a.OH * a.OW;

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
  // staging maps: dY tile 64 px x 128 B = 512 16-B pieces (2 per thread); X tile per unit 64 px x 64 B
  // = 256 pieces (1 per thread and unit)
  const int dpx = tid >> 3, dseg = tid & 7;   // + 32 rows on the second pass
  const int xpx = tid >> 2, xseg = tid & 3;

  f32x16 acc[2][WUW];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < WUW; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

  u32x4 rd[PF][2], rx[PF][WUN];
  auto gload = [&](int64_t p0, u32x4 (&rd)[2], u32x4 (&rx)[WUN]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t p = p0 + dpx + 32 * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      int co = co0 + dseg * 8;
      if (p < p_end && co < a.Cout) v = *reinterpret_cast<const u32x4*>(a.dy + (size_t)p * a.Cout + co);
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
      u32x4 v = {0u, 0u, 0u, 0u};
      if (pok && uok[u] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
        v = *reinterpret_cast<const u32x4*>(a.x + ((size_t)(b * a.H + iy) * a.W + ix) * a.Cin + uci[u] + xseg * 8);
      rx[u] = v;
    }
  };
  auto lstore = [&](const u32x4 (&rd)[2], const u32x4 (&rx)[WUN]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(&Ds[(dpx + 32 * i) * D_LD + dseg * 16]) = rd[i];
#pragma unroll
    for (int u = 0; u < WUN; ++u) *reinterpret_cast<u32x4*>(&Xs[(u * WKP + xpx) * X_LD + xseg * 16]) = rx[u];
  };

  // transposing-read lane map (every lane participates: EXEC must be full)
  const int li = lane & 15, q = li >> 2, p4 = li & 3;
  const int half16 = (lane >> 4) & 1;  // which 16 of the 32 MFMA rows/cols this 16-lane group owns
  const int h = lane >> 5;             // k half: pixels 8h .. 8h+7 of a 16-pixel MFMA step
  const unsigned char* a_base0 = &Ds[(8 * h + q) * D_LD + (half16 * 16 + p4 * 4) * 2];        // co 0..31
  const unsigned char* a_base1 = a_base0 + 64;                                                 // co 32..63
  const unsigned char* b_base = &Xs[(wave * WUW * WKP + 8 * h + q) * X_LD + (half16 * 16 + p4 * 4) * 2];

  if (p_begin < p_end) {
    gload(p_begin, rd[0], rx[0]);
    lstore(rd[0], rx[0]);
  }
  __syncthreads();
#pragma unroll
  for (int j = 1; j < PF; ++j)
    if (p_begin + (int64_t)j * WKP < p_end) gload(p_begin + (int64_t)j * WKP, rd[j], rx[j]);
  for (int64_t pb = p_begin; pb < p_end; pb += (int64_t)PF * WKP) {
#pragma unroll
   for (int j = 0; j < PF; ++j) {
    const int64_t p0 = pb + (int64_t)j * WKP;
    if (p0 >= p_end) break;
    bool more = p0 + WKP < p_end;
    if (p0 + (int64_t)PF * WKP < p_end) gload(p0 + (int64_t)PF * WKP, rd[j], rx[j]);
#pragma unroll
    for (int ks = 0; ks < WKP / 16; ++ks) {
      bf16x4 a0l = tr_read(a_base0 + (16 * ks) * D_LD), a0h = tr_read(a_base0 + (16 * ks + 4) * D_LD);
      bf16x4 a1l = tr_read(a_base1 + (16 * ks) * D_LD), a1h = tr_read(a_base1 + (16 * ks + 4) * D_LD);
      bf16x8 a0 = __builtin_shufflevector(a0l, a0h, 0, 1, 2, 3, 4, 5, 6, 7);
      bf16x8 a1 = __builtin_shufflevector(a1l, a1h, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int u = 0; u < WUW; ++u) {
        const unsigned char* bp = b_base + u * WKP * X_LD;
        bf16x4 bl = tr_read(bp + (16 * ks) * X_LD), bh = tr_read(bp + (16 * ks + 4) * X_LD);
        bf16x8 bb = __builtin_shufflevector(bl, bh, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bb, acc[0][u], 0, 0, 0);
        acc[1][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bb, acc[1][u], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      lstore(rd[(j + 1) % PF], rx[(j + 1) % PF]);
      __syncthreads();
    }
   }
  }
  const int r = lane & 31;
  const int taps = a.KH * a.KW;
#pragma unroll
  for (int u = 0; u < WUW; ++u) {
    const int uu = u0 + wave * WUW + u;
    if (uu >= a.units) continue;
    const int tap = uu / cchunks;
    const int ci = (uu - tap * cchunks) * 32 + r;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int co = co0 + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (co < a.Cout) atomicAdd(a.dw + ((size_t)co * taps + tap) * a.Cin + ci, acc[t][u][g]);
      }
    }
  }
}

int check_conv(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return WSMG_EINVAL;
  if (Cin % 32 || Cout % 32) return WSMG_EINVAL;
  if (stride != 1 && stride != 2) return WSMG_EINVAL;
  if (OH != (H + 2 * pad - KH) / stride + 1 || OW != (W + 2 * pad - KW) / stride + 1) return WSMG_EINVAL;
  if ((int64_t)B * H * W >= (1ll << 31) || (int64_t)B * OH * OW >= (1ll << 31)) return WSMG_EINVAL;
  return 0;
}

int conv_tap_inner() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("WSMG_CONV_TAP_INNER");
    v = e ? (atoi(e) != 0) : 1;
  }
  return v;
}

// register prefetch depth (k-steps in flight).  Measured on MI355X (tools/bench_conv.py): the igemm
// kernels are fastest at 1 (deeper costs a wave of occupancy), backward-weight at 2.
int conv_prefetch(int dflt) {
  static int v = -2;
  if (v == -2) {
    const char* e = getenv("WSMG_CONV_PF");
    v = e ? atoi(e) : -1;
    if (v < 1 || v > 3) v = -1;
  }
  return v > 0 ? v : dflt;
}

template <bool BWD, int PF>
void launch_igemm_pf(ConvArgsB& a, dim3 grid, bool bk64, bool bn128, hipStream_t s) {
  if (bk64 && bn128)
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BWD, 128, 64, PF>), grid, dim3(256), 0, s, a);
  else if (bk64)
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BWD, 64, 64, PF>), grid, dim3(256), 0, s, a);
  else if (bn128)
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BWD, 128, 32, PF>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BWD, 64, 32, PF>), grid, dim3(256), 0, s, a);
}

template <bool BWD>
void launch_igemm(ConvArgsB& a, int64_t mrows, int classes, hipStream_t s) {
  a.tap_inner = conv_tap_inner();
  const bool bk64 = (a.Kc % 64) == 0;
  const bool bn128 = a.N >= 128;
  const int bn = bn128 ? 128 : 64;
  a.mtiles = (int)wsmg_cdiv(mrows, BM);
  a.ntiles = (int)wsmg_cdiv(a.N, bn);
  dim3 grid((unsigned)(a.mtiles * a.ntiles), (unsigned)classes);
  switch (conv_prefetch(1)) {
    case 1: launch_igemm_pf<BWD, 1>(a, grid, bk64, bn128, s); break;
    case 3: launch_igemm_pf<BWD, 3>(a, grid, bk64, bn128, s); break;
    case 2: launch_igemm_pf<BWD, 2>(a, grid, bk64, bn128, s); break;
    default: launch_igemm_pf<BWD, 1>(a, grid, bk64, bn128, s); break;
  }
}

}  // namespace

extern "C" int wsmg_conv2d_fwd_bf16(const void* x, const void* w_ohwi, const float* bias, void* y, int out_f32, int B,
                                    int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                                    int OW, wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  ConvArgsB a{(const bf16_t*)x, (const bf16_t*)w_ohwi, bias, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, 0, 0, out_f32, 0};
  launch_igemm<false>(a, (int64_t)B * OH * OW, 1, wsmg_s(stream));
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_conv2d_bwd_data_bf16(const void* dy, const void* w_ihwo, void* dx, int out_f32, int B, int H, int W,
                                         int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                         wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  ConvArgsB a{(const bf16_t*)dy, (const bf16_t*)w_ihwo, nullptr, dx, B, OH, OW, Cout, H, W, Cin, KH, KW, stride, pad, 0, 0, out_f32, 0};
  int classes = 1;
  int64_t mmax = (int64_t)B * H * W;
  if (stride == 2) {
    classes = 4;
    mmax = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  }
  launch_igemm<true>(a, mmax, classes, wsmg_s(stream));
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_conv2d_bwd_weight_bf16(const void* x, const void* dy, float* dw_ohwi, int B, int H, int W, int Cin,
                                           int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                           wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  WgradArgsB a{(const bf16_t*)x, (const bf16_t*)dy, dw_ohwi, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, 0, 0, 0};
  a.units = KH * KW * (Cin / 32);
  a.npix = (int64_t)B * OH * OW;
  int gx = (int)wsmg_cdiv(a.units, WUN), gy = (int)wsmg_cdiv(Cout, WCO);
  int64_t want = wsmg_cdiv(2048, (int64_t)gx * gy);
  int64_t maxz = wsmg_cdiv(a.npix, WKP * 8);
  int64_t gz = want < 1 ? 1 : (want > maxz ? maxz : want);
  if (gz < 1) gz = 1;
  if (gz > 65535) gz = 65535;
  a.chunk = wsmg_cdiv(wsmg_cdiv(a.npix, gz), WKP) * WKP;
  gz = wsmg_cdiv(a.npix, a.chunk);
  switch (conv_prefetch(2)) {
    case 1: hipLaunchKernelGGL(conv_wgrad_bf16_kernel<1>, dim3(gx, gy, (unsigned)gz), dim3(256), 0, wsmg_s(stream), a); break;
    case 3: hipLaunchKernelGGL(conv_wgrad_bf16_kernel<3>, dim3(gx, gy, (unsigned)gz), dim3(256), 0, wsmg_s(stream), a); break;
    default: hipLaunchKernelGGL(conv_wgrad_bf16_kernel<2>, dim3(gx, gy, (unsigned)gz), dim3(256), 0, wsmg_s(stream), a); break;
  }
  WSMG_RETURN_LAUNCH();
}
