// 3 x 3 / stride 1 / pad 1 convolution (forward and backward-data) over a 32-CHANNEL reduction axis: the semantic classifier's
// 32 -> 32 layer at 48 x 48 both ways and its 32 -> 128 projection (mg_map_policy.py:78-86 of the reference) — round 6.
//
// Why a separate structure.  With Kc = 32 the window kernel (wsmg_conv_win3.hip) has ONE channel chunk: nine k-steps of 2 MFMAs per
// wave, a workgroup barrier and a 4 KB weight stage each, under a window load nothing precedes — 55 us for a layer that moves
// 151 MB (28 us at the rate the BatchNorm passes stream at) and holds 18 MFMAs per 32 pixels.  The reduction is 288 deep: all of a
// 32-output-channel tile's weights are 18 KB.  So here
//   * a workgroup (4 waves) keeps its channel tile's weights in LDS for its whole life and walks the pixel tiles
//     blockIdx, blockIdx + gridDim, ... (gridDim = 6 per CU — two or three tiles per workgroup at B = 512: three are resident per CU, and a
//     grid of exactly the resident workgroups queues badly behind whatever else shares the chip — and a multiple of the channel tiles, so
//     a workgroup's channel tile is fixed);
//   * per 256-pixel tile: the zero-padded window (wsmg_conv_win3.hip's geometry and swizzle, LDS-DMA) -> ONE barrier -> 9 taps x 2
//     slices of MFMAs with no barrier between them (wave tile 64 pixels x 32 channels) -> the output tile through LDS -> 16-byte stores;
//   * 50 KB of LDS: three workgroups per CU, whose load / compute / store phases overlap each other — the overlap a single
//     chunk cannot give a workgroup with itself.
// Same MFMA sequence per output element as the window kernel (tap-major, slice-minor, one accumulator): bit-identical outputs.
// BatchNorm sums: per lane over a workgroup's whole run, one set of float64 atomics per workgroup at the end.
#include "wsmg_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct K32Args {
  const bf16_t* src;  // [B][H][W][32]
  const bf16_t* wt;   // [N][3][3][32]
  const float* bias;  // [N] or null
  bf16_t* dst;        // [B][H][W][dst_ld] (this layer's N channels from channel 0 of the pointer)
  int B, H, W, N, ntiles, mtiles, relu, bwd, dst_ld;
  unsigned src_bytes, wt_bytes;
  double* stats;
  int nslab;
};

constexpr int MT = 256, WCAP = 512, ROWB = 64, WINB = WCAP * ROWB, TAPB = 32 * ROWB, WTB = 9 * TAPB, OP = 80;
constexpr int K32_LDS = WINB + WTB;

__device__ __forceinline__ void dma16k(__amdgpu_buffer_rsrc_t r, unsigned char* lds_wave_base, int byte_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
__device__ __forceinline__ unsigned short f2bfk(float f) {
  bf16_t b = (bf16_t)f;
  return __builtin_bit_cast(unsigned short, b);
}

__global__ __launch_bounds__(256) void conv_win3_k32_kernel(K32Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const win = smem;
  unsigned char* const wts = smem + WINB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane >> 2, slot = lane & 3;
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, PW = W + 2, PP = (H + 2) * PW;
  const int Mtot = a.B * H * W;
  const int n0 = ((int)blockIdx.x % a.ntiles) * 32;
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.src), 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wt), 0, (int)a.wt_bytes, 0x00020000);

  // ---- this channel tile's weights, once: piece p = (tap, 16-row half) -> wts[tap][row][64 B], 16-byte slot s of a row at s ^ ((row >> 2) & 3)
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int p = wave + 4 * j;
    const int tap = p >> 1, row = 16 * (p & 1) + lrow;
    if (p < 18) dma16k(rs_wt, wts + tap * TAPB + (p & 1) * 1024, ((n0 + row) * 9 + tap) * 64 + 16 * (slot ^ ((row >> 2) & 3)));
  }
  const int bposk[2] = {r * ROWB + (((0 + h) ^ ((r >> 2) & 3)) << 4), r * ROWB + (((2 + h) ^ ((r >> 2) & 3)) << 4)};
  const float bv = a.bias ? a.bias[n0 + r] : 0.f;
  const int sgn = a.bwd ? -1 : 1;
  const int ldd = a.dst_ld ? a.dst_ld : a.N;
  double st_s = 0.0, st_q = 0.0;

  auto padded = [&](int m) {
    const int b = m / (H * W), rr = m - b * (H * W), y = rr / W, x = rr - y * W;
    return b * PP + (y + 1) * PW + x + 1;
  };

  const int ntl = a.mtiles * a.ntiles;
  for (int tile = blockIdx.x; tile < ntl; tile += gridDim.x) {
    const int m0 = (tile / a.ntiles) * MT;
    const int mlast = (m0 + MT - 1 < Mtot ? m0 + MT - 1 : Mtot - 1);
    const int q0 = padded(m0) - PW - 1;
    const int nwin = padded(mlast) + PW + 1 - q0 + 1;
    // ---- window: piece j of this wave = entries 16 (4 j + wave) .. + 15.  (image, padded row, padded column) of the first piece's entry
    // by division, of the others by stepping 64 entries on: the divisions of all 8 pieces were most of a tile's VALU time
    {
      const int e0 = 16 * wave + lrow;
      const int q = q0 + e0;                                   // (q0 >= 0: the window starts on the pad row above the first pixel)
      int b = q / PP;
      const int rem = q - b * PP;
      int pr = rem / PW, pc = rem - pr * PW;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + 64 * j;
        const int yy = pr - 1, xx = pc - 1;
        const bool ok = e < nwin && b < a.B && yy >= 0 && yy < H && xx >= 0 && xx < W;
        dma16k(rs_src, win + (4 * j + wave) * 1024, ok ? (((b * H + yy) * W + xx) * 32) * 2 + 16 * (slot ^ ((e >> 2) & 3)) : (int)0x80000000);
        pc += 64;
        while (pc >= PW) { pc -= PW; ++pr; }
        while (pr >= H + 2) { pr -= H + 2; ++b; }
      }
    }
    int ecen[2];
    {
      const int m = m0 + wave * 64 + r;                        // second accumulator tile: 32 pixels on
      int b = m / (H * W);
      const int rr = m - b * (H * W);
      int y = rr / W, x = rr - y * W;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // (pixels past the end repeat the last pixel, as padded(min(m, Mtot - 1)) did: their rows are never stored)
        ecen[t] = m + 32 * t < Mtot ? b * PP + (y + 1) * PW + x + 1 - q0 : padded(Mtot - 1) - q0;
        x += 32;
        while (x >= W) { x -= W; ++y; }
        while (y >= H) { y -= H; ++b; }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int shift = sgn * ((tap / 3 - 1) * PW + (tap % 3 - 1));
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wts + tap * TAPB + bposk[ks]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int e = ecen[t] + shift;
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(win + e * ROWB + (((2 * ks + h) ^ ((e >> 2) & 3)) << 4));
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();   // every wave has read its last fragment: the window becomes the output staging

    // ---- epilogue: [256 rows][32 channels] at an 80-byte pitch, then 16-byte stores (4 per thread)
    float sv = 0.f, qv = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = wave * 64 + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
        float v = acc[t][g] + bv;
        if (a.relu) v = v > 0.f ? v : 0.f;
        const unsigned short o = f2bfk(v);
        *reinterpret_cast<unsigned short*>(win + row * OP + r * 2) = o;
        if (a.stats) {   // train-mode BatchNorm sums of the ROUNDED values
          const float vr = m0 + row < Mtot ? __uint_as_float((unsigned)o << 16) : 0.f;
          sv += vr;
          qv = fmaf(vr, vr, qv);
        }
      }
    st_s += (double)sv;
    st_q += (double)qv;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j;
      const int row = c >> 2, ch = c & 3;
      if (m0 + row < Mtot)
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.dst) + ((size_t)(m0 + row) * ldd + n0 + ch * 8) * 2) =
            *reinterpret_cast<const u32x4*>(win + row * OP + ch * 16);
    }
    __syncthreads();   // the staged tile has left LDS before the next window lands on it
  }

  if (a.stats) {   // one flush per workgroup: lanes r and r + 32 hold different rows of channel n0 + r; then the four waves, in order
    double* const sh = reinterpret_cast<double*>(win);
    const double s2 = st_s + __shfl_xor(st_s, 32, 64), q2 = st_q + __shfl_xor(st_q, 32, 64);
    if (h == 0) {
      sh[wave * 64 + r] = s2;
      sh[wave * 64 + 32 + r] = q2;
    }
    __syncthreads();
    if (tid < 64) {
      const double v = (sh[tid] + sh[64 + tid]) + (sh[128 + tid] + sh[192 + tid]);
      double* st = a.stats + (size_t)(((int)blockIdx.x / a.ntiles) % a.nslab) * 2 * a.N + n0 + (tid & 31) + (tid >> 5) * a.N;
      atomicAdd(st, v);
    }
  }
}

// window entries a tile of mt consecutive pixels can need (wsmg_conv_win3.hip's window_bound)
int k32_window_bound(int mt, int H, int W) {
  const int rows = (mt + W - 2) / W + 1;
  const int imgs = (mt + H * W - 2) / (H * W) + 1;
  return mt + 2 * (rows - 1) + 2 * (W + 2) * (imgs - 1) + 2 * (W + 3) + 1;
}

}  // namespace

// Kc == 32, N % 32 == 0, plain output (no mask, no split); WSMG_EINVAL otherwise (the caller then uses the general window kernel)
int wsmg_conv_win3_k32_bf16(int bwd, const void* src, const void* wt, const float* bias, void* dst, int relu, double* stats, int nslab,
                            int B, int H, int W, int N, int dst_ld, hipStream_t s) {
  if (N <= 0 || N % 32 || B <= 0 || k32_window_bound(MT, H, W) > WCAP) return WSMG_EINVAL;
  if (stats && nslab <= 0) return WSMG_EINVAL;
  K32Args a{(const bf16_t*)src, (const bf16_t*)wt, bias, (bf16_t*)dst, B, H, W, N, N / 32, 0, relu, bwd, dst_ld,
            (unsigned)((size_t)B * H * W * 32 * 2), (unsigned)((size_t)N * 9 * 32 * 2), stats, nslab};
  const int64_t M = (int64_t)B * H * W;
  a.mtiles = (int)wsmg_cdiv(M, MT);
  static int cus = 0;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_win3_k32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, K32_LDS);
    if (e != hipSuccess) return (int)e;
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    attr = true;
  }
  const int64_t tiles = (int64_t)a.mtiles * a.ntiles;
  int64_t grid = (int64_t)WSMG_TUNE("WSMG_CONV_K32_WGS", 6) * cus;
  if (grid > tiles) grid = tiles;
  grid = grid / a.ntiles * a.ntiles;            // a workgroup's channel tile is blockIdx % ntiles for every tile it walks
  if (grid <= 0) return WSMG_EINVAL;
  hipLaunchKernelGGL(conv_win3_k32_kernel, dim3((unsigned)grid), dim3(256), K32_LDS, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
