// ConvTranspose2d(64 -> 32, k4, s2, p1) forward = backward-data of the adjoint stride-2 convolution: the first layer of the semantic
// classifier (mg_map_policy.py:79 of the reference), 24 x 24 x 64 -> 48 x 48 x 32 at B = 512 — round 6.
//
// Why a kernel of its own.  An output pixel (2 i + a, 2 j + b) is a 2 x 2-tap sum over the input pixels around (i, j): four parity
// classes, each a stride-1 correlation with K = 4 taps x 64 channels.  On the implicit-GEMM kernel (wsmg_conv_bf16.hip, four classes as
// blockIdx.y) a workgroup's reduction is 4 k-steps behind a prologue and an epilogue that cost as much, the 32 output channels fill
// half of its 64-wide tile, and a class writes every other pixel — half of each 128-byte line: 76 us alone (254 TFLOP/s), 96 us
// in the update, where it sits alone on the critical path between the decoder and the classifier.  Here (the structure of
// wsmg_conv_win3_k32.hip):
//   * a workgroup (4 waves) owns one ROW parity (blockIdx & 1) and BOTH column parities: the weights of its two classes (2 x 4 taps x
//     32 x 64, 32 KB) stay in LDS for its life, and an input pixel's two output pixels leave as one full 128-byte line;
//   * per 128-pixel tile of the input grid: the zero-padded window (256 entries of 128 B, LDS-DMA, the window kernels' geometry) ->
//     ONE barrier -> 2 classes x 4 taps x 4 slices of MFMAs, no barrier between them -> the output tile through LDS -> 16-byte stores;
//   * 64 KB of LDS: two workgroups per CU.
// Same MFMA sequence per output element as the implicit-GEMM kernel (tap row, tap column, 16-deep slice; one accumulator):
// bit-identical outputs.  BatchNorm sums per lane over a workgroup's run, one set of float64 atomics per workgroup.
#include "wsmg_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct CtArgs {
  const bf16_t* src;  // [B][H][W][64]      the ConvTranspose's input (the adjoint convolution's dy)
  const bf16_t* wt;   // [32][4][4][64]     IHWO of the adjoint convolution
  bf16_t* dst;        // [B][2H][2W][32]
  int B, H, W, mtiles;
  unsigned src_bytes, wt_bytes;
  double* stats;      // [nslab][2][32] or null
  int nslab;
};

constexpr int CT_MT = 128, CT_ENT = 256, CT_ROWB = 128, CT_WINB = CT_ENT * CT_ROWB, CT_TAPB = 32 * CT_ROWB, CT_WTB = 8 * CT_TAPB;
constexpr int CT_OP = 144;                 // staged output: [input pixel][column parity][32 channels] + 16
constexpr int CT_LDS = CT_WINB + CT_WTB + CT_MT * 4;
static_assert(CT_MT * CT_OP <= CT_WINB, "the output tile is staged in the window");

__device__ __forceinline__ void dma16c(__amdgpu_buffer_rsrc_t r, unsigned char* lds_wave_base, int byte_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
__device__ __forceinline__ unsigned short f2bfc(float f) {
  bf16_t b = (bf16_t)f;
  return __builtin_bit_cast(unsigned short, b);
}

__global__ __launch_bounds__(256) void convt_k4s2_kernel(CtArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const win = smem;
  unsigned char* const wts = smem + CT_WINB;
  int* const opix = reinterpret_cast<int*>(smem + CT_WINB + CT_WTB);   // [128] output pixel (even column) of the tile's input pixels, or -1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, PW = W + 2, PP = (H + 2) * PW;
  const int Mtot = a.B * H * W;
  const int cy = (int)blockIdx.x & 1;          // parity of the kernel row: taps ky = cy, cy + 2; output rows of parity (cy + 1) & 1
  const int ty0 = (cy + 1) & 1;
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.src), 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wt), 0, (int)a.wt_bytes, 0x00020000);
  const int drow = lane >> 3, dslot = lane & 7;   // DMA roles: a 1 KB piece = 8 rows of 128 B, lane = (row, 16-byte slot)

  // ---- weights of this row parity, once: block (cx, ay, ax) = 32 rows n of [64 k] at wts[(cx * 4 + ay * 2 + ax)]; 16-byte chunk c of row n
  // at slot c ^ ((n >> 1) & 7).  32 pieces of 8 rows, 8 per wave.
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int p = 4 * j + wave;                 // piece: block p >> 2, rows 8 (p & 3) ..
    const int blk = p >> 2, n = 8 * (p & 3) + drow;
    const int cx = blk >> 2, ay = (blk >> 1) & 1, ax = blk & 1;
    const int ky = cy + 2 * ay, kx = cx + 2 * ax;
    dma16c(rs_wt, wts + p * 1024, ((n * 4 + ky) * 4 + kx) * 128 + 16 * (dslot ^ ((n >> 1) & 7)));
  }
  int bpos[4];   // byte offset of this lane's weight fragment of slice ks inside a block
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bpos[ks] = r * CT_ROWB + (((2 * ks + h) ^ ((r >> 1) & 7)) << 4);
  double st_s = 0.0, st_q = 0.0;

  auto padded = [&](int m) {
    const int b = m / (H * W), rr = m - b * (H * W), y = rr / W, x = rr - y * W;
    return b * PP + (y + 1) * PW + x + 1;
  };

  const int units = a.mtiles * 2;
  for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
    const int m0 = (unit >> 1) * CT_MT;
    const int mlast = (m0 + CT_MT - 1 < Mtot ? m0 + CT_MT - 1 : Mtot - 1);
    const int q0 = padded(m0) - PW - 1;
    const int nwin = padded(mlast) + PW + 1 - q0 + 1;
    // ---- window: piece j of this wave = entries 8 (4 j + wave) .. + 7
    {
      const int e0 = 8 * wave + drow;
      const int q = q0 + e0;
      int b = q / PP;
      const int rem = q - b * PP;
      int pr = rem / PW, pc = rem - pr * PW;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + 32 * j;
        const int yy = pr - 1, xx = pc - 1;
        const bool ok = e < nwin && b < a.B && yy >= 0 && yy < H && xx >= 0 && xx < W;
        dma16c(rs_src, win + (4 * j + wave) * 1024, ok ? (((b * H + yy) * W + xx) * 64) * 2 + 16 * (dslot ^ ((e >> 1) & 7)) : (int)0x80000000);
        pc += 32;
        while (pc >= PW) { pc -= PW; ++pr; }
        while (pr >= H + 2) { pr -= H + 2; ++b; }
      }
    }
    // this lane's input pixel (accumulator row r of the wave's 32-pixel tile) and, for the store phase, every pixel's output address
    int ecen;
    {
      const int m = m0 + wave * 32 + r;
      const int mm = m < Mtot ? m : Mtot - 1;
      const int b = mm / (H * W), rr = mm - b * (H * W), y = rr / W, x = rr - y * W;
      ecen = b * PP + (y + 1) * PW + x + 1 - q0;
      if (h == 0) opix[wave * 32 + r] = m < Mtot ? ((b * 2 * H + 2 * y + ty0) * 2 * W + 2 * x) : -1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
#pragma unroll
    for (int ay = 0; ay < 2; ++ay)
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        // source pixel of tap (ay, ax) for column parity cx: row i + (1 - cy) - ay, column j + (1 - cx) - ax
        const int erow = ecen + ((1 - cy) - ay) * PW;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
          for (int cx = 0; cx < 2; ++cx) {
            const int e = erow + (1 - cx) - ax;
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(win + e * CT_ROWB + (((2 * ks + h) ^ ((e >> 1) & 7)) << 4));
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wts + (cx * 4 + ay * 2 + ax) * CT_TAPB + bpos[ks]);
            acc[cx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[cx], 0, 0, 0);
          }
        }
      }
    __syncthreads();   // every wave has read its last fragment: the window becomes the output staging

    // ---- epilogue: [128 input pixels][column parity][32 channels] at a 144-byte pitch: an input pixel's two output pixels are one
    // 128-byte line (even column = kernel-column parity 1, odd column = parity 0)
    float sv = 0.f, qv = 0.f;
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = wave * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        const unsigned short o = f2bfc(acc[cx][g]);
        *reinterpret_cast<unsigned short*>(win + row * CT_OP + (1 - cx) * 64 + r * 2) = o;
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
      const int row = c >> 3, ch = c & 7;
      const int op = opix[row];
      if (op >= 0)
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.dst) + ((size_t)op * 32) * 2 + ch * 16) =
            *reinterpret_cast<const u32x4*>(win + row * CT_OP + ch * 16);
    }
    __syncthreads();   // the staged tile (and opix) have been read before the next window lands
  }

  if (a.stats) {   // one flush per workgroup: lanes r and r + 32 hold different rows of channel r; then the four waves, in order
    double* const sh = reinterpret_cast<double*>(win);
    const double s2 = st_s + __shfl_xor(st_s, 32, 64), q2 = st_q + __shfl_xor(st_q, 32, 64);
    if (h == 0) {
      sh[wave * 64 + r] = s2;
      sh[wave * 64 + 32 + r] = q2;
    }
    __syncthreads();
    if (tid < 64) {
      const double v = (sh[tid] + sh[64 + tid]) + (sh[128 + tid] + sh[192 + tid]);
      atomicAdd(a.stats + (size_t)(((int)blockIdx.x >> 1) % a.nslab) * 64 + (tid & 31) + (tid >> 5) * 32, v);
    }
  }
}

int ct_window_bound(int mt, int H, int W) {
  const int rows = (mt + W - 2) / W + 1;
  const int imgs = (mt + H * W - 2) / (H * W) + 1;
  return mt + 2 * (rows - 1) + 2 * (W + 2) * (imgs - 1) + 2 * (W + 3) + 1;
}

}  // namespace

// dst [B][2H][2W][32] = ConvTranspose2d(k4, s2, p1) of src [B][H][W][64] with the adjoint convolution's IHWO weights [32][4][4][64];
// WSMG_EINVAL for geometries whose window does not fit (the caller then uses the implicit-GEMM kernel)
int wsmg_convt_k4s2_bf16(const void* src, const void* w_ihwo, void* dst, double* stats, int nslab, int B, int H, int W, hipStream_t s) {
  if (B <= 0 || H <= 0 || W <= 0 || ct_window_bound(CT_MT, H, W) > CT_ENT || (stats && nslab <= 0)) return WSMG_EINVAL;
  if ((int64_t)B * 4 * H * W * 32 * 2 >= (1ll << 31) || (int64_t)B * (H + 2) * (W + 2) >= (1 << 30)) return WSMG_EINVAL;
  CtArgs a{(const bf16_t*)src, (const bf16_t*)w_ihwo, (bf16_t*)dst, B, H, W, (int)wsmg_cdiv((int64_t)B * H * W, CT_MT),
           (unsigned)((size_t)B * H * W * 64 * 2), (unsigned)(32u * 16 * 64 * 2), stats, nslab};
  static int cus = 0;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(convt_k4s2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CT_LDS);
    if (e != hipSuccess) return (int)e;
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    attr = true;
  }
  const int64_t units = (int64_t)a.mtiles * 2;
  int64_t grid = (int64_t)WSMG_TUNE("WSMG_CONVT_WGS", 6) * cus;
  if (grid > units) grid = units;
  grid &= ~(int64_t)1;                          // a workgroup's row parity is blockIdx & 1 for every unit it walks
  if (grid <= 0) return WSMG_EINVAL;
  hipLaunchKernelGGL(convt_k4s2_kernel, dim3((unsigned)grid), dim3(256), CT_LDS, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
