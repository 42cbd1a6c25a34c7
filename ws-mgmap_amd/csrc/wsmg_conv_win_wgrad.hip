// Weight gradient of the map encoder's stem (64 -> 64 channels, k8 s2 p3; map_encoder.py:29-31, `MapEncoder.cnn[0]`:
// 671 GFLOP at B = 512, the largest single kernel of the update) out of an LDS-RESIDENT INPUT WINDOW.
//
//   dW[co][ky][kx][ci] = sum over output pixels p = (b, oy, ox) of dY[p][co] * X[b][2 oy - 3 + ky][2 ox - 3 + kx][ci]
//
// Why: in the generic kernel (wsmg_conv_bf16.hip, conv_wgrad_bf16_kernel) every (tap, 32-channel) unit re-fetches its
// own shifted copy of X through the 64 B/clk/CU vector-memory path — 64 taps x 1.28 M pixels x 128 B = 10.5 GB through L1
// for a 655 MB tensor — and that path, not the MFMA, sets the pace (733 TFLOP/s, 28 % MFMA-busy).  Every input pixel of
// this layer is used by 16 taps.  Here a workgroup takes ONE kernel row ky and a 5 x 25 block of output pixels: it loads
// the 5 input rows that ky touches (5 x 56 pixels x 64 channels = 35 KB, once) and the block's dY tile (16 KB), and all
// 8 taps kx of the row read their X^T fragments out of that window — consecutive output pixels of a tap are consecutive
// entries of one column-parity plane, as in the forward window kernel (wsmg_conv_win.hip).  Memory -> LDS traffic per
// MFMA drops 3.6x and the 64 co x (8 kx x 64 ci) = 32 768 float32 accumulators of the role stay in registers over ALL
// the workgroup's tiles (persistent: 8 roles x G groups of tiles), so each workgroup ends with one accumulator flush.
//
// Reduction axis = pixels, which is the strided axis of NHWC for both operands: fragments come from the gfx950
// transposing LDS read (ds_read_b64_tr_b16), exactly as in the generic kernel.  A tile's 125 pixels are 8 K-chunks of 16
// (3 padding pixels: their dY rows are zero, their X addresses repeat the last pixel).
#include <stdlib.h>

#include "wsmg_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WinWgradArgs {
  const bf16_t* x;    // [B][H][W][64]
  const bf16_t* dy;   // [B][OH][OW][64]
  float* dw;          // [64][8][8][64] float32 (OHWI), accumulated into (slab == 0) — or the slab workspace [groups][64][8][8][64]
  int64_t slab;       // > 0: floats per slab — tile group g STORES its partial sums into slab g (its 8 roles cover the 8 kernel rows):
                      // no atomics, no zero fill, summed in slab order by wsmg_weight_grad_reduce_oihw (bit-reproducible)
  int B, H, W, OH, OW, tiles_y, tiles_x, groups;
  unsigned x_bytes, dy_bytes;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)(p));
}

constexpr int K = 8, S = 2, P = 3, TH = 5, TW = 25;
constexpr int WC = (TW - 1) * S + K;   // 56 window columns
constexpr int PC = WC / S;             // 28 entries per column-parity plane
constexpr int PITCH = 192;             // bytes per pixel entry (128 + 64): a 32-lane half of a transposing read takes 4 consecutive
                                       // entries x 64 B, which must cover the 256-B bank row exactly once — pitch = 64 or 192 mod 256
                                       // (at 144, the forward kernel's pitch, entries 0 and 2 collide: 40 % of the LDS cycles were conflicts)
constexpr int XROW = S * PC * PITCH;   // one window row: 2 parity planes
constexpr int X_BYTES = TH * XROW;     // 53 760
constexpr int NPX = 128;               // K axis of a tile: 125 pixels + 3 padding
constexpr int D_BYTES = NPX * PITCH;   // 24 576
constexpr int DPER = NPX * 8 / 256;    // 4

__global__ __launch_bounds__(256, 2) void conv_win_wgrad_kernel(WinWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // X_BYTES + D_BYTES = 78 336: two workgroups per CU
  unsigned char* const xs = lds;
  unsigned char* const ds = lds + X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Workgroup -> (kernel row ky, tile group): consecutive block ids go round the 8 XCDs, and the 8 roles of a group read
  // the same dY tiles and overlapping input rows (every input row is wanted by 2-3 of them) at about the same time — so
  // the roles of a group share an XCD (its L2): local index i = bid / 8 on XCD bid % 8 -> role i % 8, group (bid % 8) + 8 (i / 8).
  // With the roles spread over the XCDs (role = bid % 8) the kernel moved 4.2 GB through the fabric in 0.78 ms and waited on it.
  const int xcd = blockIdx.x & 7, li_ = blockIdx.x >> 3;
  const int ky = li_ & 7, grp = xcd + 8 * (li_ >> 3);
  const int tpi = a.tiles_y * a.tiles_x;
  const int ntiles = a.B * tpi;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes), dr = make_rsrc(a.dy, a.dy_bytes);

  // ---- staging (round 4: dealt by window ROW, as in wsmg_conv_s2_wgrad.hip).  A window row is 56 columns x 8 pieces = 448 pieces:
  // thread t owns piece (column t >> 3, chunk t & 7) and, for t < 192, the piece 32 columns further, of EVERY window row — the
  // columns' validity and byte offsets are computed once per tile, a row then costs one scalar base.  The dY pieces of a thread
  // (pixel (t >> 3) + 32 j of the 5 x 25 block) sit at per-thread constant offsets from the block's first pixel.  (Rounds 1-3 mapped
  // piece c = t + 256 j -> (row, column) by multiply-shift divisions per piece and tile, in gload AND in lstore: 26 pieces' worth of
  // address arithmetic beside a tile's 64 MFMAs.)
  constexpr int XPER2 = 2 * TH;
  const int wc0 = tid >> 3, pch = tid & 7;
  const bool has1 = tid < (WC - 32) * 8;
  const int xl0 = ((wc0 & 1) * PC + (wc0 >> 1)) * PITCH + pch * 16;            // LDS offsets within a window row (wc0 + 32 has wc0's parity)
  const int xl1 = xl0 + 16 * PITCH;
  int doff[DPER];       // dY piece j: byte offset from the block's first pixel, or -1 for the 3 padding pixels
#pragma unroll
  for (int j = 0; j < DPER; ++j) {
    const int q = wc0 + 32 * j;
    const int qr = (q * 41) >> 10;              // q / 25 for q < 128
    doff[j] = q < TH * TW ? ((qr * a.OW + q - qr * TW) * 64) * 2 + pch * 16 : -1;
  }
  u32x4 rx[XPER2], rd[DPER];
  auto gload = [&](int tile) {
    const int b = tile / tpi, t = tile - b * tpi;
    const int ty = t / a.tiles_x;
    const int oy0 = ty * TH, ox0 = (t - ty * a.tiles_x) * TW;
    const int ix0 = ox0 * S - P, iyb = oy0 * S - P + ky;
    const bool live = tile < ntiles;
    const int c0 = ix0 + wc0, c1 = c0 + 32;
    const bool ok0 = (unsigned)c0 < (unsigned)a.W, ok1 = has1 && (unsigned)c1 < (unsigned)a.W;
    const int cb0 = c0 * 128 + pch * 16, cb1 = cb0 + 32 * 128;
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int iy = iyb + j * S;                                               // (uniform)
      const bool rowok = live && (unsigned)iy < (unsigned)a.H;
      const int base = ((b * a.H + iy) * a.W) * 128;
      rx[2 * j] = buf_load16(xr, (rowok && ok0) ? base + cb0 : (int)0x80000000);
      rx[2 * j + 1] = buf_load16(xr, (rowok && ok1) ? base + cb1 : (int)0x80000000);
    }
    // (a block may hang over the right / bottom edge only where OW, OH are not multiples of 25 / 5: the map encoder's 50 x 50 is)
    const bool whole = live && oy0 + TH <= a.OH && ox0 + TW <= a.OW;
    const int dbase = (((b * a.OH + oy0) * a.OW + ox0) * 64) * 2;
#pragma unroll
    for (int j = 0; j < DPER; ++j) {
      bool ok = whole && doff[j] >= 0;
      if (!whole && live && doff[j] >= 0) {     // edge block: per-pixel test (never taken at 50 x 50)
        const int q = wc0 + 32 * j, qr = (q * 41) >> 10;
        ok = oy0 + qr < a.OH && ox0 + q - qr * TW < a.OW;
      }
      rd[j] = buf_load16(dr, ok ? dbase + doff[j] : (int)0x80000000);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      *reinterpret_cast<u32x4*>(xs + j * XROW + xl0) = rx[2 * j];
      if (has1) *reinterpret_cast<u32x4*>(xs + j * XROW + xl1) = rx[2 * j + 1];
    }
#pragma unroll
    for (int j = 0; j < DPER; ++j) *reinterpret_cast<u32x4*>(ds + (wc0 + 32 * j) * PITCH + pch * 16) = rd[j];
  };

  // ---- transposing-read lane map (as conv_wgrad_bf16_kernel): the lane supplies the address of pixel 8 h + q (+ 4 for
  // the second read) and channels 16 half16 + 4 p4 .. + 3; the hardware hands each lane 4 pixels of ITS channel
  const int li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const int half16 = (lane >> 4) & 1, h = lane >> 5;
  const int chan_off = (half16 * 16 + p4 * 4) * 2;
  // X entry of K index k = 16 c + 8 h + q4 (+ 4): pixel -> (row, ox); padding pixels repeat pixel 124.  Computed per chunk
  // (6 VALU instructions per address beside 8 MFMAs): a table of the 16 offsets cost 16 VGPRs and spilled.
  const int l0 = 8 * h + q4;
  auto xoff = [&](int c, int s2) {
    int p = 16 * c + l0 + 4 * s2;
    p = p < TH * TW ? p : TH * TW - 1;
    const int jr = (p * 41) >> 10;   // p / 25 for p < 128
    return jr * XROW + (p - jr * TW) * PITCH + chan_off;
  };
  const unsigned char* const a_base = ds + (8 * h + q4) * PITCH + chan_off;
  // this wave's taps kx = 2 wave, 2 wave + 1: parity plane kx & 1, entry shift kx >> 1
  const int kx0 = 2 * wave;
  const int tapoff0 = ((kx0 & 1) * PC + (kx0 >> 1)) * PITCH, tapoff1 = (((kx0 + 1) & 1) * PC + ((kx0 + 1) >> 1)) * PITCH;

  f32x16 acc[2][4];   // [co half][kx index * 2 + ci half]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[t][v][g] = 0.f;

  int tile = grp;
  if (tile >= ntiles) return;   // (whole workgroup: no tile, nothing to add)
  gload(tile);
  lstore();
  __syncthreads();
  // One wave per SIMD and workgroup, two workgroups per CU: left to itself hipcc reads two fragments, waits for them
  // (lgkmcnt(0)), issues two MFMAs, and so on — an LDS round trip per 64 MFMA clocks, 48 % MFMA-busy.  The fragments of
  // chunk c + 1 are therefore read into a second register set while the 8 MFMAs of chunk c run, interleaved explicitly.
  struct Frag { bf16x8 a[2], b[4]; };
  auto fload = [&](int c, Frag& f) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bf16x4 l = tr_read(a_base + 64 * t + (16 * c) * PITCH), hh = tr_read(a_base + 64 * t + (16 * c + 4) * PITCH);
      f.a[t] = __builtin_shufflevector(l, hh, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    const int x0 = xoff(c, 0), x1 = xoff(c, 1);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int to = (v >> 1) ? tapoff1 : tapoff0;
      const bf16x4 l = tr_read(xs + x0 + to + 64 * (v & 1)), hh = tr_read(xs + x1 + to + 64 * (v & 1));
      f.b[v] = __builtin_shufflevector(l, hh, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  for (; tile < ntiles; tile += a.groups) {
    gload(tile + a.groups);   // next tile -> registers while this one is multiplied (out of range: zeros)
    Frag cur, nxt;
    fload(0, cur);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < 8) fload(c + 1, nxt);
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[t], cur.b[v], acc[t][v], 0, 0, 0);
      if (c + 1 < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < 8) cur = nxt;
    }
    __syncthreads();
    lstore();
    __syncthreads();
  }

  // ---- one flush per workgroup: lane r = input channel (consecutive lanes -> 128-byte segments of an OHWI row)
  const int r = lane & 31;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int tap = ky * K + kx0 + (v >> 1);
    const int ci = 32 * (v & 1) + r;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int co = 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
        float* const q = a.dw + (size_t)grp * a.slab + ((size_t)co * (K * K) + tap) * 64 + ci;
        if (a.slab) *q = acc[t][v][g];
        else atomicAdd(q, acc[t][v][g]);
      }
  }
}

}  // namespace

// dW (OHWI float32) of a 64 -> 64 channel k8 s2 p3 convolution on bf16 NHWC — slab_floats == 0: accumulated into with float atomics
// (the caller zeroes it); > 0: one slab per tile group, stored (WinWgradArgs::slab); returns WSMG_EINVAL for any other shape (the
// caller then uses the generic kernel).
static bool winw_fits(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
  if (Cin != 64 || Cout != 64 || KH != 8 || KW != 8 || stride != 2 || pad != 3) return false;
  return !((size_t)B * H * W * 128 >= (1ull << 31) || (size_t)B * OH * OW * 128 >= (1ull << 31));
}
static int winw_groups(int B, int OH, int OW) {
  const int ntiles = B * ((OH + TH - 1) / TH) * ((OW + TW - 1) / TW);
  int groups = 64;   // 8 roles x 64 groups = 512 workgroups = 2 per CU (78 KB of LDS each)
  if (groups > ntiles) groups = ntiles;
  return (groups + 7) / 8 * 8;   // whole XCD rounds (groups beyond the tile count find no tile and only flush zeros)
}

// tile groups (= slabs of the deterministic form) this kernel would use; 0: the layer is not this kernel's
int wsmg_conv_win_wgrad_splits(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
  return winw_fits(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW) ? winw_groups(B, OH, OW) : 0;
}

int wsmg_conv_win_wgrad_bf16(const void* x, const void* dy, float* dw_ohwi, long long slab_floats, int B, int H, int W, int Cin, int Cout,
                             int KH, int KW, int stride, int pad, int OH, int OW, hipStream_t s) {
  if (!winw_fits(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return WSMG_EINVAL;
  WinWgradArgs a{(const bf16_t*)x, (const bf16_t*)dy, dw_ohwi, (int64_t)slab_floats, B, H, W, OH, OW, (OH + TH - 1) / TH, (OW + TW - 1) / TW, 0,
                 (unsigned)((size_t)B * H * W * 128), (unsigned)((size_t)B * OH * OW * 128)};
  const int groups = winw_groups(B, OH, OW);
  a.groups = groups;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_win_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       X_BYTES + D_BYTES);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(conv_win_wgrad_kernel, dim3((unsigned)(8 * groups)), dim3(256), X_BYTES + D_BYTES, s, a);
  WSMG_RETURN_LAUNCH();
}
