// Direct convolution with an LDS-resident input window, for the 64 -> 64 channel k8 s2 p3 stem of the map encoder
// (map_encoder.py:29-31, `MapEncoder.cnn[0]`; 671 GFLOP at B = 512, the largest layer of the update).
//
// Why: with Cout = 64 the implicit-GEMM kernel (wsmg_conv_bf16.hip) re-fetches the A operand once per tap through the
// 64 B/clk/CU vector-memory path: 16 KB of A + 8 KB of B per tap and 128-pixel tile against 256 MFMA clocks, so that
// path is the bound (37 % MFMA-busy).  Every input pixel of this layer is used by (k/s)^2 = 16 taps.  Here a workgroup
// owns a 5 x 25 block of output pixels (125 of the 128 MFMA rows) and loads their 16 x 56-pixel input window ONCE
// (114 KB from memory instead of 1 MB through L1); the 64 taps read their A fragments straight out of the window.
// Only the 8 KB weight slice of a tap still streams in (double-buffered, one barrier per tap).
//
// Window layout: [row][column parity][column / 2][144 B].  A tap reads, for consecutive output pixels of a row, input
// columns 2 tx + kx — all of one parity — so in the parity-split layout they are consecutive 144-byte entries: the 16
// lanes of a ds_read_b128 phase hit 16 different 16-byte slots (9 l mod 16), conflict-free; weight rows use the same
// 144-byte pitch.  LDS: 16*2*28*144 + 3*64*144 = 156 672 B (one workgroup per CU).
//
// One workgroup per CU means one wave per SIMD: nothing hides an LDS round trip but the wave itself, so the fragments of
// tap t+1 are read into a second register set while the 8 MFMAs of tap t run (three weight buffers: tap t is being used,
// t+1 being read ahead, t+2 being filled), and the weight slice of tap t+2 is requested from memory at the same time.
#include <stdlib.h>

#include "wsmg_common.h"

namespace {

typedef __bf16 bf16_t;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WinArgs {
  const bf16_t* x;    // [B][H][W][64]
  const bf16_t* w;    // [64][K][K][64]  (OHWI)
  const float* bias;  // [64] or null
  bf16_t* y;          // [B][OH][OW][64]
  int B, H, W, OH, OW, tiles_y, tiles_x, relu;
  unsigned x_bytes, w_bytes;
  double* stats;   // or null: [nslab][2][64] float64 batch-norm sums of the output (see conv_igemm_bf16_kernel)
  int nslab;
};

__device__ __forceinline__ int xcd_swizzle(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
__device__ __forceinline__ unsigned short f2bf_bits(float f) {
  bf16_t b = (bf16_t)f;
  return __builtin_bit_cast(unsigned short, b);
}

template <int K, int S, int P, int TH, int TW>
__global__ __launch_bounds__(256) void conv_win_fwd_kernel(WinArgs a) {
  constexpr int CI = 64, CO = 64;
  constexpr int WR = (TH - 1) * S + K;   // window rows (16)
  constexpr int WC = (TW - 1) * S + K;   // window columns (56)
  constexpr int PC = WC / S;             // columns per parity plane (28)
  constexpr int PITCH = 144;             // bytes per pixel / weight row in LDS (128 + 16)
  constexpr int WIN_BYTES = WR * S * PC * PITCH;
  constexpr int BT_BYTES = CO * PITCH;
  constexpr int CHUNKS = WR * WC * 8;    // 16-byte pieces of the window
  static_assert(WC % S == 0 && CHUNKS % 256 == 0 && TH * TW <= 128 && CI == 64, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* win = lds;
  unsigned char* bt = lds + WIN_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tpi = a.tiles_y * a.tiles_x;
  const int ntiles = a.B * tpi;
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  // persistent: one workgroup per CU walks a contiguous run of tiles (neighbouring tiles share halo rows in this XCD's L2)
  const int first = xcd_swizzle(blockIdx.x, gridDim.x) * per;
  // BatchNorm sums of this workgroup's whole run of tiles (one flush at the end: per tile it was 512 float64 atomics — 5.2 M per
  // launch at B = 512, 0.07 ms of the layer)
  double st_s[2] = {0.0, 0.0}, st_q[2] = {0.0, 0.0};   // this lane's two channels (32 f + r), the rows of its half-wave
  for (int logical = first; logical < first + per && logical < ntiles; ++logical) {
  const int b = logical / tpi, t = logical - b * tpi;
  const int oy0 = (t / a.tiles_x) * TH, ox0 = (t % a.tiles_x) * TW;
  const int iy0 = oy0 * S - P, ix0 = ox0 * S - P;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes), wr_ = make_rsrc(a.w, a.w_bytes);

  // ---- weight slices of taps 0 and 1 (two 16-byte pieces per thread each) and the input window (28 pieces per thread)
  const int bco = tid >> 3, bch = tid & 7;   // + 32 rows for the second piece
  auto wload = [&](int tap, u32x4& p0, u32x4& p1) {
    p0 = buf_load16(wr_, ((bco * K * K + tap) * CI) * 2 + bch * 16);
    p1 = buf_load16(wr_, (((bco + 32) * K * K + tap) * CI) * 2 + bch * 16);
  };
  auto wstore = [&](int buf, const u32x4& p0, const u32x4& p1) {
    *reinterpret_cast<u32x4*>(bt + buf * BT_BYTES + bco * PITCH + bch * 16) = p0;
    *reinterpret_cast<u32x4*>(bt + buf * BT_BYTES + (bco + 32) * PITCH + bch * 16) = p1;
  };
  // weight slices travel memory -> registers -> LDS; a slice is requested WD taps before it is used and parked in a
  // register ring until its LDS buffer is free (stored 2 taps ahead of use): with the request and the store in the same
  // tap the kernel waited out a full memory round trip 64 times per tile (1.30 ms for the layer)
  constexpr int RING = 7;   // a slice is requested RING taps before it is stored, RING + 2 before it is used
  u32x4 ring[RING][2];
#pragma unroll
  for (int tp = 0; tp < RING; ++tp) wload(tp, ring[tp][0], ring[tp][1]);   // taps 0 .. RING-1 (0 and 1 go to LDS below)
  constexpr int PER = CHUNKS / 256;   // 28
#pragma unroll
  for (int j0 = 0; j0 < PER; j0 += 14) {
    u32x4 v[14];
    int dst[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const int c = tid + 256 * (j0 + j);
      const int pix = c >> 3, ch = c & 7;
      const int wr = pix / WC, wc = pix - wr * WC;
      const int iy = iy0 + wr, ix = ix0 + wc;
      const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const int off = ok ? (((b * a.H + iy) * a.W + ix) * CI) * 2 + ch * 16 : (int)0x80000000;
      v[j] = buf_load16(xr, off);
      dst[j] = ((wr * S + (wc % S)) * PC + wc / S) * PITCH + ch * 16;
    }
#pragma unroll
    for (int j = 0; j < 14; ++j) *reinterpret_cast<u32x4*>(win + dst[j]) = v[j];
  }
  wstore(0, ring[0][0], ring[0][1]);
  wstore(1, ring[1][0], ring[1][1]);
  wload(RING, ring[0][0], ring[0][1]);       // slot s always holds a tap = s (mod RING)
  wload(RING + 1, ring[1][0], ring[1][1]);
  __syncthreads();

  // ---- this wave's 32 output pixels (rows of the MFMA tile) and its two 32-channel column blocks
  const int r = lane & 31, h = lane >> 5;
  const int q = wave * 32 + r;
  const int qc = q < TH * TW ? q : TH * TW - 1;   // the 3 padding rows of the 128-row tile repeat the last pixel
  const int ty = qc / TW, tx = qc - ty * TW;
  const unsigned char* a_base = win + ((ty * S * S) * PC + tx) * PITCH + h * 16;
  const unsigned char* b_base = bt + r * PITCH + h * 16;
  f32x16 acc[2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[f][g] = 0.f;

  struct Frag { bf16x8 a[4], b0[4], b1[4]; };
  auto fload = [&](int tap, Frag& fr) {
    const int ky = tap / K, kx = tap % K;
    const unsigned char* ap = a_base + ((ky * S + (kx % S)) * PC + kx / S) * PITCH;
    const unsigned char* bp = b_base + (tap % 3) * BT_BYTES;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      fr.a[kc] = *reinterpret_cast<const bf16x8*>(ap + kc * 32);
      fr.b0[kc] = *reinterpret_cast<const bf16x8*>(bp + kc * 32);
      fr.b1[kc] = *reinterpret_cast<const bf16x8*>(bp + 32 * PITCH + kc * 32);
    }
  };
  Frag cur, nxt;
  fload(0, cur);
#pragma unroll
  for (int tap = 0; tap < K * K; ++tap) {
    // (1) the weight slice of tap t+2 goes to LDS FIRST: its buffer held tap t-1, which every wave finished reading a
    // barrier ago.  LDS operations of a wave complete in order, so at the end of the tap `lgkmcnt(12)` — everything but
    // the 12 fragment reads issued after these two writes — means the writes have landed, without waiting for the tail
    // of the reads (which the next tap's MFMAs wait for one by one).
    if (tap + 2 < K * K) {
      wstore((tap + 2) % 3, ring[(tap + 2) % RING][0], ring[(tap + 2) % RING][1]);
      if (tap + 2 + RING < K * K) wload(tap + 2 + RING, ring[(tap + 2) % RING][0], ring[(tap + 2) % RING][1]);   // refill the slot
    }
    // (2) The 12 fragment reads of tap t+1 are interleaved with the 8 MFMAs of tap t (MFMA, 2 reads, MFMA, 1 read, ...): a
    // wave issues in order, so 12 reads in a row hold back its first MFMA until the LDS queue (shared with the other three
    // waves' 36 reads) has taken them — LDS time and MFMA time added up instead of overlapping (580 clocks per tap) —
    // and left to itself the scheduler folds the two fragment sets into one and reads each fragment just before its use.
    __builtin_amdgcn_sched_barrier(0);
    if (tap + 1 < K * K) fload(tap + 1, nxt);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[kc], cur.b0[kc], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a[kc], cur.b1[kc], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tap + 1 < K * K) {
      // raw barrier: __syncthreads() also waits for vmcnt(0), i.e. for every weight slice still in flight in the ring —
      // a memory round trip per tap
      if (tap + 2 < K * K) asm volatile("s_waitcnt lgkmcnt(12)\n\ts_barrier" ::: "memory");
      else asm volatile("s_barrier" ::: "memory");
      cur = nxt;
    }
  }

  // ---- epilogue through LDS (the window is dead now): C layout of the 32x32 MFMA — lane holds column r, rows
  // (g & 3) + 8 (g >> 2) + 4 h — is scattered into a [pixel][64 channels] tile (144-byte pitch), which then leaves as
  // 16-byte stores, 8 lanes per pixel (the direct route was 32 two-byte stores per lane: 0.2 ms of the layer)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  unsigned okmask = 0;   // bit g: this lane's row g of the tile is an output pixel (the BatchNorm sums leave the others out)
  if (a.stats) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int qq = wave * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
      const bool ok = qq < TH * TW && oy0 + qq / TW < a.OH && ox0 + qq % TW < a.OW;
      okmask |= ok ? 1u << g : 0u;
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int co = 32 * f + r;
    const float bv = a.bias ? a.bias[co] : 0.f;
    float sv = 0.f, qv = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int row = (g & 3) + 8 * (g >> 2) + 4 * h;
      float v = acc[f][g] + bv;
      if (a.relu) v = v > 0.f ? v : 0.f;
      const unsigned short o = f2bf_bits(v);
      *reinterpret_cast<unsigned short*>(win + (wave * 32 + row) * PITCH + co * 2) = o;
      // train-mode BatchNorm sums of the ROUNDED value, straight from the registers (a second pass over the staged tile
      // — 32 two-byte LDS reads per thread — cost 0.07 ms of the layer)
      const float vr = (okmask >> g) & 1u ? __uint_as_float((unsigned)o << 16) : 0.f;
      sv += vr;
      qv = fmaf(vr, vr, qv);
    }
    st_s[f] += (double)sv;
    st_q[f] += (double)qv;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = tid + 256 * j;           // 16-byte piece: pixel c / 8, channels (c % 8) * 8 ..
    const int qq = c >> 3, ch = c & 7;
    if (qq >= TH * TW) continue;
    const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
    if (oy >= a.OH || ox >= a.OW) continue;
    const u32x4 v = *reinterpret_cast<const u32x4*>(win + qq * PITCH + ch * 16);
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.y) + (((size_t)(b * a.OH + oy) * a.OW + ox) * CO) * 2 + ch * 16) = v;
  }
  __syncthreads();   // the output tile has left LDS before the next window is written over it
  }
  if (a.stats) {   // one flush per workgroup run: the two half-waves of a lane pair hold different rows of the same channels
    const int r = threadIdx.x & 31, h = (threadIdx.x >> 5) & 1;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const double s2 = st_s[f] + __shfl_xor(st_s[f], 32, 64), q2 = st_q[f] + __shfl_xor(st_q[f], 32, 64);
      if (h == 0) {
        double* st = a.stats + (size_t)(blockIdx.x % a.nslab) * 2 * CO + 32 * f + r;
        atomicAdd(st, s2);
        atomicAdd(st + CO, q2);
      }
    }
  }
}

}  // namespace

// Forward of a 64 -> 64 channel k8 s2 p3 convolution on bf16 NHWC; returns WSMG_EINVAL for any other shape (the
// caller then uses the implicit-GEMM kernel).  relu: fused ReLU after the bias.
int wsmg_conv_win_fwd_bf16(const void* x, const void* w_ohwi, const float* bias, void* y, int relu, double* stats, int nslab, int B,
                           int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW, hipStream_t s) {
  if (Cin != 64 || Cout != 64 || KH != 8 || KW != 8 || stride != 2 || pad != 3) return WSMG_EINVAL;
  if ((size_t)B * H * W * 128 >= (1ull << 31)) return WSMG_EINVAL;
  constexpr int TH = 5, TW = 25;
  WinArgs a{(const bf16_t*)x, (const bf16_t*)w_ohwi, bias, (bf16_t*)y, B, H, W, OH, OW, (OH + TH - 1) / TH, (OW + TW - 1) / TW, relu,
            (unsigned)((size_t)B * H * W * 128), (unsigned)(64u * 8 * 8 * 64 * 2), stats, nslab};
  auto kern = conv_win_fwd_kernel<8, 2, 3, TH, TW>;
  constexpr int LDS = (((TH - 1) * 2 + 8) * 2 * (((TW - 1) * 2 + 8) / 2) + 3 * 64) * 144;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  int ntiles = B * a.tiles_y * a.tiles_x, grid = ntiles < 256 ? ntiles : 256;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), LDS, s, a);
  WSMG_RETURN_LAUNCH();
}
