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
#include "wsmg_relu_mask.h"

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

// raw buffer resource over a tensor: 32-bit byte offsets, out-of-range lanes read 0 (no branches)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}

struct ConvArgsB {
  const bf16_t* src;  // [B][SH][SW][Kc]
  const bf16_t* wt;   // [N][KH][KW][Kc]
  const float* bias;  // [N] or null
  void* dst;          // [B][TH][TW][N] bf16 or f32
  int B, SH, SW, Kc, TH, TW, N, KH, KW, stride, pad;
  int mtiles, ntiles;
  int out_f32;
  unsigned src_bytes, wt_bytes;
  double* stats;   // or null: [nslab][2][N] float64 sums / sums of squares of the (bf16-rounded) output, accumulated into
  int nslab;
  // split-K launches (SPLIT kernels only): `ksplit` workgroups share one output tile, each over a contiguous run of k-steps
  int ksplit;
  float* part;      // [ksplit][pixels][N] float32 partial sums; splitk_finish_kernel adds them and runs the epilogue
  // round 6 (bf16 16-byte-store epilogue only): pixel pitch of dst in elements (0 = N: > N writes a channel slice of a wider tensor —
  // the concatenation that follows — in place) and the ReLU outputs [pixels][N] the gradient tile is masked with (wsmg_relu_mask.h)
  int dst_ld;
  const bf16_t* relu_z;
  void* dst2;       // output channels >= split_c go here ([pixels][N - split_c]) and the others to dst ([pixels][split_c]); null: one tensor
  int split_c;
};

// PF = k-steps of global loads kept in flight in registers (a bf16 k-step is only 256-512 MFMA
// cycles, far less than the L2/HBM latency, so one step of look-ahead leaves the loads exposed).
//
// SPLIT (rollout-size layers: a 7 x 7 map with 512 channels is ONE 128-pixel tile and 144 serial k-steps on 8 of the 256
// CUs): the k-steps of a tile are divided over `ksplit` workgroups; each leaves its float32 accumulators in `part`, and
// splitk_finish_kernel adds the partials in split order and runs the epilogue.  (One launch with a ticket per tile — the
// workgroup that arrives last reduces — was tried first: the device-scope fences around the ticket write back the whole L2
// and the last workgroup reads all the partials alone; 52 instead of 58 us for the 7 x 7 layer, 21 instead of 10 us for a
// 2-way split.)
template <bool BWD, int BN, int BK, int PF, bool SPLIT = false>
__global__ __launch_bounds__(256) void conv_igemm_bf16_kernel(ConvArgsB a) {
  constexpr int LDB = BK * 2 + 16;           // LDS row stride in bytes
  constexpr int SEGS = BK / 8;               // 16-B segments per row (8 bf16 each)
  constexpr int RPP = 256 / SEGS;            // rows staged per pass
  constexpr int APASS = BM / RPP;            // A passes per thread
  constexpr int BPASS = BN / RPP;            // B passes per thread
  constexpr int NT = BN / 64;                // 32-wide n-tiles per wave
  constexpr int OPITCH = BN * 2 + 16;   // row pitch of the output tile that is staged in the same memory by the epilogue
  constexpr int SMEM = (BM + BN) * LDB > BM * OPITCH ? (BM + BN) * LDB : BM * OPITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];   // A tile, B tile; the output tile afterwards
  unsigned char* const As = smem;
  unsigned char* const Bs = smem + BM * LDB;
  __shared__ int dpix[BM];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int tiles = a.mtiles * a.ntiles;
  const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
  const int split = SPLIT ? swz / tiles : 0;
  const int logical = SPLIT ? swz - split * tiles : swz;
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

  // Address generation is kept off the critical path (the kernel is instruction-issue bound once the
  // MFMAs are bf16): per staged row ONE 32-bit base offset and two validity bitmasks (bit t = class tap
  // t in range) are computed once; per k-step the uniform tap/chunk iterator contributes one scalar
  // delta; invalid rows get offset -1 and the buffer load returns zeros.
  // Source coordinate of class tap (ay, ax): sy = py + sg*ay, sx = px + sg*ax  (sg = +1 fwd, -1 bwd).
  const int sg = BWD ? -1 : 1;
  int roff[APASS];
  unsigned ymask[APASS], xmask[APASS];
  // (image, row, column) of this thread's first staged row by division, of the rows RPP further on by stepping (round 6: the two
  // runtime-divisor divisions per pass were ~1 000 VALU cycles per tile of a kernel that is issue-bound and has 9-18 k-steps in the
  // small layers).  Rows past the last pixel carry coordinates that are never used: their masks are empty.
  int rb_ = (m0 + lrow) / (THc * TWc);
  int ry_, rx_;
  {
    const int r0 = (m0 + lrow) - rb_ * (THc * TWc);
    ry_ = r0 / TWc;
    rx_ = r0 - ry_ * TWc;
  }
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    int m = m0 + lrow + RPP * i;
    bool pv = m < Mc;
    const int b = pv ? rb_ : 0, iy = pv ? ry_ : 0, ix = pv ? rx_ : 0;
    rx_ += RPP;
    while (rx_ >= TWc) { rx_ -= TWc; ++ry_; }
    while (ry_ >= THc) { ry_ -= THc; ++rb_; }
    int ty = ty0 + iy * tstep, tx = tx0 + ix * tstep;
    int py, px;
    if (BWD) {
      py = (a.stride == 2) ? ((ty + a.pad - cy) >> 1) : ty + a.pad;
      px = (a.stride == 2) ? ((tx + a.pad - cx) >> 1) : tx + a.pad;
    } else {
      py = ty * a.stride - a.pad;
      px = tx * a.stride - a.pad;
    }
    // taps t with 0 <= p + sg t < S form one interval [lo, hi]: closed form (the tap loops compiled to a chain of
    // data-dependent branches, run 2 x APASS times by every workgroup: a noticeable prologue for the 9-k-step launches)
    auto tapmask = [&](int p, int S_, int n) -> unsigned {
      int lo = BWD ? p - (S_ - 1) : -p, hi = BWD ? p : S_ - 1 - p;
      lo = lo < 0 ? 0 : lo;
      hi = hi > n - 1 ? n - 1 : hi;
      return hi >= lo ? ((2u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    const unsigned ym = tapmask(py, a.SH, KHc), xm = tapmask(px, a.SW, KWc);
    ymask[i] = pv ? ym : 0u;
    xmask[i] = xm;
    roff[i] = ((b * a.SH + py) * a.SW + px) * a.Kc + seg * 8;
    if (seg == 0) dpix[lrow + RPP * i] = pv ? (b * a.TH + ty) * a.TW + tx : -1;
  }
  const int taps = a.KH * a.KW;
  int woff[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    int n = n0 + lrow + RPP * i;
    woff[i] = n < a.N ? n * taps * a.Kc + seg * 8 : -0x40000000;  // far out of range -> zeros
  }
  const int kchunks = a.Kc / BK;
  const int all_steps = KHc * KWc * kchunks;
  const int per_split = SPLIT ? (all_steps + a.ksplit - 1) / a.ksplit : all_steps;
  const int step_begin = split * per_split;
  const int steps = SPLIT ? (all_steps - step_begin < per_split ? all_steps - step_begin : per_split) : all_steps;
  const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(a.wt, a.wt_bytes);

  // uniform k-step iterator: channel chunk outermost, taps innermost (consecutive steps re-read the
  // same pixels shifted by one tap: L1-friendly); advanced once per gload call, never divided.
  int it_ch = 0, it_ay = 0, it_ax = 0, it_left = steps;
  if (SPLIT) {   // the iterator starts at this workgroup's first k-step
    const int tp = KHc * KWc;
    it_ch = step_begin / tp;
    const int rem = step_begin - it_ch * tp;
    it_ay = rem / KWc;
    it_ax = rem - it_ay * KWc;
  }
  u32x4 ra[PF][APASS], rb[PF][BPASS];
  auto gload = [&](u32x4 (&ra)[APASS], u32x4 (&rb)[BPASS]) {
    const int ky = BWD ? cy + it_ay * tstep : it_ay, kx = BWD ? cx + it_ax * tstep : it_ax;
    const int sdelta = sg * (it_ay * a.SW + it_ax) * a.Kc + it_ch * BK;
    // past the last k-step (the loop runs a multiple of PF steps) every piece is out of range -> zeros
    const bool more = SPLIT ? it_left > 0 : it_ch < kchunks;
    if (SPLIT) --it_left;
    const int wdelta = more ? (ky * a.KW + kx) * a.Kc + it_ch * BK : -0x20000000;
    const unsigned live = more ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      bool ok = ((ymask[i] >> it_ay) & (xmask[i] >> it_ax) & live) != 0;
      ra[i] = buf_load16(rs_src, ok ? (roff[i] + sdelta) * 2 : -1);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) rb[i] = buf_load16(rs_wt, (woff[i] + wdelta) * 2);
    if (++it_ax == KWc) { it_ax = 0; if (++it_ay == KHc) { it_ay = 0; ++it_ch; } }
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

  // register set j holds the k-step with (step % PF == j); LDS holds the current step.  The loop body is
  // branch-free and runs a multiple of PF steps (the surplus steps load zeros): with conditional steps
  // the compiler routes the loop-carried accumulators through VGPR copies (64-128 v_accvgpr moves per
  // iteration), which costs more than the padding.
  gload(ra[0], rb[0]);
  lstore(ra[0], rb[0]);
  __syncthreads();
#pragma unroll
  for (int j = 1; j < PF; ++j) gload(ra[j], rb[j]);
  for (int step0 = 0; step0 < steps; step0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      gload(ra[j], rb[j]);  // set j is free: its step is in LDS
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
      lstore(ra[(j + 1) % PF], rb[(j + 1) % PF]);
      __syncthreads();
    }
  }

  // bf16 output, no accumulate: the tile is transposed through LDS (the operand tiles are dead) and leaves as 16-byte
  // stores, 8 channels per lane — the C layout of the 32x32 MFMA (a lane holds ONE channel of 16 rows) made the direct
  // route 16 two-byte stores per accumulator tile and lane, which was a third of the short-reduction launches
  // (backward-data of the 64-channel layers: 9 k-steps per tile)
  if (SPLIT) {
    float* const mine = a.part + (size_t)split * Mc * a.N;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int n = n0 + wn + 32 * u + r;
      if (n >= a.N) continue;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int m = m0 + wm + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
          if (m < Mc) mine[(size_t)m * a.N + n] = acc[t][u][g];     // (forward, one class: pixel index = row index)
        }
    }
    return;
  }

  constexpr int OP = OPITCH;
  {
    if ((a.out_f32 & 1) == 0 && (a.N & 7) == 0) {
      // (flag 4, add to the existing output: the sum and the ReLU run in the 16-byte store loop below, on the bf16-rounded
      //  tile — the arithmetic of conv -> bf16, then a separate add + ReLU pass, without that pass)
      const bool relu_here = (a.out_f32 & 6) == 2;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int col = wn + 32 * u + r;
        const float bv = (a.bias && n0 + col < a.N) ? a.bias[n0 + col] : 0.f;
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g = 0; g < 16; ++g) {
            const int row = wm + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
            float v = acc[t][u][g] + bv;
            if (relu_here) v = v > 0.f ? v : 0.f;
            const unsigned short o = f2bf_bits(v);
            *reinterpret_cast<unsigned short*>(smem + row * OP + col * 2) = o;
            if (a.stats) {   // (rows past the last pixel of the class are the only invalid ones: dpix[row] < 0 <=> m0 + row >= Mc)
              const float vr = m0 + row < Mc ? __uint_as_float((unsigned)o << 16) : 0.f;
              sv += vr;
              qv = fmaf(vr, vr, qv);
            }
          }
        if (a.stats) {
          // Train-mode BatchNorm statistics of THIS output (map_encoder.py:10-12: every conv of the map stack feeds one):
          // per-channel sum and sum of squares of the bf16-ROUNDED values, taken from the registers while the tile is
          // staged (a second pass over the staged tile was 32-64 two-byte LDS reads per thread), added to one of `nslab`
          // float64 slabs (slab = m-tile mod nslab: 72 adders per address instead of 4608) — the separate statistics pass
          // over y (one full read of every conv output) disappears.
          sv += __shfl_xor(sv, 32, 64);
          qv += __shfl_xor(qv, 32, 64);
          if (h == 0 && n0 + col < a.N) {
            double* st = a.stats + (size_t)((logical / a.ntiles) % a.nslab) * 2 * a.N + n0 + col;
            atomicAdd(st, (double)sv);
            atomicAdd(st + a.N, (double)qv);
          }
        }
      }
      __syncthreads();
      constexpr int CPR = BN / 8;   // 16-byte pieces per row
      const int ldd = a.dst_ld ? a.dst_ld : a.N;
      // (the pieces of relu_z are requested together, ahead of the store loop: see conv_win3's epilogue)
      constexpr int ROWS = BM * CPR / 256;
      u32x4 zr[ROWS];
      if (a.relu_z) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
          const int c = tid + 256 * j;
          const int row = c / CPR, ch = c - row * CPR;
          const int dp = dpix[row];
          const int n = n0 + ch * 8;
          zr[j] = (dp >= 0 && n < a.N) ? *reinterpret_cast<const u32x4*>(a.relu_z + (size_t)dp * a.N + n) : u32x4{0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int j = 0; j < BM * CPR / 256; ++j) {
        const int c = tid + 256 * j;
        const int row = c / CPR, ch = c - row * CPR;
        const int dp = dpix[row];
        const int n = n0 + ch * 8;
        if (dp < 0 || n >= a.N) continue;
        u32x4* out = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.dst) + ((size_t)dp * ldd + n) * 2);
        if (a.dst2) {
          const bool hi = n >= a.split_c;
          out = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(hi ? a.dst2 : a.dst) +
                                         ((size_t)dp * (hi ? a.N - a.split_c : a.split_c) + (hi ? n - a.split_c : n)) * 2);
        }
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * OP + ch * 16);
        if (a.relu_z) v = relu_mask8(v, zr[j]);
        if (a.out_f32 & 4) {
          const u32x4 old = *out;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float lo = __uint_as_float(v[q] << 16) + __uint_as_float(old[q] << 16);
            float hi = __uint_as_float(v[q] & 0xffff0000u) + __uint_as_float(old[q] & 0xffff0000u);
            if (a.out_f32 & 2) { lo = lo > 0.f ? lo : 0.f; hi = hi > 0.f ? hi : 0.f; }
            v[q] = (unsigned)f2bf_bits(lo) | ((unsigned)f2bf_bits(hi) << 16);
          }
        }
        *out = v;
      }
      return;
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
        if (a.out_f32 & 4) {   // accumulate into the existing output (a convolution over concatenated inputs, part by part)
          const size_t o = (size_t)dp * a.N + n;
          v += (a.out_f32 & 1) ? reinterpret_cast<const float*>(a.dst)[o]
                               : __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(a.dst)[o] << 16);
        }
        if (a.out_f32 & 2) v = v > 0.f ? v : 0.f;   // fused ReLU
        if (a.out_f32 & 1)
          reinterpret_cast<float*>(a.dst)[(size_t)dp * a.N + n] = v;
        else
          reinterpret_cast<unsigned short*>(a.dst)[(size_t)dp * a.N + n] = f2bf_bits(v);
      }
    }
  }
}

// ----------------------------------------------------------------------------- backward weight
// dW[co][tap][ci] = sum over pixels p of dY[p][co] * X[p shifted by tap][ci]: both operands have the pixel
// axis as K, which is the strided axis of NHWC, so both go through LDS and are fetched with the
// transposing read.  A workgroup owns 32*TC output channels x 4*TU "units" (unit = one tap x 32 input
// channels) over a slice of the pixel reduction; wave w owns units 4w*TU/4.. i.e. TU consecutive units,
// and every wave reuses its TC dY fragments across its TU X fragments (LDS reads per MFMA:
// (2 TC + 2 TU) / (TC TU) -> 3 for <2,1>, 2 for <2,2>, 1.5 for <4,2>; the kernel is LDS-bandwidth-bound).
constexpr int WKP = 64;          // pixels per k-step
constexpr int X_LD = 64;         // X tile row stride (bytes): 32 bf16, rows land 16 banks apart

struct WgradArgsB {
  const bf16_t* x;   // [B][H][W][Cin]
  const bf16_t* dy;  // [B][OH][OW][Cout]
  float* dw;         // [Cout][KH][KW][Cin] f32
  int B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW;
  int units;
  int64_t npix, chunk;
  unsigned x_bytes, dy_bytes;
  int gx, gy;   // unit tiles, co tiles (1-D grid of gx*gy*gz blocks, XCD-swizzled)
  int64_t slab;  // > 0: floats per slab — `dw` is a workspace [gz][Cout][KH][KW][Cin] and the workgroup of pixel chunk z STORES its
                 // partial tile into slab z (plain stores: no atomics, no zero fill — every element of every slab is written once);
                 // wsmg_weight_grad_reduce_oihw adds the slabs in slab order, so dW is bit-reproducible (run.py:107-108 of the
                 // reference asks for deterministic kernels).  0: float atomics into a zeroed OHWI dW
};

__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)(p));
}

template <int TC, int TU, int PF, int UPT>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(WgradArgsB a) {
  constexpr int WCO = 32 * TC;              // output channels per workgroup
  constexpr int WUN = 4 * TU;               // units per workgroup
  constexpr int NG = WUN / UPT;             // tap groups: UPT consecutive units share one tap (Cin/32 % UPT == 0)
  constexpr int D_LD = TC == 4 ? 320 : 192; // dY tile row stride (bytes): 4 consecutive rows 64 B apart mod 256
  constexpr int DSEG = 4 * TC;              // 16-B pieces per dY row
  constexpr int DROWS = 256 / DSEG;         // dY rows staged per pass (TC passes)
  __shared__ __attribute__((aligned(16))) unsigned char Ds[WKP * D_LD];
  __shared__ __attribute__((aligned(16))) unsigned char Xs[WUN * WKP * X_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // all (unit, co) tiles of one pixel chunk are consecutive logical blocks => same XCD => the chunk's
  // dY / X slices are fetched into ONE L2 instead of eight
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

  // a group's UPT units are consecutive 32-channel chunks of ONE tap: one bounds test and one base offset
  // per group and k-step, the units inside it differ by a 64-byte immediate
  int gky[NG], gkx[NG], gci[NG];
  bool gok[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    int uu = u0 + g * UPT;
    gok[g] = uu < a.units;
    int t = gok[g] ? uu / cchunks : 0;
    gci[g] = gok[g] ? (uu - t * cchunks) * 32 : 0;
    gky[g] = t / a.KW;
    gkx[g] = t - gky[g] * a.KW;
  }
  // staging maps: dY tile 64 px x 64*TC B = 256*TC 16-B pieces (TC per thread); X tile per unit
  // 64 px x 64 B = 256 pieces (1 per thread and unit)
  const int dpx = tid / DSEG, dseg = tid % DSEG;   // + DROWS rows per pass
  const int xpx = tid >> 2, xseg = tid & 3;

  f32x16 acc[TC][TU];
#pragma unroll
  for (int t = 0; t < TC; ++t)
#pragma unroll
    for (int u = 0; u < TU; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

  // Address generation without per-step divisions or branches (the kernel is issue-bound): each thread
  // keeps the (image, oy, ox) of the pixel it stages and advances it by WKP pixels per k-step; unit
  // offsets are thread-invariant; invalid pieces get offset -1 and the buffer load returns zeros.
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(a.dy, a.dy_bytes);
  const bool co_ok = (co0 + dseg * 8) < a.Cout;
  int gdelta[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) gdelta[g] = (gky[g] * a.W + gkx[g]) * a.Cin + gci[g] + xseg * 8;
  // pixel iterator of the X staging row (pixel p_begin + xpx + k*WKP)
  int it_b, it_oy, it_ox;
  {
    int64_t p = p_begin + xpx;
    if (p >= a.npix) p = a.npix - 1;
    it_b = (int)(p / ohw);
    int rem = (int)(p - (int64_t)it_b * ohw);
    it_oy = rem / a.OW;
    it_ox = rem - it_oy * a.OW;
  }
  const int adv_b = WKP / ohw, adv_rem = WKP - adv_b * ohw;
  const int adv_oy = adv_rem / a.OW, adv_ox = adv_rem - adv_oy * a.OW;
  int it_p = (int)p_begin;  // first pixel of the k-step about to be loaded
  const int pend = (int)p_end;
  u32x4 rd[PF][TC], rx[PF][WUN];
  auto gload = [&](u32x4 (&rd)[TC], u32x4 (&rx)[WUN]) {
#pragma unroll
    for (int i = 0; i < TC; ++i) {
      int p = it_p + dpx + DROWS * i;
      bool ok = co_ok && p < pend;
      rd[i] = buf_load16(rs_dy, ok ? (p * a.Cout + co0 + dseg * 8) * 2 : -1);
    }
    const bool pok = (it_p + xpx) < pend;
    const int y0 = it_oy * a.stride - a.pad, x0 = it_ox * a.stride - a.pad;
    const int base = ((it_b * a.H + y0) * a.W + x0) * a.Cin;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      bool ok = pok && gok[g] && (unsigned)(y0 + gky[g]) < (unsigned)a.H && (unsigned)(x0 + gkx[g]) < (unsigned)a.W;
      const int off = ok ? (base + gdelta[g]) * 2 : (int)0x80000000;   // 2 GiB + 64 j is still out of range
#pragma unroll
      for (int j = 0; j < UPT; ++j) rx[g * UPT + j] = buf_load16(rs_x, off + 64 * j);
    }
    // advance by WKP pixels
    it_p += WKP;
    it_ox += adv_ox;
    if (it_ox >= a.OW) { it_ox -= a.OW; ++it_oy; }
    it_oy += adv_oy;
    if (it_oy >= a.OH) { it_oy -= a.OH; ++it_b; }
    it_b += adv_b;
  };
  auto lstore = [&](const u32x4 (&rd)[TC], const u32x4 (&rx)[WUN]) {
#pragma unroll
    for (int i = 0; i < TC; ++i) *reinterpret_cast<u32x4*>(&Ds[(dpx + DROWS * i) * D_LD + dseg * 16]) = rd[i];
#pragma unroll
    for (int u = 0; u < WUN; ++u) *reinterpret_cast<u32x4*>(&Xs[(u * WKP + xpx) * X_LD + xseg * 16]) = rx[u];
  };

  // transposing-read lane map (every lane participates: EXEC must be full)
  const int li = lane & 15, q = li >> 2, p4 = li & 3;
  const int half16 = (lane >> 4) & 1;  // which 16 of the 32 MFMA rows/cols this 16-lane group owns
  const int h = lane >> 5;             // k half: pixels 8h .. 8h+7 of a 16-pixel MFMA step
  const unsigned char* a_base = &Ds[(8 * h + q) * D_LD + (half16 * 16 + p4 * 4) * 2];   // + 64 t: co 32t..32t+31
  const unsigned char* b_base = &Xs[(wave * TU * WKP + 8 * h + q) * X_LD + (half16 * 16 + p4 * 4) * 2];

  // branch-free main loop over a multiple of PF k-steps (pieces past p_end load zeros), see the igemm loop
  gload(rd[0], rx[0]);
  lstore(rd[0], rx[0]);
  __syncthreads();
#pragma unroll
  for (int j = 1; j < PF; ++j) gload(rd[j], rx[j]);
  for (int64_t pb = p_begin; pb < p_end; pb += (int64_t)PF * WKP) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      gload(rd[j], rx[j]);
#pragma unroll
      for (int ks = 0; ks < WKP / 16; ++ks) {
        bf16x8 af[TC], bfr[TU];
#pragma unroll
        for (int t = 0; t < TC; ++t) {
          bf16x4 l = tr_read(a_base + 64 * t + (16 * ks) * D_LD), hh = tr_read(a_base + 64 * t + (16 * ks + 4) * D_LD);
          af[t] = __builtin_shufflevector(l, hh, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
          const unsigned char* bp = b_base + u * WKP * X_LD;
          bf16x4 l = tr_read(bp + (16 * ks) * X_LD), hh = tr_read(bp + (16 * ks + 4) * X_LD);
          bfr[u] = __builtin_shufflevector(l, hh, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
          for (int t = 0; t < TC; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], bfr[u], acc[t][u], 0, 0, 0);
      }
      __syncthreads();
      lstore(rd[(j + 1) % PF], rx[(j + 1) % PF]);
      __syncthreads();
    }
  }
  const int r = lane & 31;
  const int taps = a.KH * a.KW;
#pragma unroll
  for (int u = 0; u < TU; ++u) {
    const int uu = u0 + wave * TU + u;
    if (uu >= a.units) continue;
    const int tap = uu / cchunks;
    const int ci = (uu - tap * cchunks) * 32 + r;
#pragma unroll
    for (int t = 0; t < TC; ++t) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int co = co0 + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (co < a.Cout) {
          float* const q = a.dw + (size_t)bz * a.slab + ((size_t)co * taps + tap) * a.Cin + ci;
          if (a.slab) *q = acc[t][u][g];      // lanes = consecutive input channels: two 128-byte segments per instruction
          else atomicAdd(q, acc[t][u][g]);
        }
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
  // 32-bit byte offsets into the buffer resources
  if ((int64_t)B * H * W * Cin * 2 >= (1ll << 31) || (int64_t)B * OH * OW * Cout * 2 >= (1ll << 31)) return WSMG_EINVAL;
  return 0;
}

// register prefetch depth (k-steps in flight).  Measured on MI355X (tools/bench_conv.py): the igemm
// kernels are fastest at 1 (deeper costs occupancy, which hides more latency than the prefetch does).
int conv_prefetch(int dflt) { return dflt; }

int g_win3_tile = -1;
int win3_tile() {   // WSMG_CONV_WIN3: 0 = off, 1 = by shape (default), 512 / 256 = that many pixels per workgroup
  if (g_win3_tile < 0) g_win3_tile = WSMG_TUNE("WSMG_CONV_WIN3", 1);
  return g_win3_tile;
}
// The LDS-window kernel (wsmg_conv_win3.hip) for a 3 x 3 / stride 1 / pad 1 layer of M pixels, Kc reduction channels and N output
// channels: 0 = no (implicit-GEMM kernel), else pixels per workgroup.  Measured at B = 512 (tools/ab_win3.sh): at Kc = 32 (9
// k-steps) its prologue costs more than the window saves; 512-pixel tiles move half the weight bytes per MFMA of 256-pixel
// ones but need >= 4 rounds of workgroups over the 256 CUs to keep the last round's idle CUs cheap (N = 128 layers: 256).
// round 6: a 32-channel reduction axis takes wsmg_conv_win3_k32.hip (by-shape choice only: a forced tile keeps the general window kernel,
// which is how the tests hold the two against each other; WSMG_CONV_K32=0: A/B)
bool k32_choice(int64_t M, int Kc, int N) {
  // (N = 128, the classifier's 32 -> 128 projection: 41 us on either kernel alone, 48.7 against 55.6 us in the update — rocprofv3 --stats)
  return win3_tile() == 1 && Kc == 32 && (N == 32 || N == 64 || N == 128) && M >= 2 * 256 * 256 && WSMG_TUNE("WSMG_CONV_K32", 1) != 0;
}
int win3_choice(int64_t M, int Kc, int N) {
  const int t = win3_tile();
  if (t == 0 || M < 256 * 256 || Kc % 32 || N % 32) return 0;
  if (N % 64) {                      // 32-channel tiles (round 3; WSMG_CONV_WIN3_N32=0: implicit-GEMM kernel, 512 / 256: the tile).
    // The classifier's 32 -> 32 layer at 48 x 48 (B = 512), alone: forward 0.071 -> 0.063 (512-pixel tiles) -> 0.055 ms (256),
    // backward-data 0.075 -> 0.063 -> 0.056: the implicit-GEMM kernel pads the 32 channels to a 64-wide tile and re-fetches the
    // pixels per tap; nine k-steps are too few to hide the window's load, so this is 390 TFLOP/s, not 800
    const int n32 = WSMG_TUNE("WSMG_CONV_WIN3_N32", 256);
    if (t != 1) return t;
    if (n32 == 0 || M < 2 * 256 * 256) return 0;
    return n32;
  }
  if (Kc < 64) return 0;
  if (N % 128) {                     // 64-channel tiles (round 3; WSMG_CONV_WIN3_N64=0: the implicit-GEMM kernel, 512 / 256: the tile)
    const int n64 = WSMG_TUNE("WSMG_CONV_WIN3_N64", 1);
    if (t != 1) return t;           // (forced by wsmg_conv_debug_win3_tile / WSMG_CONV_WIN3)
    // measured alone at B = 512, 24 x 24 (tools/bench_conv.py): orig0 (256 -> 64) forward 0.118 -> 0.103 ms, orig2 (192 -> 64)
    // forward 0.091 -> 0.083 and backward-data (64 -> 192) 0.106 -> 0.090; orig1 (64 -> 64: 18 k-steps, one 64-channel tile)
    // 0.037 -> 0.043: that one stays on the implicit-GEMM kernel
    if (n64 == 0 || M < 2 * 256 * 256 || (Kc < 128 && N < 192)) return 0;
    return n64 == 1 ? 512 : n64;
  }
  if (t != 1) return t;
  if (M < 2 * 256 * 256) return 0;   // 12 x 12 maps at B = 512 (73 728 pixels, 128 -> 128): 0.035 vs 0.033 ms
  return ((M + 511) / 512) * (N / 128) >= 1024 ? 512 : 256;
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
int launch_igemm(ConvArgsB& a, int64_t mrows, int classes, hipStream_t s) {
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
  return 0;
}

// split-K plan of a forward layer: 1 = none.  Only layers that leave most of the chip idle (fewer than 128 tiles of 128 pixels
// x 64 channels) with a long reduction (>= 16 k-steps) are split, and every split keeps at least 4 k-steps: the finishing
// launch costs about what 10 k-steps do.
int splitk_plan(int64_t M, int Kc, int N, int KH, int KW) {
  const int bk = (Kc % 64) == 0 ? 64 : 32;
  const int steps = KH * KW * (Kc / bk);
  const int tiles = (int)(wsmg_cdiv(M, BM) * wsmg_cdiv(N, 64));
  const int target = 256, minsteps = 4, maxtiles = 128;
  if (tiles >= maxtiles || steps < 4 * minsteps || (N & 7)) return 1;
  int ks = (int)wsmg_cdiv(target, tiles);
  if (ks > steps / minsteps) ks = steps / minsteps;
  if (ks > 32) ks = 32;
  if (ks < 2) return 1;
  const int per = (int)wsmg_cdiv(steps, ks);
  return (int)wsmg_cdiv(steps, per);     // no empty split
}

// y[m][n..n+7] = epilogue(sum over splits of part[s][m][n..n+7]): bias, + the existing y (flag 4), ReLU (flag 2), bf16 or
// float32 (flag 1) — one thread per 8 channels.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                            void* __restrict__ y, int64_t MN8, int N, int ksplit, int flags) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= MN8) return;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  const int n = (int)((i * 8) % N);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  const f32x4v* p = reinterpret_cast<const f32x4v*>(part) + i * 2;
  for (int s = 0; s < ksplit; ++s, p += MN8 * 2) {
    const f32x4v a0 = p[0], a1 = p[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] += a0[j]; v[4 + j] += a1[j]; }
  }
  if (bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += bias[n + j];
  }
  if (flags & 1) {
    f32x4v* o = reinterpret_cast<f32x4v*>(y) + i * 2;
    if (flags & 4) {
      const f32x4v a0 = o[0], a1 = o[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += a0[j]; v[4 + j] += a1[j]; }
    }
    if (flags & 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    o[0] = f32x4v{v[0], v[1], v[2], v[3]};
    o[1] = f32x4v{v[4], v[5], v[6], v[7]};
  } else {
    u32x4* o = reinterpret_cast<u32x4*>(y) + i;
    if (flags & 4) {
      const u32x4 old = *o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[2 * j] += __uint_as_float(old[j] << 16);
        v[2 * j + 1] += __uint_as_float(old[j] & 0xffff0000u);
      }
    }
    u32x4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = v[2 * j], hi = v[2 * j + 1];
      if (flags & 2) { lo = lo > 0.f ? lo : 0.f; hi = hi > 0.f ? hi : 0.f; }
      out[j] = (unsigned)f2bf_bits(lo) | ((unsigned)f2bf_bits(hi) << 16);
    }
    *o = out;
  }
}

}  // namespace

extern "C" int wsmg_conv2d_splitk_plan(int B, int OH, int OW, int Cin, int Cout, int KH, int KW, int* ksplit,
                                       long long* part_floats) {
  if (!ksplit || !part_floats) return WSMG_EINVAL;
  *ksplit = splitk_plan((int64_t)B * OH * OW, Cin, Cout, KH, KW);
  *part_floats = (long long)*ksplit * B * OH * OW * Cout;
  return 0;
}

// Forward convolution with the reduction split over `ksplit` workgroups per tile (wsmg_conv2d_splitk_plan gives ksplit and the
// size of `part`), two launches: partial sums, then sum + epilogue.  flags as wsmg_conv2d_fwd_bf16.
extern "C" int wsmg_conv2d_fwd_bf16_splitk(const void* x, const void* w_ohwi, const float* bias, void* y, int flags, int ksplit,
                                           float* part, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                           int stride, int pad, int OH, int OW, wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  const int64_t M = (int64_t)B * OH * OW;
  if (ksplit < 2 || ksplit != splitk_plan(M, Cin, Cout, KH, KW) || !part || (flags & ~7)) return WSMG_EINVAL;
  ConvArgsB a{(const bf16_t*)x, (const bf16_t*)w_ohwi, nullptr, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, 0, 0, flags,
              (unsigned)((size_t)B * H * W * Cin * 2), (unsigned)((size_t)Cout * KH * KW * Cin * 2), nullptr, 0, ksplit, part};
  a.mtiles = (int)wsmg_cdiv(M, BM);
  a.ntiles = (int)wsmg_cdiv(Cout, 64);
  dim3 grid((unsigned)(a.mtiles * a.ntiles * ksplit));
  if (Cin % 64 == 0)
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<false, 64, 64, 1, true>), grid, dim3(256), 0, wsmg_s(stream), a);
  else
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<false, 64, 32, 1, true>), grid, dim3(256), 0, wsmg_s(stream), a);
  const int64_t mn8 = M * Cout / 8;
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)wsmg_cdiv(mn8, 256)), dim3(256), 0, wsmg_s(stream), part, bias, y, mn8,
                     Cout, ksplit, flags);
  WSMG_RETURN_LAUNCH();
}

namespace {
// y_ld (0 = Cout): pixel pitch of y in elements — y may be a channel slice of a wider tensor
int conv_fwd_bf16_impl(const void* x, const void* w_ohwi, const float* bias, void* y, int out_f32, double* stats, int nslab, int y_ld, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       int OH, int OW, wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  if (stats && (nslab <= 0 || (out_f32 & 5) != 0 || (Cout & 7) != 0)) return WSMG_EINVAL;   // statistics: bf16 output, no accumulate
  const bool ex = y_ld && y_ld != Cout;
  if (ex) {
    if ((out_f32 & 5) != 0 || (Cout & 7) != 0) return WSMG_EINVAL;
    if (y_ld && (y_ld < Cout || (y_ld & 7) || ((uintptr_t)y & 15))) return WSMG_EINVAL;
    if ((int64_t)B * OH * OW * (y_ld ? y_ld : Cout) * 2 >= (1ll << 31)) return WSMG_EINVAL;
  }
  if (!ex && Cin == 64 && Cout == 64 && KH == 8 && KW == 8 && stride == 2 && pad == 3 && (out_f32 & 5) == 0) {
    // the map encoder's stem: direct convolution out of an LDS-resident input window (wsmg_conv_win.hip); WSMG_CONV_WIN=0
    // keeps the implicit-GEMM kernel (A/B)
    if (WSMG_TUNE("WSMG_CONV_WIN", 1)) {
      int rc = wsmg_conv_win_fwd_bf16(x, w_ohwi, bias, y, (out_f32 & 2) != 0, stats, nslab, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, wsmg_s(stream));
      if (rc != WSMG_EINVAL) return rc;
    }
  }
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && OH == H && OW == W && (out_f32 & 5) == 0) {
    if (k32_choice((int64_t)B * OH * OW, Cin, Cout)) {
      int rc = wsmg_conv_win3_k32_bf16(0, x, w_ohwi, bias, y, (out_f32 & 2) != 0, stats, nslab, B, H, W, Cout, y_ld, wsmg_s(stream));
      if (rc != WSMG_EINVAL) return rc;
    }
    if (const int mt = win3_choice((int64_t)B * OH * OW, Cin, Cout)) {
      int rc = wsmg_conv_win3_bf16(0, x, w_ohwi, bias, y, (out_f32 & 2) != 0, stats, nslab, B, H, W, Cin, Cout, mt, win3_tile() == 1, nullptr, y_ld, nullptr, 0, wsmg_s(stream));
      if (rc != WSMG_EINVAL) return rc;
    }
  }
  ConvArgsB a{(const bf16_t*)x, (const bf16_t*)w_ohwi, bias, y, B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, 0, 0, out_f32,
              (unsigned)((size_t)B * H * W * Cin * 2), (unsigned)((size_t)Cout * KH * KW * Cin * 2), stats, nslab};
  a.dst_ld = y_ld;
  if (int e = launch_igemm<false>(a, (int64_t)B * OH * OW, 1, wsmg_s(stream))) return e;
  WSMG_RETURN_LAUNCH();
}
}  // namespace

extern "C" int wsmg_conv2d_fwd_bf16_stats(const void* x, const void* w_ohwi, const float* bias, void* y, int out_f32,
                                          double* stats, int nslab, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                          int stride, int pad, int OH, int OW, wsmg_stream_t stream) {
  return conv_fwd_bf16_impl(x, w_ohwi, bias, y, out_f32, stats, nslab, 0, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, stream);
}

extern "C" int wsmg_conv2d_fwd_bf16_ex(const void* x, const void* w_ohwi, const float* bias, void* y, int flags, double* stats, int nslab,
                                       int y_ld, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                       int stride, int pad, int OH, int OW, wsmg_stream_t stream) {
  return conv_fwd_bf16_impl(x, w_ohwi, bias, y, flags, stats, nslab, y_ld, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, stream);
}

// tests / tools: choose the window kernel's tile (0 = off, 1 = by shape, 256, 512) for the calls that follow; returns the previous choice
extern "C" int wsmg_conv_debug_win3_tile(int mt) {
  const int old = win3_tile();
  if (mt == 0 || mt == 1 || mt == 256 || mt == 512) g_win3_tile = mt;
  return old;
}

extern "C" int wsmg_conv2d_fwd_bf16(const void* x, const void* w_ohwi, const float* bias, void* y, int out_f32, int B,
                                    int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                                    int OW, wsmg_stream_t stream) {
  return wsmg_conv2d_fwd_bf16_stats(x, w_ohwi, bias, y, out_f32, nullptr, 0, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, stream);
}

namespace {
int conv_bwd_data_bf16_impl(const void* dy, const void* w_ihwo, void* dx, int out_f32, double* stats, int nslab, const void* relu_y,
                            void* dx2, int split_c, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                            int OW, wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  if (stats && (nslab <= 0 || (out_f32 & 5) != 0 || (Cin & 7) != 0)) return WSMG_EINVAL;
  if (relu_y && ((out_f32 & 5) != 0 || (Cin & 7) != 0 || ((uintptr_t)relu_y & 15))) return WSMG_EINVAL;
  if (dx2 && ((out_f32 & 5) != 0 || (Cin & 7) != 0 || split_c <= 0 || split_c >= Cin || (split_c & 7) || ((uintptr_t)dx2 & 15))) return WSMG_EINVAL;
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && OH == H && OW == W && (out_f32 & 7) == 0) {
    if (!relu_y && !dx2 && k32_choice((int64_t)B * H * W, Cout, Cin)) {
      int rc = wsmg_conv_win3_k32_bf16(1, dy, w_ihwo, nullptr, dx, 0, stats, nslab, B, H, W, Cin, 0, wsmg_s(stream));
      if (rc != WSMG_EINVAL) return rc;
    }
    if (const int mt = win3_choice((int64_t)B * H * W, Cout, Cin)) {
      int rc = wsmg_conv_win3_bf16(1, dy, w_ihwo, nullptr, dx, 0, stats, nslab, B, H, W, Cout, Cin, mt, win3_tile() == 1, relu_y, 0, dx2, split_c, wsmg_s(stream));
      if (rc != WSMG_EINVAL) return rc;
    }
  }
  // the classifier's ConvTranspose2d(64 -> 32, k4, s2, p1) forward: wsmg_convt_k4s2.hip (by-shape choice only, like the k32 kernel;
  // WSMG_CONVT_K4S2=0: A/B)
  if (KH == 4 && KW == 4 && stride == 2 && pad == 1 && Cin == 32 && Cout == 64 && H == 2 * OH && W == 2 * OW && (out_f32 & 7) == 0 &&
      !relu_y && !dx2 && win3_tile() == 1 && (int64_t)B * OH * OW >= 2 * 256 * 256 && WSMG_TUNE("WSMG_CONVT_K4S2", 1) != 0) {
    int rc = wsmg_convt_k4s2_bf16(dy, w_ihwo, dx, stats, nslab, B, OH, OW, wsmg_s(stream));
    if (rc != WSMG_EINVAL) return rc;
  }
  ConvArgsB a{(const bf16_t*)dy, (const bf16_t*)w_ihwo, nullptr, dx, B, OH, OW, Cout, H, W, Cin, KH, KW, stride, pad, 0, 0, out_f32,
              (unsigned)((size_t)B * OH * OW * Cout * 2), (unsigned)((size_t)Cout * KH * KW * Cin * 2), stats, nslab};
  a.relu_z = (const bf16_t*)relu_y;
  a.dst2 = dx2;
  a.split_c = split_c;
  int classes = 1;
  int64_t mmax = (int64_t)B * H * W;
  if (stride == 2) {
    classes = 4;
    mmax = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  }
  if (int e = launch_igemm<true>(a, mmax, classes, wsmg_s(stream))) return e;
  WSMG_RETURN_LAUNCH();
}
}  // namespace

extern "C" int wsmg_conv2d_bwd_data_bf16_stats(const void* dy, const void* w_ihwo, void* dx, int out_f32, double* stats, int nslab,
                                               int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                               int OH, int OW, wsmg_stream_t stream) {
  return conv_bwd_data_bf16_impl(dy, w_ihwo, dx, out_f32, stats, nslab, nullptr, nullptr, 0, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, stream);
}

extern "C" int wsmg_conv2d_bwd_data_bf16_ex(const void* dy, const void* w_ihwo, void* dx, const void* relu_y, void* dx2, int split_c,
                                            int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                            wsmg_stream_t stream) {
  return conv_bwd_data_bf16_impl(dy, w_ihwo, dx, 0, nullptr, 0, relu_y, dx2, split_c, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, stream);
}

// ConvTranspose2d forward of the rollout route (the semantic classifier's first layer, mg_map_policy.py:59-61): the
// backward-data kernel of the adjoint convolution WITH an epilogue bias over the Cin output channels (the folded eval-mode
// BatchNorm shift) and the flag word's ReLU bit.  x = dy [B][OH][OW][Cout], y = dx [B][H][W][Cin].
extern "C" int wsmg_conv_transpose2d_infer_bf16(const void* x, const void* w_ihwo, const float* bias, void* y, int flags, int B,
                                                int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                                                int OW, wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  if (flags & ~3) return WSMG_EINVAL;
  ConvArgsB a{(const bf16_t*)x, (const bf16_t*)w_ihwo, bias, y, B, OH, OW, Cout, H, W, Cin, KH, KW, stride, pad, 0, 0, flags,
              (unsigned)((size_t)B * OH * OW * Cout * 2), (unsigned)((size_t)Cout * KH * KW * Cin * 2), nullptr, 0};
  int classes = 1;
  int64_t mmax = (int64_t)B * H * W;
  if (stride == 2) {
    classes = 4;
    mmax = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  }
  launch_igemm<true>(a, mmax, classes, wsmg_s(stream));
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_conv2d_bwd_data_bf16(const void* dy, const void* w_ihwo, void* dx, int out_f32, int B, int H, int W,
                                         int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                         wsmg_stream_t stream) {
  return wsmg_conv2d_bwd_data_bf16_stats(dy, w_ihwo, dx, out_f32, nullptr, 0, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, stream);
}

namespace {
// Which kernel takes the weight gradient of a layer, and over how many partial sums (pixel chunks / image ranges / tile groups) its
// reduction is split — the number of slabs of the deterministic form.
struct WgradPlan {
  int kind;      // 0 generic (conv_wgrad_bf16_kernel), 1 k8 stem window kernel, 2 3 x 3 window kernel, 3 k5 / k7 stride-2 window kernel
  int nsplit;
  int tc, tu, gx, gy;
  int64_t chunk;
};

WgradPlan wgrad_plan_bf16(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
  WgradPlan p{0, 0, 0, 0, 0, 0, 0};
  // the map encoder's stem: LDS-window variant (wsmg_conv_win_wgrad.hip); WSMG_WGRAD_WIN=0 keeps the generic kernel (A/B)
  if (WSMG_TUNE("WSMG_WGRAD_WIN", 1)) {
    if (const int n = wsmg_conv_win_wgrad_splits(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) { p.kind = 1; p.nsplit = n; return p; }
  }
  // the other two stride-2 layers (k5 64 -> 128, k7 256 -> 64): wsmg_conv_s2_wgrad.hip; WSMG_WGRAD_S2WIN=0 keeps the generic kernel (A/B)
  if (const int n = wsmg_conv_s2_wgrad_splits(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) { p.kind = 3; p.nsplit = n; return p; }
  // 3 x 3 stride-1 layers: zero-padded LDS window (wsmg_conv_win3_wgrad.hip); WSMG_WGRAD_WIN3=0 (or the tests' tile switch = 0)
  // keeps the generic kernel (A/B)
  const int use_w3w = WSMG_TUNE("WSMG_WGRAD_WIN3", 1);
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && OH == H && OW == W && use_w3w && win3_tile() && (int64_t)B * H * W >= 256 * 256) {
    if (const int n = wsmg_conv_win3_wgrad_splits(B, H, W, Cin, Cout)) { p.kind = 2; p.nsplit = n; return p; }
  }
  const int units = KH * KW * (Cin / 32);
  const int64_t npix = (int64_t)B * OH * OW;
  // tile shape (measured per layer, tools/bench_conv.py): 128 co x 8 units where both dimensions are
  // large (cated 765 vs 660 TFLOP/s), else 64 co x 4 units, which keeps 5 workgroups per CU resident
  // (the 64-channel k8 stem, 128 units of 64 input channels: 64 co x 8 units measured 0.94 vs 0.99 ms; every other
  // 64-wide layer is faster with 4 units)
  int tc = (Cout % 128 == 0 && units >= 48) ? 4 : 2, tu = (tc == 4 || (Cin == 64 && units >= 128)) ? 2 : 1;
  const int gx = (int)wsmg_cdiv(units, 4 * tu), gy = (int)wsmg_cdiv(Cout, 32 * tc);
  // workgroups over the pixel reduction (the k8 stem's 64 x 8 tile: 0.90 ms at 2048, 0.96 at 1024).  Every workgroup ends
  // with its tile going to memory — float atomics, 40-50 % of the launch for the small layers, or (slab form) plain stores
  // that the reduce launch reads back: below 40 GFLOP fewer, longer workgroups win (768: -0.08 ms over the six small
  // layers of the update; tools sweep, WSMG_WGRAD_WANT)
  const double gflop = 2.0 * (double)npix * Cout * Cin * KH * KW * 1e-9;
  int64_t target = tc == 4 ? 1024 : (gflop < 40.0 ? 768 : 2048);
  int64_t want = wsmg_cdiv(target, (int64_t)gx * gy);
  int64_t maxz = wsmg_cdiv(npix, WKP * 8);
  int64_t gz = want < 1 ? 1 : (want > maxz ? maxz : want);
  if (gz < 1) gz = 1;
  if (gz > 65535) gz = 65535;
  p.chunk = wsmg_cdiv(wsmg_cdiv(npix, gz), WKP) * WKP;
  p.nsplit = (int)wsmg_cdiv(npix, p.chunk);
  p.tc = tc; p.tu = tu; p.gx = gx; p.gy = gy;
  return p;
}

// slab_floats == 0: atomics into the zeroed OHWI dW; > 0: `dw` is the slab workspace (WgradArgsB::slab)
int launch_wgrad_bf16(const void* x, const void* dy, float* dw, long long slab_floats, int B, int H, int W, int Cin, int Cout, int KH,
                      int KW, int stride, int pad, int OH, int OW, hipStream_t stream) {
  const WgradPlan p = wgrad_plan_bf16(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW);
  if (p.kind == 1) return wsmg_conv_win_wgrad_bf16(x, dy, dw, slab_floats, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, stream);
  if (p.kind == 2) return wsmg_conv_win3_wgrad_bf16(x, dy, dw, slab_floats, B, H, W, Cin, Cout, stream);
  if (p.kind == 3) return wsmg_conv_s2_wgrad_bf16(x, dy, dw, slab_floats, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, stream);
  WgradArgsB a{(const bf16_t*)x, (const bf16_t*)dy, dw, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, 0, 0, 0,
               (unsigned)((size_t)B * H * W * Cin * 2), (unsigned)((size_t)B * OH * OW * Cout * 2), 0, 0, (int64_t)slab_floats};
  a.units = KH * KW * (Cin / 32);
  a.npix = (int64_t)B * OH * OW;
  a.chunk = p.chunk;
  a.gx = p.gx;
  a.gy = p.gy;
  const int tc = p.tc, tu = p.tu;
  dim3 grid((unsigned)((int64_t)p.gx * p.gy * p.nsplit));
  const int pf = conv_prefetch(1) >= 2 ? 2 : 1;
  const int cch = Cin / 32, wun = 4 * tu;
  int upt = 1;
  for (int c = wun; c > 1; c >>= 1)
    if (cch % c == 0) { upt = c; break; }
#define WSMG_WGRAD(TC_, TU_, PF_, UPT_) \
  hipLaunchKernelGGL((conv_wgrad_bf16_kernel<TC_, TU_, PF_, UPT_>), grid, dim3(256), 0, stream, a)
#define WSMG_WGRAD_U4(TC_, TU_, PF_) \
  do { if (upt >= 4) WSMG_WGRAD(TC_, TU_, PF_, 4); else if (upt == 2) WSMG_WGRAD(TC_, TU_, PF_, 2); else WSMG_WGRAD(TC_, TU_, PF_, 1); } while (0)
#define WSMG_WGRAD_U8(TC_, TU_, PF_) \
  do { if (upt == 8) WSMG_WGRAD(TC_, TU_, PF_, 8); else WSMG_WGRAD_U4(TC_, TU_, PF_); } while (0)
  if (tc == 4) {
    if (pf == 1) WSMG_WGRAD_U8(4, 2, 1); else WSMG_WGRAD_U8(4, 2, 2);
  } else if (tu == 2) {
    if (pf == 1) WSMG_WGRAD_U8(2, 2, 1); else WSMG_WGRAD_U8(2, 2, 2);
  } else {
    if (pf == 1) WSMG_WGRAD_U4(2, 1, 1); else WSMG_WGRAD_U4(2, 1, 2);
  }
#undef WSMG_WGRAD_U8
#undef WSMG_WGRAD_U4
#undef WSMG_WGRAD
  WSMG_RETURN_LAUNCH();
}
}  // namespace

extern "C" int wsmg_conv2d_bwd_weight_bf16(const void* x, const void* dy, float* dw_ohwi, int B, int H, int W, int Cin,
                                           int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                           wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  return launch_wgrad_bf16(x, dy, dw_ohwi, 0, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, wsmg_s(stream));
}

// Deterministic weight gradient, step 1 of 2 (the plan): how many partial sums (`nsplit`) the layer's reduction is split into and
// how many floats of workspace that takes — nsplit slabs of Cout*KH*KW*Cin floats.
extern "C" int wsmg_conv2d_bwd_weight_bf16_plan(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH,
                                                int OW, int* nsplit, long long* ws_floats) {
  if (!nsplit || !ws_floats) return WSMG_EINVAL;
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  const WgradPlan p = wgrad_plan_bf16(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW);
  *nsplit = p.nsplit;
  *ws_floats = (long long)p.nsplit * Cout * KH * KW * Cin;
  return 0;
}

// Deterministic weight gradient, step 2: the same kernels as wsmg_conv2d_bwd_weight_bf16, but every workgroup STORES its partial
// tile into its slab of `ws` ([nsplit][Cout][KH][KW][Cin] float32, nothing to zero) instead of adding it into dW with float
// atomics.  `nsplit` / `ws_floats` must be what the plan call returned for this geometry (checked).  Step 3 is
// wsmg_weight_grad_reduce_oihw, which adds the slabs in slab order.
extern "C" int wsmg_conv2d_bwd_weight_bf16_slabs(const void* x, const void* dy, float* ws, int nsplit, long long ws_floats, int B, int H,
                                                 int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW,
                                                 wsmg_stream_t stream) {
  if (int e = check_conv(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)) return e;
  const WgradPlan p = wgrad_plan_bf16(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW);
  const long long slab = (long long)Cout * KH * KW * Cin;
  if (!ws || nsplit != p.nsplit || ws_floats < slab * p.nsplit) return WSMG_EINVAL;
  return launch_wgrad_bf16(x, dy, ws, slab, B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, wsmg_s(stream));
}
