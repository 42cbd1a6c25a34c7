// Weight gradient of the 3 x 3, stride-1, pad-1 layers of the map stack out of a ZERO-PADDED LDS PIXEL WINDOW filled by
// LDS-DMA:   dW[co][ky][kx][ci] = sum over pixels p of dY[p][co] * X[p + (ky - 1, kx - 1)][ci].
//
// Why.  The generic kernel (wsmg_conv_bf16.hip, conv_wgrad_bf16_kernel) stages, per (tap, 32-channel) unit, its own
// shifted copy of X: 80 FLOP per byte moved from L2 into LDS, against the ≈140 that the ≈30 B/clk/CU of that path need
// to keep the MFMAs fed — it runs 25-30 % MFMA-busy.  Every input pixel of these layers is used by all nine taps.  Here
// the reduction axis is walked in ZERO-PADDED coordinates: within an image, entry t = y (W + 2) + x' (x' = 0 and W + 1
// are pad columns); a k-step is 48 consecutive entries (three 16-deep MFMA slices; 24 x 26 = 13 k-steps per image), and
//
//   * the dY tile of a k-step is its 48 entries x the workgroup's output channels, pad entries zero-filled by the DMA
//     (out-of-range buffer offsets), so a pad entry contributes nothing;
//   * the X window of a k-step is entries t - (W + 3) .. t + 47 + (W + 3) x the workgroup's input channels, pads and
//     rows outside the image zero-filled — tap (ky, kx) of EVERY entry of the k-step is the same window shifted by
//     (ky - 1)(W + 2) + (kx - 1) rows: nine B fragments per slice from one 13 KB image, with compile-time offsets.
//
// A workgroup (8 waves = WCO output-channel tiles x WCI input-channel tiles of 32) owns 32 WCO x 9 x 32 WCI of dW over a
// range of images; a wave holds its nine 32 x 32 accumulators in registers over the whole range and flushes once with
// float32 atomics.  L2 -> LDS traffic per MFMA is 2.1x lower than the generic kernel's at <4, 2>; the price is the pad
// columns (26 / 24 = 8 % more MFMAs).
//
// Both operands are K(= entry)-major in LDS exactly as they sit in HBM and are fetched with the gfx950 transposing read
// (ds_read_b64_tr_b16), whose 32-lane half takes 4 consecutive rows x 64 B: these must cover the 256-B bank row once.
// LDS-DMA writes rows back to back (no padding), so rows are XOR-swizzled in 64-byte quarters — 256-B rows: quarter ^=
// row & 3; 128-B rows: quarter ^= (row >> 1) & 1 — applied on the source side of the DMA and in the read address (a
// constant per lane and tap: the slice and half offsets are multiples of 4 rows).
//
// Pipeline: NSTG stages, the DMA of k-step s + NSTG - 1 is issued at the top of step s after a counted `vmcnt` + one barrier.
// NSTG = 4 since round 5 (V421: 103 KB, ONE workgroup per CU where three stages let two share it; measured per layer at B = 512,
// interleaved: 0.190-0.197 -> 0.182-0.184 ms (256 <-> 128), 0.358 -> 0.344 (256 -> 256); 5 and 6 stages 0.188-0.194 / 0.356-0.369;
// V222 0.112 -> 0.106 and 0.097 -> 0.093, V412 0.0425 -> 0.042; V118 unchanged at 0.047 and left at three).  The 25
// pieces of a step are issued by waves 0-3 ONLY (one per SIMD, 8 slots each, spare slots to a dummy area): 25 KB per step
// is ≈1 000 cycles of the CU's L2 -> LDS path, and a wave's DMA instructions stall in issue until the path takes them —
// with every wave issuing its share at the top of the step, all eight stood there 1 400-1 700 cycles (s_memtime) while the
// matrix pipe idled.  Now waves 4-7 go straight to their 27 MFMAs and have the pipe to themselves while waves 0-3 load;
// then waves 0-3 multiply.
#include <stdlib.h>

#include "wsmg_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

struct W3wArgs {
  const bf16_t* x;    // [B][H][W][Cin]
  const bf16_t* dy;   // [B][H][W][Cout]
  float* dw;          // [Cout][3][3][Cin] float32, accumulated into (slab == 0) — or the slab workspace [gz][Cout][3][3][Cin]
  int64_t slab;       // > 0: floats per slab — the workgroup of image range z STORES its partial tile into slab z (no atomics,
                      // no zero fill: every element of every slab is written exactly once); wsmg_weight_grad_reduce_oihw adds the slabs
  int B, H, W, Cin, Cout;
  int gco, gci, gz;   // workgroup grid: output-channel tiles, input-channel tiles, image ranges
  int imgs;           // images per range
  unsigned mPW;       // n / (W + 2) == __umulhi(n, mPW)
  unsigned x_bytes, dy_bytes;
};

constexpr int RED_BYTES = 4 * 9 * 16 * 64 * 4;   // the closing reduction of a pixel-split tile: four waves' nine accumulator tiles (144 KB)
constexpr int OOB = (int)0x80000000;

__device__ __forceinline__ int xcd_swizzle_w(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_wave_base, int byte_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)(p));
}
// 64-byte quarter swizzle of a row of `pitch` bytes (see the header)
template <int PITCH>
__device__ __forceinline__ int quarter_xor(int row) {
  static_assert(PITCH == 256 || PITCH == 128 || PITCH == 64, "row pitches of 4, 2 or 1 channel tiles");
  return PITCH == 256 ? (row & 3) : (PITCH == 128 ? ((row >> 1) & 1) : 0);   // 64-byte rows: four consecutive rows ARE one bank row
}

// WCO x WCI: 32-channel tiles of dY / X per workgroup; WK: waves that share one (co, ci) tile pair and split the k-step's 16-entry
// slices between them (small-channel layers, round 4: with Cout x Cin = 32 x 32 there is ONE tile pair — eight waves take one
// eighth of the entries each and their accumulators meet in LDS when the workgroup is done); SL: slices per wave and k-step;
// XCAP: window rows (KS + 2 (W + 3) <= XCAP); NLW: waves that issue the DMA pieces (8: all; 4: waves 0-3, one per SIMD)
template <int WCO, int WCI, int WK, int SL, int XCAP, int NLW, int NSTG = 3>
__global__ __launch_bounds__(512) void conv_win3_wgrad_kernel(W3wArgs a) {
  static_assert(WCO * WCI * WK == 8, "8 waves");
  static_assert(NSTG >= 3 && NSTG <= 6, "stages: the DMA of k-step s + NSTG - 1 is issued at the top of step s");
  constexpr int KS = 16 * WK * SL;                          // entries per k-step
  constexpr int NPAIR = WCO * WCI;
  constexpr int DP = 64 * WCO, XP = 64 * WCI;               // row pitches (bytes)
  constexpr int D_STAGE = KS * DP, X_STAGE = XCAP * XP, STAGE = D_STAGE + X_STAGE;
  constexpr int ND = D_STAGE / 1024, NX = X_STAGE / 1024;   // 1 KB DMA pieces per stage
  static_assert(D_STAGE % 1024 == 0 && X_STAGE % 1024 == 0, "whole DMA pieces");
  constexpr int DRPP = 1024 / DP, XRPP = 1024 / XP;         // rows per piece
  constexpr int DSL = DP / 16, XSL = XP / 16;               // 16-byte slots per row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the DMA role branches below must not be exec-masked
  const int H = a.H, W = a.W, PW = W + 2, HPW = H * PW;
  const int nseg = (HPW + KS - 1) / KS;
  // workgroup -> (tile, image range): the tiles of one range are consecutive logical ids (same XCD: they read the same images)
  const int logical = xcd_swizzle_w(blockIdx.x, gridDim.x);
  const int ntile = a.gco * a.gci;
  const int z = logical / ntile, tl = logical - z * ntile;
  const int co0 = (tl / a.gci) * 32 * WCO, ci0 = (tl % a.gci) * 32 * WCI;
  const int img0 = z * a.imgs;
  const int img1 = img0 + a.imgs < a.B ? img0 + a.imgs : a.B;
  const int steps = (img1 - img0) * nseg;
  if (steps <= 0) return;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dy), 0, (int)a.dy_bytes, 0x00020000);

  // ---- DMA roles.  Issuing a piece with its address arithmetic and role branches redone from scratch took ≈300 cycles of its
  // wave (s_memtime; 60 instructions), four of them at the top of a step 1 300 of the step's 4 300 cycles with the matrix
  // pipe idle: a slot's role is fixed at compile time and its per-lane state is kept and ADVANCED by 48 entries per step,
  // branch-free (a dozen VALU instructions per piece).  What is left, 230-330 cycles per piece, is the L2 -> LDS path
  // itself: 25 KB per step at ≈30 B/clk/CU.  (A piece issued BETWEEN the step's MFMAs and transposing reads costs 600+
  // cycles — 4 700-5 400 cycles per step against 3 300-3 800 with the pieces at the top of the step.)
  constexpr int SD = (ND + NLW - 1) / NLW, SX = (NX + NLW - 1) / NLW, NSLOT = SD + SX;   // slots: dY pieces first, then X pieces
  const bool loader = wave < NLW;
  const int adv_y = KS / PW, adv_x = KS - adv_y * PW;
  // per slot: (y, x') of this lane's row in the k-step being loaded, their values at segment 0, the constant part of the
  // byte offset, and (uniform) the LDS destination within a stage — the dummy area for a slot past the last piece
  int sy[NSLOT], sx[NSLOT], sy0[NSLOT], sx0[NSLOT], cb[NSLOT], dst[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const bool isd = j < SD;
    const int pi = NLW * (isd ? j : j - SD) + wave;
    const bool piece = loader && pi < (isd ? ND : NX);
    if (isd) {
      const int row = DRPP * pi + lane / DSL, slot = lane % DSL;
      const int y = (int)__umulhi((unsigned)row, a.mPW);
      sy0[j] = piece ? y : (1 << 20);                    // rows of no use are never valid
      sx0[j] = row - y * PW;
      cb[j] = co0 * 2 + 16 * (slot ^ (quarter_xor<DP>(row) << 2));
      dst[j] = piece ? pi * 1024 : NSTG * STAGE;
    } else {
      const int row = XRPP * pi + lane / XSL, slot = lane % XSL;
      const int t2 = row + PW - 1;                       // entry - (W + 3) + 2 (W + 2): non-negative
      const int y2 = (int)__umulhi((unsigned)t2, a.mPW);
      sy0[j] = piece && row < KS + 2 * PW + 2 ? y2 - 2 : (1 << 20);
      sx0[j] = t2 - y2 * PW;
      cb[j] = ci0 * 2 + 16 * (slot ^ (quarter_xor<XP>(row) << 2));
      dst[j] = piece ? D_STAGE + pi * 1024 : NSTG * STAGE;
    }
    sy[j] = sy0[j];
    sx[j] = sx0[j];
  }
  int it_img = img0, it_seg = 0;   // the k-step being loaded (uniform)
  auto dma_slot = [&](int j, int stage) {
    if (loader) {
      const bool newimg = it_seg + 1 == nseg;           // the next k-step opens the next image
      {
        const bool ok = it_img < img1 && (unsigned)sy[j] < (unsigned)H && sx[j] >= 1 && sx[j] <= W;
        const int pix = (it_img * H + sy[j]) * W + sx[j] - 1;
        unsigned char* const d = smem + (dst[j] < NSTG * STAGE ? stage * STAGE + dst[j] : NSTG * STAGE);
        if (j < SD) dma16(rs_dy, d, ok ? pix * (a.Cout * 2) + cb[j] : OOB);
        else dma16(rs_x, d, ok ? pix * (a.Cin * 2) + cb[j] : OOB);
        const int nx = sx[j] + adv_x;
        const bool wrap = nx >= PW;
        sx[j] = newimg ? sx0[j] : (wrap ? nx - PW : nx);
        sy[j] = newimg ? sy0[j] : sy[j] + adv_y + (wrap ? 1 : 0);
      }
    }
  };
  auto dma_advance = [&]() { if (++it_seg == nseg) { it_seg = 0; ++it_img; } };
  auto dma_step = [&](int stage) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) dma_slot(j, stage);
    dma_advance();
  };

  // ---- MFMA roles: wave -> (output-channel tile tco, input-channel tile tci); transposing-read lane map as in
  // conv_wgrad_bf16_kernel: the lane supplies row 8 h + q4 (+ 4 for the second read) and channels 16 half16 + 4 p4 .. + 3
  const int pair = wave % NPAIR, wk = wave / NPAIR;
  const int tco = pair % WCO, tci = pair / WCO;
  const int li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const int half16 = (lane >> 4) & 1, h = lane >> 5;
  const int chan_off = (half16 * 16 + p4 * 4) * 2;
  // dY: row 16 c + 8 h + q4 (+ 4): row & 3 == q4 (256-B rows), (row >> 1) & 1 == q4 >> 1 (128-B rows)
  // (wave wk of a tile pair owns slices wk SL .. wk SL + SL - 1 of every k-step: a multiple of 16 rows, which leaves the swizzle alone)
  const int a_off = (16 * SL * wk + 8 * h + q4) * DP + ((tco ^ quarter_xor<DP>(q4)) << 6) + chan_off;
  // X: row 16 c + 8 h + q4 (+ 4) + (W + 3) + shift(tap): the part that is not a multiple of 4 is q4 + (W + 3) + shift
  // (two 16-bit offsets per register: the nine accumulators leave no room for nine address registers)
  unsigned b_pack[5];
#pragma unroll
  for (int tp = 0; tp < 5; ++tp) {
    unsigned v = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int tap = 2 * tp + k < 9 ? 2 * tp + k : 8;
      const int o = q4 + PW + 1 + (tap / 3 - 1) * PW + (tap % 3 - 1);
      v |= (unsigned)(D_STAGE + (16 * SL * wk + 8 * h + o) * XP + ((tci ^ quarter_xor<XP>(o)) << 6) + chan_off) << (16 * k);
    }
    b_pack[tp] = v;
  }
  static_assert(D_STAGE + XCAP * XP < 65536, "packed 16-bit LDS offsets");
  f32x16 acc[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[tap][g] = 0.f;

#pragma unroll
  for (int st = 0; st < NSTG - 1; ++st) dma_step(st);
  for (int s = 0; s < steps; ++s) {
    if (loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * NSLOT) : "memory");   // this wave's pieces of step s (those of the next NSTG - 2 steps may be in flight)
    __builtin_amdgcn_s_barrier();                      // everybody's; and stage (s + NSTG - 1) % NSTG = (s - 1) % NSTG is no longer read
    dma_step((s + NSTG - 1) % NSTG);
    const unsigned char* const sb = smem + (s % NSTG) * STAGE;
    // 27 MFMAs per wave and step, each with its own B fragment (2 transposing reads): read -> use back to back leaves the
    // LDS round trip exposed 27 times per step, so the fragments run PD MFMAs ahead in a small register ring, and the
    // interleave is pinned (1 MFMA, 2 reads, ...): left alone, hipcc groups the reads and waits for all of them.
    constexpr int NI = SL * 9, PD = 4;   // (depth 2 / 4 / 6 measured within 1 % of each other)
    auto rdA = [&](int c) {
      const bf16x4 l = tr_read(sb + a_off + (16 * c) * DP), hh = tr_read(sb + a_off + (16 * c + 4) * DP);
      return __builtin_shufflevector(l, hh, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    const unsigned char* bad[9];   // this step's nine window addresses (unpacked once per step: per use it was 3 VALU per MFMA)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) bad[tap] = sb + (int)((tap & 1) ? b_pack[tap >> 1] >> 16 : b_pack[tap >> 1] & 0xffffu);
    auto rdB = [&](int i) {
      const int c = i / 9, tap = i % 9;
      const bf16x4 l = tr_read(bad[tap] + (16 * c) * XP), hh = tr_read(bad[tap] + (16 * c + 4) * XP);
      return __builtin_shufflevector(l, hh, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 af = rdA(0), afn = af, bq[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) bq[i] = rdB(i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      acc[i % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[i % PD], acc[i % 9], 0, 0, 0);
      if (i + PD < NI) bq[i % PD] = rdB(i + PD);
      if (i % 9 == 4 && i / 9 + 1 < SL) afn = rdA(i / 9 + 1);   // next slice's dY fragment, half a slice ahead
      if (i % 9 == 8) af = afn;
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
      if (i % 9 == 4 && i / 9 + 1 < SL) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- pixel-split tiles: the WK partial sums of a tile pair meet in LDS (the stages are dead), upper half into lower half, in a
  // fixed order (bit-reproducible); the pair's wave 0 then flushes alone
  if constexpr (WK > 1) {
    __syncthreads();
    float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int rr = WK / 2; rr >= 1; rr >>= 1) {
      if (wk >= rr && wk < 2 * rr) {
        float* const dst = red + (size_t)(((wk - rr) * NPAIR + pair) * 144) * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int g = 0; g < 16; ++g) dst[(tap * 16 + g) * 64] = acc[tap][g];
      }
      __syncthreads();
      if (wk < rr) {
        const float* const src = red + (size_t)((wk * NPAIR + pair) * 144) * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int g = 0; g < 16; ++g) acc[tap][g] += src[(tap * 16 + g) * 64];
      }
      __syncthreads();
    }
    if (wk != 0) return;
  }

  // ---- one flush per workgroup: lane r = input channel (consecutive lanes -> 128-byte segments of an OHWI row)
  const int r = lane & 31;
  const int ci = ci0 + 32 * tci + r;
  if (a.slab) {     // deterministic form: plain stores into this image range's slab
    float* const dst = a.dw + (size_t)z * a.slab;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int co = co0 + 32 * tco + (g & 3) + 8 * (g >> 2) + 4 * h;
        dst[((size_t)co * 9 + tap) * a.Cin + ci] = acc[tap][g];
      }
    return;
  }
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = co0 + 32 * tco + (g & 3) + 8 * (g >> 2) + 4 * h;
      atomicAdd(a.dw + ((size_t)co * 9 + tap) * a.Cin + ci, acc[tap][g]);
    }
}

int w3w_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;     // (MI355X; also the answer where no device can be asked — the plan call of a GPU-less build check)
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  return cus;
}

// The tile shapes (WCO, WCI, WK, SL, XCAP):
//   V421: 128 co x 64 ci, 48-entry k-steps — the wide layers (cated 256 -> 256, enc6 128 -> 256, encoded_lin 256 -> 128), W <= 24
//   V222:  64 co x 64 ci, two waves per tile pair, 64-entry k-steps — the decoder's 64-output-channel layers (256 / 192 / 64 -> 64), W <= 24
//   V412: 128 co x 32 ci, two waves per pair, 64-entry k-steps — map_classified_linear (32 -> 128), W <= 24
//   V118:  32 co x 32 ci, eight waves on the one pair, 256-entry k-steps — the classifier's 32 -> 32 layer at 48 x 48, W <= 48
// (round 4: the generic kernel ran these small-channel layers at 240-680 TFLOP/s — it re-fetches X once per tap through the vector
//  memory path, and with 32 channels half of its 64-wide dY tile is padding; out of the window they are bound by the bytes of x and
//  dy, read once.)
enum W3wVariant { V_NONE = 0, V421, V222, V412, V118 };

struct W3wShape { int wco, wci, ks, xcap; };
constexpr W3wShape kShape[5] = {{0, 0, 0, 0}, {4, 2, 48, 104}, {2, 2, 64, 120}, {4, 1, 64, 128}, {1, 1, 256, 368}};

W3wVariant w3w_variant(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H < 1 || W < 16) return V_NONE;   // W < 16: the pad columns and the last k-step's unused entries cost more than
                                                  // the window saves (12 x 12: 0.091 vs 0.052 ms)
  if ((size_t)B * H * W * (size_t)(Cin > Cout ? Cin : Cout) * 2 >= (1ull << 31)) return V_NONE;
  const unsigned PW = (unsigned)(W + 2);
  const uint64_t nmax = (uint64_t)(H + 4) * PW + 256 + 368;   // (largest k-step + window of the shapes above)
  if (nmax * PW >= (1ull << 32)) return V_NONE;
  auto fits = [&](W3wVariant v) { return kShape[v].ks + 2 * (W + 3) <= kShape[v].xcap; };
  if (Cout % 128 == 0 && Cin % 64 == 0) return fits(V421) ? V421 : V_NONE;
  if (Cout % 64 == 0 && Cin % 64 == 0) return fits(V222) ? V222 : V_NONE;
  if (Cout % 128 == 0 && Cin % 32 == 0) return fits(V412) ? V412 : V_NONE;
  if (Cout % 32 == 0 && Cin % 32 == 0) return fits(V118) ? V118 : V_NONE;
  return V_NONE;
}

// image ranges: one workgroup per CU (its LDS), every range the same number of images (no range is empty)
void plan_w3w(W3wArgs& a, W3wVariant v) {
  a.gco = a.Cout / (32 * kShape[v].wco);
  a.gci = a.Cin / (32 * kShape[v].wci);
  const int ntile = a.gco * a.gci;
  int gz = w3w_cus() / ntile;
  if (gz < 1) gz = 1;
  if (gz > a.B) gz = a.B;
  a.imgs = (a.B + gz - 1) / gz;
  a.gz = (a.B + a.imgs - 1) / a.imgs;
}

template <int WCO, int WCI, int WK, int SL, int XCAP, int NLW, int NSTG = 3>
int launch_w3w(W3wArgs& a, W3wVariant v, hipStream_t s) {
  constexpr int STAGES = NSTG * (16 * WK * SL * 64 * WCO + XCAP * 64 * WCI) + 1024;
  constexpr int LDS = (WK > 1 && RED_BYTES > STAGES) ? RED_BYTES : STAGES;
  static_assert(LDS <= 160 * 1024, "LDS of one CU");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_win3_wgrad_kernel<WCO, WCI, WK, SL, XCAP, NLW, NSTG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  plan_w3w(a, v);
  const int ntile = a.gco * a.gci;
  hipLaunchKernelGGL((conv_win3_wgrad_kernel<WCO, WCI, WK, SL, XCAP, NLW, NSTG>), dim3((unsigned)(ntile * a.gz)), dim3(512), LDS, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

// image ranges (= slabs of the deterministic form) this kernel would use for the layer; 0: the layer is not this kernel's
int wsmg_conv_win3_wgrad_splits(int B, int H, int W, int Cin, int Cout) {
  const W3wVariant v = w3w_variant(B, H, W, Cin, Cout);
  if (v == V_NONE) return 0;
  W3wArgs a{nullptr, nullptr, nullptr, 0, B, H, W, Cin, Cout, 0, 0, 0, 0, 0, 0, 0};
  plan_w3w(a, v);
  return a.gz;
}

// dW of a 3 x 3 / stride 1 / pad 1 convolution on bf16 NHWC for the shapes w3w_variant() names; WSMG_EINVAL otherwise (the caller
// then uses the generic kernel).  slab_floats == 0: dW (OHWI float32) is ACCUMULATED INTO with float atomics (the caller zeroes it);
// slab_floats > 0: `dw_ohwi` is a workspace of wsmg_conv_win3_wgrad_splits() slabs of that many floats, every one written whole
// (see W3wArgs::slab).
int wsmg_conv_win3_wgrad_bf16(const void* x, const void* dy, float* dw_ohwi, long long slab_floats, int B, int H, int W, int Cin, int Cout,
                              hipStream_t s) {
  const W3wVariant v = w3w_variant(B, H, W, Cin, Cout);
  if (v == V_NONE) return WSMG_EINVAL;
  const unsigned PW = (unsigned)(W + 2);
  W3wArgs a{(const bf16_t*)x, (const bf16_t*)dy, dw_ohwi, (int64_t)slab_floats, B, H, W, Cin, Cout, 0, 0, 0, 0, (unsigned)((1ull << 32) / PW + 1),
            (unsigned)((size_t)B * H * W * Cin * 2), (unsigned)((size_t)B * H * W * Cout * 2)};
  // V421: waves 0-3 load (measured at B = 512: 0.373 / 0.205 / 0.206 ms on the 256->256, 128->256 and 256->128 layers against 0.404 /
  // 0.216 / 0.216 with all eight loading, and 0.421 / 0.247 / 0.225 for the generic kernel); four LDS stages except for the 32 x 32
  // tiling (three: profiles/r05_win3w_stages_ab.txt — five and six lost, eight loaders lost; those forms are gone)
  switch (v) {
    case V421: return launch_w3w<4, 2, 1, 3, 104, 4, 4>(a, v, s);
    case V222: return launch_w3w<2, 2, 2, 2, 120, 8, 4>(a, v, s);
    case V412: return launch_w3w<4, 1, 2, 2, 128, 8, 4>(a, v, s);
    case V118: return launch_w3w<1, 1, 8, 2, 368, 8>(a, v, s);
    default: return WSMG_EINVAL;
  }
}
