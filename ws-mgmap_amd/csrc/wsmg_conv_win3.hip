// 3 x 3, stride-1, pad-1 convolution (forward and backward-data) for the wide layers of the map stack, out of an
// LDS-RESIDENT, ZERO-PADDED PIXEL WINDOW filled by LDS-DMA.
//
// Why a third structure.  Cycle stamps (s_memtime) in a 256 x 256 LDS-DMA implicit-GEMM tile showed what paces these layers:
// an LDS-DMA wave-instruction (1 KB) costs its wave 100-190 cycles of issue, and an im2col k-step (one tap, 32 channels:
// 16 KB of pixels + 16 KB of weights) is 4 of them per wave — 600 cycles beside the step's 512 MFMA cycles — while the L2 ->
// LDS path (≈30 B/clk/CU) runs at 2/3 of its peak.  The pixels are the waste: every input pixel is fetched 9 times, once per
// tap.  Here a workgroup takes MT CONSECUTIVE output pixels (flattened b, y, x — a tile may cross image rows and images) and
// per 32-channel chunk loads the pixels those touch ONCE, into a window laid out in zero-padded geometry:
//
//     window entry e  <->  padded position q0 + e,   q(b, y, x) = (b (H + 2) + y + 1) (W + 2) + x + 1
//
// so that tap (dy, dx) of every tile row is the same window shifted by dy (W + 2) + dx entries, and taps that fall outside
// the image land on pad entries, which the DMA zero-fills (out-of-range buffer offsets) — no masks, no selects in the k-loop.
// The nine taps of a chunk then read their A fragments from the one window; only the weights (MT-independent, 8 KB per
// k-step at 128 output channels) stream per tap.  Memory -> LDS bytes per MFMA drop 2.4x against the im2col tile at MT = 512
// and the DMA instructions per wave and k-step from 4 to 1.5.
//
//   * 8 waves as 4 (pixels) x 2 (channels); wave tile MT/4 x 64 of v_mfma_f32_32x32x16_bf16 accumulators;
//   * tile rows are 64 B (32 channels); 16-byte chunk c of entry / row i sits at slot c ^ ((i >> 2) & 3): every
//     ds_read_b128 lane group then covers all 16 slots of a 256-B bank row, for any tap shift (the XOR is applied on the
//     SOURCE side of the DMA, whose LDS destination is lane-linear);
//   * k-loop: channel chunk outer, tap inner (unrolled); weights in 4 stages, 3 k-steps ahead; the next chunk's window
//     (double buffered) is issued one piece per wave over the first taps of the current chunk; ONE barrier per k-step after a
//     counted `vmcnt` (the count depends on the tap: taps that carry a window piece issue 2 DMAs, the others 1); the first
//     slice's fragments of a step are read during the step before, so no MFMA waits on LDS right after a barrier;
//   * backward-data of a stride-1 3 x 3 convolution is the same sum with the shifts negated and the weights as IHWO.
//
// Epilogue through LDS in two 64-column halves: bias / ReLU / BatchNorm sums as in conv_igemm_bf16_kernel.
//
// Where a tile's cycles go (s_memtime, cated layer 256 -> 256 at B = 512, MT = 512): address arithmetic + first window 13 %,
// k-loop 79 % (1 294 cycles per k-step against 1 024 of MFMA), epilogue 8 %; whole kernel 54 % MFMA-busy at 1.9 GHz
// (the clock gives back as the MFMA density rises) = 1.15 PFLOP/s against 0.95 for the implicit-GEMM kernel.  Tried on this
// structure and dropped (each measured in one run against the form above): waves 4-7 issuing their DMAs between the two
// slices (slower: the wave-uniform branch splits the step's scheduling region); accumulators transposed for 8-byte
// epilogue stores, through LDS (no gain) and straight to memory (2.5x slower: a store costs the lines it touches, 32 per
// instruction in that form); two-byte stores straight from the accumulators (equal); a persistent form, one workgroup per
// CU walking its tiles with the next tile's first window prefetched under the last chunk (equal alone — every CU then
// writes its 128 KB tile at the same moment — and 0.2-0.3 ms per update SLOWER in the policy, where its 256 whole-CU
// workgroups queue behind whatever else is resident).
#include <stdlib.h>

#include <type_traits>

#include "wsmg_common.h"
#include "wsmg_relu_mask.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Win3Args {
  const bf16_t* src;  // [B][H][W][Kc]
  const bf16_t* wt;   // [N][3][3][Kc]
  const float* bias;  // [N] or null
  bf16_t* dst;        // [B][H][W][N]
  int B, H, W, Kc, N;
  int mtiles, ntiles;
  int relu, bwd;
  unsigned src_bytes, wt_bytes;
  double* stats;
  int nslab;
  // round 5 — two tile sizes in ONE launch: blocks [0, nbig) take MT-pixel tiles from pixel 0, blocks [nbig, nbig + nsmall) take
  // MT/2-pixel tiles from pixel m_split (0 / 0 / 0: every block is a big tile)
  int nbig, nsmall, m_split;
  // round 6: pixel pitch of dst in elements (0 = N: larger writes a channel slice of a wider tensor in place)
  int dst_ld;
  const bf16_t* relu_z;   // [pixels][N] or null: the ReLU outputs the gradient tile is masked with before it is stored (wsmg_relu_mask.h)
  // output channels >= split_c (a multiple of the channel tile) go to dst2 [pixels][N - split_c] instead, channels below it to dst
  // [pixels][split_c]: the gradient of a two-part concatenation leaves as its two parts (dst2 null: one tensor)
  bf16_t* dst2;
  int split_c;
};

constexpr int DK = 32, ROWB = 64, NST = 4;
constexpr int OPITCH = 64 * 2 + 16;          // epilogue staging: 64 columns of bf16 + 16

__device__ __forceinline__ int xcd_swizzle3(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}
__device__ __forceinline__ unsigned short f2bf3(float f) {
  bf16_t b = (bf16_t)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_wave_base, int byte_off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
// DMA instructions a wave issues in the k-step at tap `tap`: a window piece of the next chunk (taps 0 .. npw - 1) and the weights of 3 steps on
constexpr int n_issue(int tap, int npw) { return ((tap % 9 + 9) % 9) < npw ? 2 : 1; }

// MT: pixels per workgroup (512 or 256); NPW: window pieces (16 entries each) per wave — window capacity 128 NPW entries;
// NT: output channels per workgroup — 128 (8 waves as 4 along the pixels x 2 along the channels) or, round 3, 64 for the
// 64-channel layers (8 x 1: every wave takes MT / 8 pixels x all 64 channels; waves 4-7 have no weight rows to fetch and issue
// their weight DMA as a zero-fill into a dummy KB, so that every wave's `vmcnt` counts the same instructions)
// AUX: 0 = the plain store loop (the forward passes, plain gradients: the round-5 kernel, register for register); 1 = the gradient
// tile is masked with relu_z and / or stored as two tensors (dst2 / split_c).
template <int MT, int NPW, int NT, int AUX>
__device__ __forceinline__ void win3_tile_body(const Win3Args& a, const int bid, const int nblocks, const int m_base, const int blk_base) {
  constexpr int WN = NT >= 128 ? 2 : 1, WM = 8 / WN;   // waves along the channels / along the pixels
  constexpr int NU = NT >= 64 ? 2 : 1;         // 32-column accumulator tiles per wave (NT = 32: one)
  constexpr int TT = MT / (32 * WM);           // 32-row accumulator tiles per wave along the pixels
  constexpr int BSTAGE = NT * ROWB;            // 8 KB (4 KB) of weights per k-step
  constexpr int WCAP = 128 * NPW;              // window entries
  constexpr int WINB = WCAP * ROWB;
  constexpr int OFF_B = 2 * WINB;
  constexpr int OFF_DUMMY = OFF_B + NST * BSTAGE;
  static_assert(NPW <= 6, "window pieces ride on taps 0 .. NPW - 1");
  static_assert(MT * OPITCH <= OFF_DUMMY, "epilogue staging fits under the k-loop's LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int logical = xcd_swizzle3(bid, nblocks);
  const int m0 = m_base + (logical / a.ntiles) * MT, n0 = (logical % a.ntiles) * NT;
  const int H = a.H, W = a.W, PW = W + 2, PP = (H + 2) * PW;
  const int Mtot = a.B * H * W;

  auto padded = [&](int m) {   // flattened pixel -> padded position
    const int b = m / (H * W), rr = m - b * (H * W), y = rr / W, x = rr - y * W;
    return b * PP + (y + 1) * PW + x + 1;
  };
  const int mlast = (m0 + MT - 1 < Mtot ? m0 + MT - 1 : Mtot - 1);
  const int q0 = padded(m0) - PW - 1;                       // window entry 0 (the first pixel's upper-left neighbour)
  const int nwin = padded(mlast) + PW + 1 - q0 + 1;         // entries in use (host checked nwin <= WCAP for every tile)

  // ---- DMA roles.  Window piece j of this wave = entries 16 (8 j + wave) .. + 15; lane = (entry in the piece, slot).
  const int lrow = lane >> 2, slot = lane & 3;
  int aoff[NPW];   // byte offset of (source pixel, channel 0, this lane's swizzled chunk), or out of range (pad / past the window)
  {
    // (image, padded row, padded column) of the first piece's entry by division, of the others by stepping 128 entries on (round 6: two
    // runtime-divisor divisions per piece and two per accumulator tile below were ~20 per lane and tile — 3 000 cycles of a tile that has
    // one workgroup per CU and nothing to hide them under; q0 >= 0: the window starts on the pad row above the tile's first pixel)
    const int q = q0 + 16 * wave + lrow;
    int b = q / PP;
    const int rem = q - b * PP;
    int pr = rem / PW, pc = rem - pr * PW;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int e = 16 * (8 * j + wave) + lrow;
      const int yy = pr - 1, xx = pc - 1;
      const bool ok = e < nwin && b < a.B && yy >= 0 && yy < H && xx >= 0 && xx < W;
      aoff[j] = ok ? (((b * H + yy) * W + xx) * a.Kc) * 2 + 16 * (slot ^ ((e >> 2) & 3)) : (int)0x80000000;
      pc += 128;
      while (pc >= PW) { pc -= PW; ++pr; }
      while (pr >= H + 2) { pr -= H + 2; ++b; }
    }
  }
  // weights: piece `wave` of a k-step = rows 16 wave .. + 15 of the [128][32] tile
  const int brow = 16 * wave + lrow;
  const int boff = (n0 + brow) * 9 * a.Kc * 2 + 16 * (slot ^ ((brow >> 2) & 3));
  const __amdgpu_buffer_rsrc_t rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.src), 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wt), 0, (int)a.wt_bytes, 0x00020000);
  const int nchunks = a.Kc / DK;

  auto dma_window_piece = [&](int j, int chunk) {   // piece j of chunk's window -> buffer chunk & 1 (past the last chunk: zeros)
    unsigned char* const dst = smem + (chunk & 1) * WINB + (8 * j + wave) * 1024;
    dma16(rs_src, dst, chunk < nchunks ? aoff[j] + chunk * (DK * 2) : (int)0x80000000);
  };
  auto dma_weights = [&](int chunk, int tap) {       // k-step (chunk, tap) -> stage (9 chunk + tap) % NST
    const int step = 9 * chunk + tap;
    const bool mine = 16 * wave < NT;                // (NT = 64: the tile has 64 weight rows, pieces 0-3)
    unsigned char* const dst = mine ? smem + OFF_B + (step % NST) * BSTAGE + wave * 1024 : smem + OFF_DUMMY;
    dma16(rs_wt, dst, (mine && chunk < nchunks) ? boff + (tap * a.Kc + chunk * DK) * 2 : (int)0x80000000);
  };

  // ---- MFMA roles: wave tile (MT / WM) x 64 at (wm, wn)
  const int wm = (wave % WM) * (MT / WM), wn = (wave / WM) * 64;
  const int r = lane & 31, h = lane >> 5;
  int ecen[TT];    // window entry of this lane's row of accumulator tile t, centre tap
  {
    const int m = m0 + wm + r;
    int b = m / (H * W);
    const int rr = m - b * (H * W);
    int y = rr / W, x = rr - y * W;
    const int elast = padded(Mtot - 1) - q0;                 // rows past the last pixel repeat it (never stored)
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      ecen[t] = m + 32 * t < Mtot ? b * PP + (y + 1) * PW + x + 1 - q0 : elast;
      x += 32;
      while (x >= W) { x -= W; ++y; }
      while (y >= H) { y -= H; ++b; }
    }
  }
  int bpos[NU][2];  // byte offset of this lane's weight fragment (column tile u, 16-deep slice ks) in a stage
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int n = wn + 32 * u + r;
      bpos[u][ks] = n * ROWB + (((2 * ks + h) ^ ((n >> 2) & 3)) << 4);
    }
  f32x16 acc[TT][NU];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  // ---- prologue: the whole window of chunk 0, then the weights of k-steps 0, 1, 2
#pragma unroll
  for (int j = 0; j < NPW; ++j) dma_window_piece(j, 0);
  dma_weights(0, 0);
  dma_weights(0, 1);
  dma_weights(0, 2);

  const int sgn = a.bwd ? -1 : 1;
  // Fragments of a k-step's first 16-deep slice are read one step EARLY (during the previous step's MFMAs), so the MFMAs
  // that follow a barrier never wait on LDS: the wait at the top of step s therefore covers the weights of step s + 1.
  bf16x8 af0[TT], bf0[NU];
  auto read_slice = [&](bf16x8 (&af)[TT], bf16x8 (&bfr)[NU], const unsigned char* win, const unsigned char* bst, int shift, int ks) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      const int e = ecen[t] + shift;
      af[t] = *reinterpret_cast<const bf16x8*>(win + e * ROWB + (((2 * ks + h) ^ ((e >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) bfr[u] = *reinterpret_cast<const bf16x8*>(bst + bpos[u][ks]);
  };
  auto mfma_slice = [&](const bf16x8 (&af)[TT], const bf16x8 (&bfr)[NU]) {
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int u = 0; u < NU; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], bfr[u], acc[t][u], 0, 0, 0);
  };
  auto tap_shift = [&](int tap) {
    int shift = sgn * ((tap / 3 - 1) * PW + (tap % 3 - 1));
    asm volatile("" : "+s"(shift));   // keep the 9 taps' fragment addresses from being hoisted out of the chunk loop (they spill)
    return shift;
  };
  auto kstep = [&](auto tapc, int chunk) {
    constexpr int tap = decltype(tapc)::value;
    constexpr int ntap = (tap + 1) % 9;
    const unsigned char* const win = smem + (chunk & 1) * WINB;
    const unsigned char* const nwinp = smem + ((tap == 8 ? chunk + 1 : chunk) & 1) * WINB;
    const int step = 9 * chunk + tap;
    // this wave's pieces of everything up to the weights of step + 1 have landed; then everybody's have, and nobody reads
    // the stage / window buffer about to be overwritten any more
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_issue(tap + 8, NPW)) : "memory");
    __builtin_amdgcn_s_barrier();
    // (issuing the DMAs of waves 4-7 between the two slices instead, so that one wave of each SIMD always has MFMAs ready,
    // measured slower: the wave-uniform branch splits the step into blocks the scheduler cannot interleave across)
    if constexpr (tap < NPW) dma_window_piece(tap, chunk + 1);
    dma_weights(chunk + (tap + 3) / 9, (tap + 3) % 9);
    bf16x8 af1[TT], bf1[NU];
    read_slice(af1, bf1, win, smem + OFF_B + (step % NST) * BSTAGE, tap_shift(tap), 1);
    mfma_slice(af0, bf0);
    read_slice(af0, bf0, nwinp, smem + OFF_B + ((step + 1) % NST) * BSTAGE, tap_shift(ntap), 0);
    mfma_slice(af1, bf1);
  };
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   // the window of chunk 0 and the weights of step 0 (steps 1, 2 may be in flight)
  __syncthreads();
  read_slice(af0, bf0, smem, smem + OFF_B, tap_shift(0), 0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    kstep(std::integral_constant<int, 0>{}, chunk);
    kstep(std::integral_constant<int, 1>{}, chunk);
    kstep(std::integral_constant<int, 2>{}, chunk);
    kstep(std::integral_constant<int, 3>{}, chunk);
    kstep(std::integral_constant<int, 4>{}, chunk);
    kstep(std::integral_constant<int, 5>{}, chunk);
    kstep(std::integral_constant<int, 6>{}, chunk);
    kstep(std::integral_constant<int, 7>{}, chunk);
    kstep(std::integral_constant<int, 8>{}, chunk);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-fill DMAs past the last k-step must not land in the staging below
  __syncthreads();

  // ---- epilogue: two passes (column tile u of every wave: 32 WN staged columns) through LDS, then 16-byte stores
  constexpr int PCS = 4 * WN;                               // 16-byte pieces per staged row
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int scol = (wave / WM) * 32 + r;                  // column in the staged half
    const float bv = a.bias ? a.bias[n0 + wn + 32 * u + r] : 0.f;
    float sv = 0.f, qv = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = wm + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
        float v = acc[t][u][g] + bv;
        if (a.relu) v = v > 0.f ? v : 0.f;
        const unsigned short o = f2bf3(v);
        *reinterpret_cast<unsigned short*>(smem + row * OPITCH + scol * 2) = o;
        if (a.stats) {   // train-mode BatchNorm sums of the ROUNDED values, from the registers (see conv_igemm_bf16_kernel)
          const float vr = m0 + row < Mtot ? __uint_as_float((unsigned)o << 16) : 0.f;
          sv += vr;
          qv = fmaf(vr, vr, qv);
        }
      }
    if (a.stats) {
      sv += __shfl_xor(sv, 32, 64);
      qv += __shfl_xor(qv, 32, 64);
      if (h == 0) {
        double* st = a.stats + (size_t)((logical / a.ntiles) % a.nslab) * 2 * a.N + n0 + wn + 32 * u + r;
        atomicAdd(st, (double)sv);
        atomicAdd(st + a.N, (double)qv);
      }
    }
    __syncthreads();
    const int ldd = a.dst_ld ? a.dst_ld : a.N;
    if constexpr (AUX == 0) {
#pragma unroll
      for (int j = 0; j < MT * PCS / 512; ++j) {
        const int c = tid + 512 * j;
        const int row = c / PCS, ch = c % PCS;                // 16-byte piece ch of the staged row: staged columns 8 ch .. 8 ch + 7
        const int n = n0 + (ch >> 2) * 64 + 32 * u + (ch & 3) * 8;
        if (m0 + row >= Mtot) continue;
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.dst) + ((size_t)(m0 + row) * ldd + n) * 2) =
            *reinterpret_cast<const u32x4*>(smem + row * OPITCH + ch * 16);
      }
      __syncthreads();
    } else {
      // the pieces of relu_z are requested in groups of (up to) four rows, ahead of those rows' stores: inside a plain store loop every
      // load would wait behind the previous row's store (the compiler cannot prove z and dst apart)
      constexpr int ROWS = MT * PCS / 512;
      constexpr int GR = ROWS < 4 ? ROWS : 4;
#pragma unroll
      for (int j0 = 0; j0 < ROWS; j0 += GR) {
        u32x4 zr[GR];
        if (a.relu_z) {
#pragma unroll
          for (int jj = 0; jj < GR; ++jj) {
            const int c = tid + 512 * (j0 + jj);
            const int row = c / PCS, ch = c % PCS;
            const int n = n0 + (ch >> 2) * 64 + 32 * u + (ch & 3) * 8;
            zr[jj] = *reinterpret_cast<const u32x4*>(a.relu_z + (size_t)(m0 + row < Mtot ? m0 + row : m0) * a.N + n);
          }
        }
#pragma unroll
        for (int jj = 0; jj < GR; ++jj) {
          const int c = tid + 512 * (j0 + jj);
          const int row = c / PCS, ch = c % PCS;                // 16-byte piece ch of the staged row: staged columns 8 ch .. 8 ch + 7
          const int n = n0 + (ch >> 2) * 64 + 32 * u + (ch & 3) * 8;
          if (m0 + row >= Mtot) continue;
          u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * OPITCH + ch * 16);
          if (a.relu_z) v = relu_mask8(v, zr[jj]);
          if (a.dst2) {
            const bool hi = n >= a.split_c;
            unsigned char* const base = reinterpret_cast<unsigned char*>(hi ? a.dst2 : a.dst);
            *reinterpret_cast<u32x4*>(base + ((size_t)(m0 + row) * (hi ? a.N - a.split_c : a.split_c) + (hi ? n - a.split_c : n)) * 2) = v;
          } else {
            *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(a.dst) + ((size_t)(m0 + row) * ldd + n) * 2) = v;
          }
        }
      }
      __syncthreads();
    }
  }
}

template <int MT, int NPW, int NT, int AUX>
__global__ __launch_bounds__(512) void conv_win3_kernel(Win3Args a) {
  win3_tile_body<MT, NPW, NT, AUX>(a, blockIdx.x, gridDim.x, 0, 0);
}

// Two tile sizes in one launch (round 5).  The wide layers launch 1 152 workgroups of one CU each on 256 CUs: 4.5 rounds, i.e. the
// last round runs on half the chip for a whole tile's time (5 rounds' time for 4.5 rounds' work: 10 %; the 64-channel layers, 576
// workgroups, pay 3 rounds for 2.25).  Here the whole rounds keep the big tile and the remainder is cut into tiles of half the
// pixels — twice the workgroups, each done in about half the time — dispatched last (block order).  Same arithmetic per output
// element (a tile's size changes which workgroup computes a pixel, not how).
template <int MT, int NPW, int NPW_S, int NT, int AUX>
__global__ __launch_bounds__(512) void conv_win3_mixed_kernel(Win3Args a) {
  if ((int)blockIdx.x < a.nbig) win3_tile_body<MT, NPW, NT, AUX>(a, blockIdx.x, a.nbig, 0, 0);
  else win3_tile_body<MT / 2, NPW_S, NT, AUX>(a, (int)blockIdx.x - a.nbig, a.nsmall, a.m_split, a.nbig / a.ntiles);
}

template <int MT, int NPW, int NT, int AUX>
int launch_win3_aux(Win3Args& a, hipStream_t s) {
  constexpr int LDS = 2 * 128 * NPW * ROWB + NST * NT * ROWB + 1024;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_win3_kernel<MT, NPW, NT, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  const int64_t M = (int64_t)a.B * a.H * a.W;
  a.mtiles = (int)wsmg_cdiv(M, MT);
  a.ntiles = a.N / NT;
  hipLaunchKernelGGL((conv_win3_kernel<MT, NPW, NT, AUX>), dim3((unsigned)(a.mtiles * a.ntiles)), dim3(512), LDS, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
// which store loop a launch needs: 1 = mask / split output
inline int aux_form(const Win3Args& a) { return (a.relu_z || a.dst2) ? 1 : 0; }
template <int MT, int NPW, int NT>
int launch_win3(Win3Args& a, hipStream_t s) {
  return aux_form(a) ? launch_win3_aux<MT, NPW, NT, 1>(a, s) : launch_win3_aux<MT, NPW, NT, 0>(a, s);
}

int g_cus3 = 0;
int win3_cus() {
  if (!g_cus3) {
    int dev = 0;
    hipDeviceProp_t prop;
    g_cus3 = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      g_cus3 = prop.multiProcessorCount;
  }
  return g_cus3;
}

// Split of a layer's M pixels x (N / NT) channel tiles into whole rounds of MT-pixel tiles and a remainder of MT/2-pixel tiles; false
// when the plain launch is as good (whole rounds already, fewer than one round, or a remainder that would still need two rounds)
bool mixed_plan(Win3Args& a, int64_t M, int MT, int NT) {
  const int cus = win3_cus();
  const int ntiles = a.N / NT;
  const int64_t big = wsmg_cdiv(M, MT) * ntiles;
  const int64_t full = big / cus * cus;
  const int64_t rem = big - full;
  if (full == 0 || rem == 0 || 2 * rem > cus || full % ntiles) return false;
  const int64_t m_split = full / ntiles * MT;               // pixels covered by the whole rounds
  if (m_split >= M) return false;
  a.nbig = (int)full;
  a.m_split = (int)m_split;
  a.nsmall = (int)(wsmg_cdiv(M - m_split, MT / 2) * ntiles);
  return true;
}

template <int MT, int NPW, int NPW_S, int NT, int AUX>
int launch_win3_mixed_aux(Win3Args& a, hipStream_t s) {
  constexpr int LDS = 2 * 128 * NPW * ROWB + NST * NT * ROWB + 1024;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_win3_mixed_kernel<MT, NPW, NPW_S, NT, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  a.ntiles = a.N / NT;
  a.mtiles = 0;
  hipLaunchKernelGGL((conv_win3_mixed_kernel<MT, NPW, NPW_S, NT, AUX>), dim3((unsigned)(a.nbig + a.nsmall)), dim3(512), LDS, s, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
template <int MT, int NPW, int NPW_S, int NT>
int launch_win3_mixed(Win3Args& a, hipStream_t s) {
  return aux_form(a) ? launch_win3_mixed_aux<MT, NPW, NPW_S, NT, 1>(a, s) : launch_win3_mixed_aux<MT, NPW, NPW_S, NT, 0>(a, s);
}

// window entries a tile of mt consecutive pixels can need: the pixels, 2 pads per image row crossed, the pad rows between
// images, and one padded row + 1 entry of halo on either side
int window_bound(int mt, int H, int W) {
  const int rows = (mt + W - 2) / W + 1;             // image rows a run of mt pixels can touch
  const int imgs = (mt + H * W - 2) / (H * W) + 1;   // images
  return mt + 2 * (rows - 1) + 2 * (W + 2) * (imgs - 1) + 2 * (W + 3) + 1;
}

}  // namespace

// 3 x 3 / stride 1 / pad 1, bf16 in / bf16 out, N % 64 == 0 (128-channel tiles when N % 128 == 0), Kc % 32 == 0; WSMG_EINVAL otherwise (the caller then uses the
// implicit-GEMM kernel).  bwd = 0: forward (src = x, wt = OHWI); 1: backward-data (src = dy, wt = IHWO).
int wsmg_conv_win3_bf16(int bwd, const void* src, const void* wt, const float* bias, void* dst, int relu, double* stats, int nslab,
                        int B, int H, int W, int Kc, int N, int mt, int mixed, const void* relu_z, int dst_ld, void* dst2, int split_c,
                        hipStream_t s) {
  if (N <= 0 || N % 32 || Kc % DK || B <= 0) return WSMG_EINVAL;
  if (dst2 && (dst_ld || split_c <= 0 || split_c >= N || (split_c & 7) || ((uintptr_t)dst2 & 15))) return WSMG_EINVAL;
  if ((int64_t)B * (H + 2) * (W + 2) * 1 > (1 << 30) || (int64_t)B * H * W * (Kc > N ? Kc : N) * 2 >= (1ll << 31)) return WSMG_EINVAL;
  if ((dst_ld && (dst_ld < N || (dst_ld & 7))) || ((uintptr_t)relu_z & 15)) return WSMG_EINVAL;
  Win3Args a{(const bf16_t*)src, (const bf16_t*)wt, bias, (bf16_t*)dst, B, H, W, Kc, N, 0, 0, relu, bwd,
             (unsigned)((size_t)B * H * W * Kc * 2), (unsigned)((size_t)N * 9 * Kc * 2), stats, nslab, 0, 0, 0, dst_ld, (const bf16_t*)relu_z, (bf16_t*)dst2, split_c};
  const int64_t M = (int64_t)B * H * W;
  // mixed == 0 (a tile size forced through wsmg_conv_debug_win3_tile) or WSMG_CONV_WIN3_MIXED=0: one tile size per launch (rounds 2-4; A/B)
  const bool mix = mixed && WSMG_TUNE("WSMG_CONV_WIN3_MIXED", 1) != 0;
  if (N % 128 == 0) {
    if (mt == 512 && window_bound(512, H, W) <= 128 * 6) {
      if (mix && window_bound(256, H, W) <= 128 * 4 && mixed_plan(a, M, 512, 128)) return launch_win3_mixed<512, 6, 4, 128>(a, s);
      return launch_win3<512, 6, 128>(a, s);
    }
    if (mt == 256 && window_bound(256, H, W) <= 128 * 4) {
      if (mix && window_bound(128, H, W) <= 128 * 2 && mixed_plan(a, M, 256, 128)) return launch_win3_mixed<256, 4, 2, 128>(a, s);
      return launch_win3<256, 4, 128>(a, s);
    }
    return WSMG_EINVAL;
  }
  if (N % 64 == 0) {   // 64-channel tiles (N = 64, 192, ...)
    if (mt == 512 && window_bound(512, H, W) <= 128 * 6) {
      if (mix && window_bound(256, H, W) <= 128 * 4 && mixed_plan(a, M, 512, 64)) return launch_win3_mixed<512, 6, 4, 64>(a, s);
      return launch_win3<512, 6, 64>(a, s);
    }
    if (mt == 256 && window_bound(256, H, W) <= 128 * 4) return launch_win3<256, 4, 64>(a, s);
    return WSMG_EINVAL;
  }
  // 32-channel tiles (N = 32, 96, ...)
  if (mt == 512 && window_bound(512, H, W) <= 128 * 6) return launch_win3<512, 6, 32>(a, s);
  if (mt == 256 && window_bound(256, H, W) <= 128 * 4) return launch_win3<256, 4, 32>(a, s);
  return WSMG_EINVAL;
}
