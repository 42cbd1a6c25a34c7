// Weight gradient of the map stack's other two stride-2 layers out of an LDS-RESIDENT INPUT WINDOW (round 4):
//
//   enc3  MapEncoder.cnn[3]        64 -> 128 channels, k5 s2 p1, 50 x 50 -> 24 x 24   (map_encoder.py:23 of the reference)
//   stem  MapDecoder.layer0 conv1  256 -> 64 channels, k7 s2 p3, 24 x 24 -> 12 x 12   (map_encoder.py:76)
//
//   dW[co][ky][kx][ci] = sum over output pixels p = (b, oy, ox) of dY[p][co] * X[b][2 oy - P + ky][2 ox - P + kx][ci]
//
// Why.  The generic kernel (wsmg_conv_bf16.hip, conv_wgrad_bf16_kernel) stages, for every (tap, 32-channel) unit, its own shifted
// copy of X through the vector-memory path: 25 / 49 taps x the input, 640-680 TFLOP/s at 0.22 MFMA-busy, and 120 / 67 MB of slabs
// for the ordered reduce.  This is the design of the k8 stem's kernel (wsmg_conv_win_wgrad.hip) with its constants turned into
// template parameters: a workgroup takes ONE kernel row ky and one 64-channel chunk of the input (its ROLE) and walks tiles of
// TH x TW output pixels; per tile it loads the TH input rows that ky touches (all columns the K taps kx need: column-parity planes, so
// that consecutive output pixels of a tap are consecutive entries) and the tile's dY once, and all K taps kx read their X^T
// fragments out of that window.  The K x (co tile) x (ci half) accumulators of a wave stay in registers over ALL the workgroup's
// tiles (persistent: roles x groups of tiles): one flush per workgroup, one slab per group.
//
// Differences from the k8 kernel: (1) 64-byte-pitch PLANES instead of padded 192-byte entries — X as [ci half][row][parity][column]
// x 64 B and dY as [co tile][pixel] x 64 B: four consecutive entries are one 256-byte bank row, which is what a 32-lane half of
// the transposing read (ds_read_b64_tr_b16) takes, with no padding (the 192-byte pitch cost a third of the window's LDS);
// (2) wave -> (co tile, ci half), all K taps: one dY fragment and K X fragments per 16-pixel chunk (reads per MFMA 1 + 1 / K);
// (3) 8 waves where the layer has four co tiles (enc3).
#include <stdlib.h>

#include "wsmg_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct S2wArgs {
  const bf16_t* x;    // [B][H][W][Cin]
  const bf16_t* dy;   // [B][OH][OW][Cout]
  float* dw;          // [Cout][K][K][Cin] float32 (OHWI), accumulated into (slab == 0) — or the slab workspace [groups][Cout][K][K][Cin]
  int64_t slab;       // > 0: floats per slab — group g STORES its roles' partial sums into slab g (its roles cover dW once)
  int B, H, W, OH, OW, Cin, tiles_y, groups;
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

// K, P: kernel size and padding (stride 2); TH x TW: output pixels of a tile (TW = OW: whole rows); NCO: 32-channel tiles of Cout;
// NW: waves (NW = 2 NCO: wave -> (co tile, ci half)); NCC: 64-channel chunks of Cin (roles = K x NCC)
template <int K, int P, int TH, int TW, int NCO, int NW, int NCC>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void conv_s2_wgrad_kernel(S2wArgs a) {
  static_assert(NW == 2 * NCO, "wave -> (co tile, ci half)");
  constexpr int NT = NW * 64;
  constexpr int S = 2;
  constexpr int WC = ((TW - 1) * S + K + 1) & ~1;   // window columns (even)
  constexpr int PC = WC / S;                        // entries per column-parity plane
  constexpr int XROW = S * PC * 64;                 // one window row of one ci half (bytes)
  constexpr int XPLANE = TH * XROW + 64;            // one ci half (+ 64: the two halves' stores of a pixel land 64 B apart mod 256)
  constexpr int NPXR = TH * TW;                     // output pixels of a tile
  constexpr int NPX = (NPXR + 15) & ~15;            // K axis of the MFMAs
  constexpr int NCH = NPX / 16;
  constexpr int DPLANE = NPX * 64 + 64;             // one co tile of dY
  constexpr int X_BYTES = 2 * XPLANE;               // (+ NCO * DPLANE of dY behind it)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const xs = lds;
  unsigned char* const ds = lds + X_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup -> (role, group): the roles of a group read the same dY tiles and overlapping input rows at about the same time, so they
  // share an XCD (its L2): local index i = bid / 8 on XCD bid % 8 -> role i % R, group (bid % 8) + 8 (i / R)
  constexpr int R = K * NCC;
  const int xcd = blockIdx.x & 7, li_ = blockIdx.x >> 3;
  const int role = li_ % R, grp = xcd + 8 * (li_ / R);
  const int ky = role % K, cc = role / K;
  const int ntiles = a.B * a.tiles_y;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x, a.x_bytes), dr = make_rsrc(a.dy, a.dy_bytes);

  // ---- staging.  Address arithmetic per piece is what paces a tile otherwise (a piece = 16 bytes; 13-17 pieces per thread and tile
  // against 60 MFMAs: with a division-based piece -> (row, column) map per piece the VALU work beside a tile's MFMAs was as long
  // as the MFMAs).  So the pieces are dealt by ROW: thread t owns piece (column t >> 3, chunk t & 7) of EVERY window row — its
  // column validity, its offset within an input row and its LDS offset within a window row are per-thread constants and a row
  // costs one scalar base; the tile's dY is one contiguous run of memory (TW = OW, TH | OH): piece c of it is base + 16 c.
  static_assert(WC * 8 <= NT, "one window row per pass");
  static_assert(NPX == NPXR, "tiles are whole chunks of 16 pixels");
  constexpr int XPER = TH;
  constexpr int DPIECES = NPX * NCO * 4;
  constexpr int DPER = (DPIECES + NT - 1) / NT;
  constexpr int COUT = 32 * NCO;
  const int xwc = tid >> 3, xch = tid & 7;
  const bool colok = tid < WC * 8 && (unsigned)(xwc - P) < (unsigned)a.W;
  const int xcol = ((xwc - P) * a.Cin + cc * 64) * 2 + xch * 16;                                      // within an input row
  const int xlds = (xch >> 2) * XPLANE + ((xwc & 1) * PC + (xwc >> 1)) * 64 + (xch & 3) * 16;         // within a window row
  u32x4 rx[XPER], rd[DPER];
  auto gload = [&](int tile) {
    const int b = tile / a.tiles_y, t = tile - b * a.tiles_y;
    const int oy0 = t * TH;
    const int iyb = oy0 * S - P + ky;
    const bool live = tile < ntiles;
    const int rowbytes = a.W * a.Cin * 2;
#pragma unroll
    for (int j = 0; j < XPER; ++j) {
      const int iy = iyb + j * S;                                    // (uniform)
      const bool rowok = live && (unsigned)iy < (unsigned)a.H;
      const int base = (b * a.H + iy) * rowbytes;
      rx[j] = buf_load16(xr, (rowok && colok) ? base + xcol : (int)0x80000000);
    }
    const int dbase = ((b * a.OH + oy0) * a.OW) * COUT * 2;
#pragma unroll
    for (int j = 0; j < DPER; ++j) {
      const int c = tid + NT * j;
      rd[j] = buf_load16(dr, (live && c < DPIECES) ? dbase + c * 16 : (int)0x80000000);
    }
  };
  auto lstore = [&]() {
    if (tid < WC * 8) {
#pragma unroll
      for (int j = 0; j < XPER; ++j) *reinterpret_cast<u32x4*>(xs + xlds + j * XROW) = rx[j];
    }
#pragma unroll
    for (int j = 0; j < DPER; ++j) {
      const int c = tid + NT * j;
      const int q = c / (NCO * 4), ch = c % (NCO * 4);
      if (c < DPIECES) *reinterpret_cast<u32x4*>(ds + (ch >> 2) * DPLANE + q * 64 + (ch & 3) * 16) = rd[j];
    }
  };

  // ---- transposing-read lane map (as conv_wgrad_bf16_kernel): the lane supplies the address of pixel 8 h + q4 (+ 4 for the second
  // read) and channels 16 half16 + 4 p4 .. + 3 of its tile; the hardware hands each lane 4 pixels of ITS channel
  const int li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const int half16 = (lane >> 4) & 1, h = lane >> 5;
  const int chan_off = (half16 * 16 + p4 * 4) * 2;
  const int cot = wave % NCO, cih = wave / NCO;
  const int l0 = 8 * h + q4;
  const unsigned char* const a_base = ds + cot * DPLANE + l0 * 64 + chan_off;
  const unsigned char* const x_base = xs + cih * XPLANE + chan_off;
  auto xoff = [&](int c, int s2) {   // window entry of pixel 16 c + 8 h + q4 + 4 s2 at tap kx = 0 (padding pixels repeat the last one)
    int p = 16 * c + l0 + 4 * s2;
    p = p < NPXR ? p : NPXR - 1;
    const int jr = p / TW;
    return jr * XROW + (p - jr * TW) * 64;
  };

  f32x16 acc[K];
#pragma unroll
  for (int v = 0; v < K; ++v)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[v][g] = 0.f;

  int tile = grp;
  if (tile < ntiles) {
    gload(tile);
    lstore();
    __syncthreads();
    struct Frag { bf16x8 a, b[K]; };
    auto fload = [&](int c, Frag& f) {
      const bf16x4 l = tr_read(a_base + (16 * c) * 64), hh = tr_read(a_base + (16 * c + 4) * 64);
      f.a = __builtin_shufflevector(l, hh, 0, 1, 2, 3, 4, 5, 6, 7);
      const int x0 = xoff(c, 0), x1 = xoff(c, 1);
#pragma unroll
      for (int v = 0; v < K; ++v) {
        const int to = ((v & 1) * PC + (v >> 1)) * 64;   // tap kx = v: parity plane kx & 1, entry shift kx >> 1
        const bf16x4 l2 = tr_read(x_base + x0 + to), h2 = tr_read(x_base + x1 + to);
        f.b[v] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    for (; tile < ntiles; tile += a.groups) {
      gload(tile + a.groups);   // next tile -> registers while this one is multiplied (out of range: zeros)
      Frag cur, nxt;
      fload(0, cur);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) fload(c + 1, nxt);
#pragma unroll
        for (int v = 0; v < K; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.a, cur.b[v], acc[v], 0, 0, 0);
        if (c + 1 < NCH) {
          // the 2 (K + 1) fragment reads of chunk c + 1 spread over the K MFMAs of chunk c
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
          for (int i = 1; i < K; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < NCH) cur = nxt;
      }
      __syncthreads();
      lstore();
      __syncthreads();
    }
  }

  // ---- one flush per workgroup: lane r = input channel (consecutive lanes -> 128-byte segments of an OHWI row)
  const int r = lane & 31;
  const int ci = cc * 64 + 32 * cih + r;
#pragma unroll
  for (int v = 0; v < K; ++v) {
    const int tap = ky * K + v;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int co = 32 * cot + (g & 3) + 8 * (g >> 2) + 4 * h;
      float* const q = a.dw + (size_t)grp * a.slab + ((size_t)co * (K * K) + tap) * a.Cin + ci;
      if (a.slab) *q = acc[v][g];
      else if (grp < ntiles) atomicAdd(q, acc[v][g]);
    }
  }
}

struct S2wShape { int K, P, H, Cin, Cout, TH, groups; };
// enc3: 8 waves, one workgroup per CU: 6 groups x 5 roles per XCD (30 of its 32 CUs); stem: 4 waves, two workgroups per CU:
// 2 groups x 28 roles per XCD (56 of 64 slots)
constexpr S2wShape kEnc3{5, 1, 50, 64, 128, 8, 48}, kStem{7, 3, 24, 256, 64, 12, 16};

const S2wShape* s2w_shape(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
  if (stride != 2 || KH != KW || H != W || OH != OW) return nullptr;
  if ((size_t)B * H * W * Cin * 2 >= (1ull << 31) || (size_t)B * OH * OW * Cout * 2 >= (1ull << 31)) return nullptr;
  if (!WSMG_TUNE("WSMG_WGRAD_S2WIN", 1)) return nullptr;
  for (const S2wShape* s : {&kEnc3, &kStem})
    if (KH == s->K && pad == s->P && H == s->H && Cin == s->Cin && Cout == s->Cout && OH == (H + 2 * pad - KH) / 2 + 1) return s;
  return nullptr;
}

int s2w_groups(const S2wShape* s, int B, int OH) {
  const int ntiles = B * ((OH + s->TH - 1) / s->TH);
  int groups = s->groups;
  if (groups > ntiles) groups = ntiles;
  return (groups + 7) / 8 * 8;   // whole XCD rounds (groups beyond the tile count find no tile and flush zeros)
}

template <int K, int P, int TH, int TW, int NCO, int NW, int NCC>
int launch_s2w(S2wArgs& a, hipStream_t s) {
  constexpr int WC = ((TW - 1) * 2 + K + 1) & ~1;
  constexpr int NPX = (TH * TW + 15) & ~15;
  constexpr int LDS = 2 * (TH * WC * 64 + 64) + NCO * (NPX * 64 + 64);
  static_assert(LDS <= 160 * 1024, "LDS of one CU");
  auto kern = conv_s2_wgrad_kernel<K, P, TH, TW, NCO, NW, NCC>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(K * NCC * a.groups)), dim3(NW * 64), LDS, s, a);
  WSMG_RETURN_LAUNCH();
}

}  // namespace

// tile groups (= slabs of the deterministic form) this kernel would use; 0: the layer is not this kernel's
int wsmg_conv_s2_wgrad_splits(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
  const S2wShape* s = s2w_shape(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW);
  return s ? s2w_groups(s, B, OH) : 0;
}

// dW (OHWI float32) of the two layers named at the top on bf16 NHWC — slab_floats == 0: accumulated into with float atomics (the
// caller zeroes it); > 0: one slab per tile group, stored; WSMG_EINVAL for any other shape (the caller then uses the generic kernel).
int wsmg_conv_s2_wgrad_bf16(const void* x, const void* dy, float* dw_ohwi, long long slab_floats, int B, int H, int W, int Cin, int Cout,
                            int KH, int KW, int stride, int pad, int OH, int OW, hipStream_t s) {
  const S2wShape* sh = s2w_shape(B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW);
  if (!sh) return WSMG_EINVAL;
  S2wArgs a{(const bf16_t*)x, (const bf16_t*)dy, dw_ohwi, (int64_t)slab_floats, B, H, W, OH, OW, Cin, (OH + sh->TH - 1) / sh->TH,
            s2w_groups(sh, B, OH), (unsigned)((size_t)B * H * W * Cin * 2), (unsigned)((size_t)B * OH * OW * Cout * 2)};
  if (sh == &kEnc3) return launch_s2w<5, 1, 8, 24, 4, 8, 1>(a, s);
  return launch_s2w<7, 3, 12, 12, 2, 4, 4>(a, s);
}
