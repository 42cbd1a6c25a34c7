// Operator 1: RGB-D -> egocentric bird's-eye-view map (rollout path, no backward).
// Replaces, from common/rgb_mapping.py of the reference:
//   ComputeSpatialLocs.forward :153-176, ProjectToGroundPlane.forward :184-232 (incl. the
//   third-party torch_scatter.scatter_max CUDA kernel), RotateTensor.forward :239-250,
//   to_grid.get_grid_coords :100-103, get_grid :106-139 and Mapping.project_feat_to_map :32-72.
//
// Integer part is bit-exact with the reference's float32 arithmetic: every operation below is
// the same IEEE float32 operation in the same order, with FMA contraction disabled for this
// file.  Sources the reference funnels to cell 0 with -1e16 (invalid depth / height filter /
// out of range) are skipped instead, cells are initialised "empty" and emitted as 0 — the
// same result without the cell-0 hot spot.
//
// HBM layout: features NCHW in (as the frozen UNet produces them), one E x E channel plane per
// workgroup lives in LDS for the scatter; everything downstream of the first rotation is NHWC
// (channel-contiguous, like the persistent global map [P][G][G][C]) so that the bilinear
// gathers and the max-fuse read-modify-write are fully coalesced 256-B channel runs.
#include "wsmg_common.h"

#pragma clang fp contract(off)

namespace {

// ----------------------------------------------------------------------------- index
struct IndexArgs {
  const float* depth;
  int32_t* lin;
  int B, Hd, Wd, Hf, Wf, E;
  float depth_scale, local_scale, half, cx, cy, fx, fy, K;
};

// (grid.y = sample, 32-bit index arithmetic inside it: the flat 64-bit form spent three 64-bit divisions per element — most of its
//  12.6 us at cfg4 — on finding (b, hf, wf))
__global__ __launch_bounds__(256) void bev_index_kernel(IndexArgs a) {
  const unsigned per = (unsigned)(a.Hf * a.Wf);
  const int b = blockIdx.y;
  for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < per; p += gridDim.x * blockDim.x) {
    const unsigned hfu = p / (unsigned)a.Wf;
    const int hf = (int)hfu, wf = (int)(p - hfu * (unsigned)a.Wf);
    const size_t i = (size_t)b * per + p;
    // (arange * K).long(): int64 * python float -> float32 product, truncated
    int ih = (int)((float)hf * a.K);
    int iw = (int)((float)wf * a.K);
    float z = a.depth[((size_t)b * a.Hd + ih) * a.Wd + iw] * a.depth_scale;
    float xx = ((float)iw - a.cx) / a.fx;
    float yy = ((float)(a.Hd - ih) - a.cy) / a.fy;
    float X = xx * z;
    float Y = yy * z;
    bool valid = (z != 0.f) && (Y > -1.5f) && (Y < 0.1f);
    float xg = rintf(X / a.local_scale + a.half);
    float yg = rintf(-(z / a.local_scale) + a.half);
    float Ef = (float)a.E;
    bool inr = (yg < Ef) && (yg >= 0.f) && (xg < Ef) && (xg >= 0.f);
    a.lin[i] = (valid && inr) ? (int)yg * a.E + (int)xg : -1;
  }
}

// Round 6 — the index launch also COMPACTS the valid sources.  75-80 % of the sources of a frame are invalid (no depth, above the
// horizon, outside the map: SURVEY section 7), and every (sample, channel) plane workgroup of the scatter walked all Hf x Wf index
// entries to find the rest — 40 times per sample at cfg4, eight dependent trips of index -> feature -> LDS atomic each.  Here a
// workgroup owns a block of CB = 8192 consecutive sources, writes their index entries as before, and packs the valid ones —
// (source << 16) | cell, both < 65 536 — to the front of ITS block of `clist`, count in cnt[sample][block]: no global atomics,
// nothing to zero, any order inside a block (the scatter is a max: order-independent, bit-exact).  The scatter then walks
// sum(cnt) entries instead of Hf x Wf.
constexpr int CB = 8192;
__device__ __forceinline__ int index_of(const IndexArgs& a, int b, unsigned p) {
  const unsigned hfu = p / (unsigned)a.Wf;
  const int hf = (int)hfu, wf = (int)(p - hfu * (unsigned)a.Wf);
  int ih = (int)((float)hf * a.K);
  int iw = (int)((float)wf * a.K);
  float z = a.depth[((size_t)b * a.Hd + ih) * a.Wd + iw] * a.depth_scale;
  float xx = ((float)iw - a.cx) / a.fx;
  float yy = ((float)(a.Hd - ih) - a.cy) / a.fy;
  float X = xx * z;
  float Y = yy * z;
  bool valid = (z != 0.f) && (Y > -1.5f) && (Y < 0.1f);
  float xg = rintf(X / a.local_scale + a.half);
  float yg = rintf(-(z / a.local_scale) + a.half);
  float Ef = (float)a.E;
  bool inr = (yg < Ef) && (yg >= 0.f) && (xg < Ef) && (xg >= 0.f);
  return (valid && inr) ? (int)yg * a.E + (int)xg : -1;
}
__global__ __launch_bounds__(1024) void bev_index_compact_kernel(IndexArgs a, unsigned* __restrict__ clist, int* __restrict__ cnt) {
  __shared__ int lcount;
  const unsigned per = (unsigned)(a.Hf * a.Wf);
  const int b = blockIdx.y, k = blockIdx.x;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  unsigned* const mine = clist + (size_t)b * per + (size_t)k * CB;
#pragma unroll 1
  for (int u = 0; u < CB / 1024; ++u) {
    const unsigned p = (unsigned)k * CB + threadIdx.x + 1024u * u;
    int v = -1;
    if (p < per) {
      v = index_of(a, b, p);
      a.lin[(size_t)b * per + p] = v;
    }
    const bool ok = v >= 0;
    const unsigned long long m = __ballot(ok);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(&lcount, __popcll(m));
      base = __shfl(base, leader, 64);
      if (ok) mine[base + __popcll(m & ((1ull << lane) - 1ull))] = (p << 16) | (unsigned)v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) cnt[b * (int)gridDim.x + k] = lcount;
}

// ----------------------------------------------------------------------------- scatter-max
__device__ __forceinline__ unsigned f2key(float v) {
  unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Scatter-max of one (sample, map channel) into its LDS plane `tg` (keys; 0 = empty), by the 1024 threads of a workgroup.
// adaptive_max_pool1d window of output channel c over the Cf feature channels: [ws, we).
// Memory-level parallelism is the whole game here: the plane leaves room for ONE workgroup per CU at E = 200 (16 waves), and a
// trip is index -> feature -> atomic.  Round 2 went from one source per thread and trip (14 GB/s per CU: pure latency) to eight
// with their loads issued together (16 GB/s per CU x 256); round 3 also issues the (up to four) window channels of all eight
// sources together instead of one channel per dependent loop trip, and fetches the NEXT trip's indices before this trip's
// features are used, so that an index round trip is never exposed.
// WU: window channels fetched together (the launcher picks the widest window of the geometry, up to 4; at Cf == C it is 1, and
// the 40 VGPRs of that form keep two workgroups per CU at E = 100, which the 78 of WU = 4 do not).
template <int WU>
__device__ __forceinline__ void scatter_plane(const float* __restrict__ feat, const int32_t* __restrict__ lb, int b, int c, int Cf,
                                              int HW, int C, unsigned* tg, int tid) {
  const int ws = (int)(((int64_t)c * Cf) / C);
  const int we = (int)((((int64_t)(c + 1)) * Cf + C - 1) / C);
  const float* fb = feat + ((size_t)b * Cf + ws) * HW;
  constexpr int U = 8;
  const int nwin = we - ws;
  if constexpr (WU == 1) {
    // Cf == C (the rollout geometry, one feature channel per map channel): the round-2 loop, which at B = 8 already runs at
    // 6.6 TB/s — the pipelined form below measured 30.6 instead of 23.8 us there (56 instead of 34 VGPRs)
    for (int s0 = tid; s0 < HW; s0 += 1024 * U) {
      int cell[U];
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = s0 + 1024 * u;
        cell[u] = s < HW ? lb[s] : -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = cell[u] >= 0 ? fb[s0 + 1024 * u] : 0.f;
      for (int w = 1; w < nwin; ++w) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (cell[u] >= 0) v[u] = fmaxf(v[u], fb[(size_t)w * HW + s0 + 1024 * u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (cell[u] >= 0) atomicMax(&tg[cell[u]], f2key(v[u]));
    }
    return;
  }
  int nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int s = tid + 1024 * u;
    nxt[u] = s < HW ? lb[s] : -1;
  }
  for (int s0 = tid; s0 < HW; s0 += 1024 * U) {
    int cell[U];
    float f[U][WU];
#pragma unroll
    for (int u = 0; u < U; ++u) cell[u] = nxt[u];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < WU; ++w) f[u][w] = (cell[u] >= 0 && w < nwin) ? fb[(size_t)w * HW + s0 + 1024 * u] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + 1024 * (U + u);
      nxt[u] = s < HW ? lb[s] : -1;
    }
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = f[u][0];
#pragma unroll
      for (int w = 1; w < WU; ++w)
        if (w < nwin) v[u] = fmaxf(v[u], f[u][w]);
    }
    for (int w = WU; w < nwin; ++w) {       // wider windows (Cf > 4 C): the rest one channel per trip
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (cell[u] >= 0) v[u] = fmaxf(v[u], fb[(size_t)w * HW + s0 + 1024 * u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (cell[u] >= 0) atomicMax(&tg[cell[u]], f2key(v[u]));
  }
}

// The same scatter from the compacted source list of bev_index_compact_kernel: `nblk` blocks of CB entries, cnt[k] valid at the
// front of block k.  The flat entry index i in [0, sum cnt) is mapped to (block, offset) against the prefix sums in registers.
template <int WU>
__device__ __forceinline__ void scatter_plane_compact(const float* __restrict__ feat, const unsigned* __restrict__ cl, const int* __restrict__ cnt,
                                                      int nblk, int b, int c, int Cf, int HW, int C, unsigned* tg, int tid) {
  const int ws = (int)(((int64_t)c * Cf) / C);
  const int we = (int)((((int64_t)(c + 1)) * Cf + C - 1) / C);
  const float* fb = feat + ((size_t)b * Cf + ws) * HW;
  const int nwin = we - ws;
  // (the list is short — ~16 k entries per sample at cfg4, 16 per thread — and every entry is a dependent chain entry -> feature ->
  //  LDS atomic: eight of them in flight per thread — 16 spilled at WU = 2 and cost the E = 100 geometry its second workgroup per CU)
  constexpr int U = 8;
  int off[9];            // prefix sums of the (up to 8) block counts
  off[0] = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) off[k + 1] = off[k] + (k < nblk ? cnt[k] : 0);
  const int N = off[8];
  auto entry = [&](int i) -> unsigned {
    if (i >= N) return 0xffffffffu;
    int k = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) k += (i >= off[j]) ? 1 : 0;
    return cl[(size_t)k * CB + (i - off[k])];
  };
  unsigned nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) nxt[u] = entry(tid + 1024 * u);
  for (int i0 = tid; i0 < N; i0 += 1024 * U) {
    unsigned e[U];
    float f[U][WU];
#pragma unroll
    for (int u = 0; u < U; ++u) e[u] = nxt[u];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int w = 0; w < WU; ++w) f[u][w] = (e[u] != 0xffffffffu && w < nwin) ? fb[(size_t)w * HW + (e[u] >> 16)] : 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) nxt[u] = entry(i0 + 1024 * (U + u));
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = f[u][0];
#pragma unroll
      for (int w = 1; w < WU; ++w)
        if (w < nwin) v[u] = fmaxf(v[u], f[u][w]);
    }
    for (int w = WU; w < nwin; ++w) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e[u] != 0xffffffffu) v[u] = fmaxf(v[u], fb[(size_t)w * HW + (e[u] >> 16)]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (e[u] != 0xffffffffu) atomicMax(&tg[e[u] & 0xffffu], f2key(v[u]));
  }
}

// one workgroup = (sample, CG consecutive map channels); the CG E*E planes live in LDS
template <int WU>
__global__ __launch_bounds__(1024) void bev_scatter_kernel(const float* __restrict__ feat,
                                                           const int32_t* __restrict__ lin, int Cf, int HW, int C,
                                                           int E2, int CG, float* __restrict__ out) {
  extern __shared__ unsigned tile[];
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CG;
  const int tid = threadIdx.x;
  for (int i = tid; i < CG * E2; i += 1024) tile[i] = 0u;  // 0 = empty (below every float key)
  __syncthreads();
  const int32_t* lb = lin + (size_t)b * HW;
  for (int g = 0; g < CG; ++g) {
    int c = c0 + g;
    if (c >= C) break;
    scatter_plane<WU>(feat, lb, b, c, Cf, HW, C, tile + (size_t)g * E2, tid);
  }
  __syncthreads();
  for (int i = tid; i < CG * E2; i += 1024) {
    int g = i / E2;
    if (c0 + g >= C) break;
    unsigned k = tile[i];
    out[((size_t)b * C + c0) * E2 + i] = k ? key2f(k) + 0.0f : 0.0f;
  }
}

// ----------------------------------------------------------------------------- sampling grid maths
// affine_grid(align_corners=False) base coordinate: linspace(-1,1,W)[j] * (W-1) / W
__device__ __forceinline__ float base_coord(int j, int W) {
  float step = 2.0f / (float)(W - 1);
  float v = (j < W / 2) ? -1.0f + step * (float)j : 1.0f - step * (float)(W - 1 - j);
  return v * (float)(W - 1) / (float)W;
}
// grid_sample(align_corners=False) un-normalisation
__device__ __forceinline__ float unnorm(float g, int W) { return ((g + 1.0f) * (float)W - 1.0f) / 2.0f; }

struct Taps {
  int x0, y0;          // north-west corner
  float w00, w01, w10, w11;  // nw, ne, sw, se
};
__device__ __forceinline__ Taps make_taps(float ix, float iy) {
  Taps t;
  float fx0 = floorf(ix), fy0 = floorf(iy);
  t.x0 = (int)fx0;
  t.y0 = (int)fy0;
  float x1 = fx0 + 1.0f, y1 = fy0 + 1.0f;
  t.w00 = (x1 - ix) * (y1 - iy);
  t.w01 = (ix - fx0) * (y1 - iy);
  t.w10 = (x1 - ix) * (iy - fy0);
  t.w11 = (ix - fx0) * (iy - fy0);
  return t;
}

struct Rot { float c, s; };

// Block order of the gather kernels: hardware block b runs on XCD b % 8; logical block = a contiguous eighth of the grid per XCD,
// so that output rows that share source rows (the +1 taps) meet in one L2 instead of fetching them into several.
__device__ __forceinline__ int bev_block() {
  const int nb = gridDim.x, bid = blockIdx.x;
  const int q = nb >> 3, r = nb & 7, x = bid & 7;
  return x * q + (x < r ? x : r) + (bid >> 3);
}

// rotation of an E x E map: gx = bx*c + by*s ; gy = -bx*s + by*c
__device__ __forceinline__ Taps rot_taps(int x, int y, int E, Rot r) {
  float bx = base_coord(x, E), by = base_coord(y, E);
  float gx = bx * r.c + by * r.s;
  float gy = bx * (-r.s) + by * r.c;
  return make_taps(unnorm(gx, E), unnorm(gy, E));
}

// scatter-max and the first rotation in ONE launch (round 3): the E x E plane of a channel is complete in LDS when the scatter
// ends, so the rotation samples it there (4 LDS reads per output pixel) and only the ROTATED plane goes to memory — NCHW, whole
// rows, coalesced — instead of plane out, plane in through 4-byte gathers, NHWC out.  map_fuse_planes_kernel consumes the planes.
// Same arithmetic per output pixel as rotate_nchw_to_nhwc_kernel (tap order nw, ne, sw, se).
// Round 5 — the rotation's arithmetic diet (cfg4: scatter alone 165 us, scatter + rotation 270-295: the rotation phase was VALU time,
// not LDS or HBM time — every (pixel, channel) item recomputed the tap geometry with four IEEE divisions (base_coord: 2 / (W - 1)
// and / W, twice) and an integer division p / E): the two base coordinates come from a table in the LDS left over beside the
// plane (E floats, the same function's values: bit-identical), y = p / E is a multiply-high by a host-computed magic number.
template <int WU>
__global__ __launch_bounds__(1024) void bev_scatter_rotate_kernel(const float* __restrict__ feat, const int32_t* __restrict__ lin,
                                                                  const float* __restrict__ heading, float sign, int Cf, int HW,
                                                                  int C, int E, int CG, unsigned magicE, int table,
                                                                  float* __restrict__ out, const unsigned* __restrict__ clist,
                                                                  const int* __restrict__ cnt, int nblk) {
  extern __shared__ unsigned tile[];
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * CG;
  const int tid = threadIdx.x;
  const int E2 = E * E;
  float* const bc = reinterpret_cast<float*>(tile + (size_t)CG * E2);    // [E] base coordinates (only when `table`)
  for (int i = tid; i < CG * E2; i += 1024) tile[i] = 0u;
  if (table)
    for (int i = tid; i < E; i += 1024) bc[i] = base_coord(i, E);
  __syncthreads();
  const int32_t* lb = lin + (size_t)b * HW;
  for (int g = 0; g < CG; ++g) {
    int c = c0 + g;
    if (c >= C) break;
    if (clist) scatter_plane_compact<WU>(feat, clist + (size_t)b * HW, cnt + (size_t)b * nblk, nblk, b, c, Cf, HW, C, tile + (size_t)g * E2, tid);
    else scatter_plane<WU>(feat, lb, b, c, Cf, HW, C, tile + (size_t)g * E2, tid);
  }
  __syncthreads();
  for (int i = tid; i < CG * E2; i += 1024) {   // keys -> the values bev_scatter_kernel writes, in place
    const unsigned k = tile[i];
    tile[i] = __float_as_uint(k ? key2f(k) + 0.0f : 0.0f);
  }
  __syncthreads();
  const float t = sign * heading[b];
  const Rot r{cosf(t), sinf(t)};
  for (int g = 0; g < CG; ++g) {
    const int c = c0 + g;
    if (c >= C) break;
    const float* pb = reinterpret_cast<const float*>(tile + (size_t)g * E2);
    float* ob = out + ((size_t)b * C + c) * E2;
    for (int p = tid; p < E2; p += 1024) {
      const int y = (int)__umulhi((unsigned)p, magicE), x = p - y * E;
      const float bx = table ? bc[x] : base_coord(x, E), by = table ? bc[y] : base_coord(y, E);
      const float gx = bx * r.c + by * r.s;
      const float gy = bx * (-r.s) + by * r.c;
      const Taps tp = make_taps(unnorm(gx, E), unnorm(gy, E));
      // branch-free (round 5, end): one unsigned compare per bound, every tap read from a clamped address, a tap outside the plane
      // dropped by a select AFTER its product (what the four branches did, without four exec-mask round trips per pixel)
      const bool x0ok = (unsigned)tp.x0 < (unsigned)E, x1ok = (unsigned)(tp.x0 + 1) < (unsigned)E;
      const bool y0ok = (unsigned)tp.y0 < (unsigned)E, y1ok = (unsigned)(tp.y0 + 1) < (unsigned)E;
      const int xa = x0ok ? tp.x0 : 0, xb = x1ok ? tp.x0 + 1 : 0;
      const int ya = y0ok ? tp.y0 * E : 0, yb = y1ok ? (tp.y0 + 1) * E : 0;
      const float q00 = pb[ya + xa], q01 = pb[ya + xb], q10 = pb[yb + xa], q11 = pb[yb + xb];
      float v = 0.f, t;
      t = v + q00 * tp.w00; v = (y0ok && x0ok) ? t : v;
      t = v + q01 * tp.w01; v = (y0ok && x1ok) ? t : v;
      t = v + q10 * tp.w10; v = (y1ok && x0ok) ? t : v;
      t = v + q11 * tp.w11; v = (y1ok && x1ok) ? t : v;
      ob[p] = v;
    }
  }
}

// first rotation: NCHW planes in (scatter output), NHWC out through an LDS transpose
__global__ __launch_bounds__(256) void rotate_nchw_to_nhwc_kernel(const float* __restrict__ in,
                                                                  const float* __restrict__ heading, float sign,
                                                                  int C, int E, float* __restrict__ out) {
  __shared__ float sh[64 * 65];
  // one workgroup = an 8 x 8 block of output pixels x all channels: the taps of a block fall into a ~12 x 12
  // source patch per channel plane (a 64 x 1 line would touch up to 64 source rows at 45 degrees)
  const int b = blockIdx.y;
  const int nbx = (E + 7) >> 3;
  const int by = blockIdx.x / nbx, bx = blockIdx.x - by * nbx;
  const int tid = threadIdx.x;
  const int pl = tid & 63, cq = tid >> 6;
  const int E2 = E * E;
  float t = sign * heading[b];
  Rot r{cosf(t), sinf(t)};
  const int y = by * 8 + (pl >> 3), x = bx * 8 + (pl & 7);
  const bool pok = y < E && x < E;
  Taps tp = rot_taps(pok ? x : 0, pok ? y : 0, E, r);
  bool x0ok = tp.x0 >= 0 && tp.x0 < E, x1ok = tp.x0 + 1 >= 0 && tp.x0 + 1 < E;
  bool y0ok = tp.y0 >= 0 && tp.y0 < E, y1ok = tp.y0 + 1 >= 0 && tp.y0 + 1 < E;
#pragma unroll 4
  for (int c = cq; c < C; c += 4) {
    const float* pb = in + ((size_t)b * C + c) * E2;
    float v = 0.f;
    if (pok) {
      if (y0ok && x0ok) v += pb[tp.y0 * E + tp.x0] * tp.w00;
      if (y0ok && x1ok) v += pb[tp.y0 * E + tp.x0 + 1] * tp.w01;
      if (y1ok && x0ok) v += pb[(tp.y0 + 1) * E + tp.x0] * tp.w10;
      if (y1ok && x1ok) v += pb[(tp.y0 + 1) * E + tp.x0 + 1] * tp.w11;
    }
    sh[pl * 65 + c] = v;
  }
  __syncthreads();
  for (int i = tid; i < 64 * C; i += 256) {
    int q = i / C, c = i - q * C;
    const int qy = by * 8 + (q >> 3), qx = bx * 8 + (q & 7);
    if (qy < E && qx < E) out[((size_t)b * E2 + qy * E + qx) * C + c] = sh[q * 65 + c];
  }
}

// final rotation: NHWC in, NHWC out (thread = (pixel, 4 channels): the tap geometry is computed once per
// 16 bytes instead of once per float, and every access is a 16-byte one; per-element arithmetic order unchanged)
__global__ __launch_bounds__(256) void rotate_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ heading,
                                                          float sign, int C, int E, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int E2 = E * E, C4 = C >> 2;
  float t = sign * heading[b];
  Rot r{cosf(t), sinf(t)};
  const f32x4* ib = reinterpret_cast<const f32x4*>(in + (size_t)b * E2 * C);
  f32x4* ob = reinterpret_cast<f32x4*>(out + (size_t)b * E2 * C);
  for (int64_t i = (int64_t)bev_block() * blockDim.x + threadIdx.x; i < (int64_t)E2 * C4;
       i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C4);
    int p = (int)(i / C4);
    int y = p / E, x = p - y * E;
    Taps tp = rot_taps(x, y, E, r);
    bool x0ok = tp.x0 >= 0 && tp.x0 < E, x1ok = tp.x0 + 1 >= 0 && tp.x0 + 1 < E;
    bool y0ok = tp.y0 >= 0 && tp.y0 < E, y1ok = tp.y0 + 1 >= 0 && tp.y0 + 1 < E;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (y0ok && x0ok) { f32x4 q = ib[((size_t)tp.y0 * E + tp.x0) * C4 + c]; for (int j = 0; j < 4; ++j) v[j] += q[j] * tp.w00; }
    if (y0ok && x1ok) { f32x4 q = ib[((size_t)tp.y0 * E + tp.x0 + 1) * C4 + c]; for (int j = 0; j < 4; ++j) v[j] += q[j] * tp.w01; }
    if (y1ok && x0ok) { f32x4 q = ib[((size_t)(tp.y0 + 1) * E + tp.x0) * C4 + c]; for (int j = 0; j < 4; ++j) v[j] += q[j] * tp.w10; }
    if (y1ok && x1ok) { f32x4 q = ib[((size_t)(tp.y0 + 1) * E + tp.x0 + 1) * C4 + c]; for (int j = 0; j < 4; ++j) v[j] += q[j] * tp.w11; }
    ob[i] = v;
  }
}

// ----------------------------------------------------------------------------- global map
struct Pose { float gx, gy; };  // integer-valued grid cell of the agent (to_grid.get_grid_coords)
__device__ __forceinline__ Pose grid_cell(const float* gps, int b, int G, float cmax, float cmin, float gsz) {
  Pose p;
  p.gx = rintf((cmax - gps[b * 2 + 0]) / gsz);
  p.gy = rintf((gps[b * 2 + 1] - cmin) / gsz);
  return p;
}

// full_global_map[:bs] *= masks (episode reset).  Untouched when mask == 1.
__global__ __launch_bounds__(256) void map_reset_kernel(float* gm, const float* masks, int64_t n4_per_env) {
  const int b = blockIdx.y;
  const float m = masks[b];
  if (m == 1.0f) return;
  f32x4* g = reinterpret_cast<f32x4*>(gm) + (size_t)b * n4_per_env;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4_per_env; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 v = g[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= m;
    g[i] = v;
  }
}

struct MapArgs {
  int B, C, E, G, lo;
  float cmax, cmin, gsz, halfG;
};

// paste (centre) + translate (bilinear) + max-fuse, restricted to the (E+4)^2 window the pasted
// map can reach; outside it the translated view is exactly 0 and the map (>= 0) is unchanged.
__global__ __launch_bounds__(256) void map_fuse_kernel(const float* __restrict__ ego, float* __restrict__ gm,
                                                       const float* __restrict__ gps, MapArgs a) {
  const int b = blockIdx.y;
  const int WN = a.E + 4;
  Pose ps = grid_cell(gps, b, a.G, a.cmax, a.cmin, a.gsz);
  const float tx = -(ps.gy - a.halfG) / a.halfG;
  const float ty = -(ps.gx - a.halfG) / a.halfG;
  const int wy0 = a.lo + (int)(ps.gx - a.halfG) - 2;
  const int wx0 = a.lo + (int)(ps.gy - a.halfG) - 2;
  const int C4 = a.C >> 2;
  const f32x4* eb = reinterpret_cast<const f32x4*>(ego + (size_t)b * a.E * a.E * a.C);
  f32x4* gb = reinterpret_cast<f32x4*>(gm + (size_t)b * a.G * a.G * a.C);
  const int hi = a.lo + a.E;
  for (int64_t i = (int64_t)bev_block() * blockDim.x + threadIdx.x; i < (int64_t)WN * WN * C4;
       i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C4);
    int p = (int)(i / C4);
    int wy = p / WN, wx = p - wy * WN;
    int Y = wy0 + wy, X = wx0 + wx;
    if (Y < 0 || Y >= a.G || X < 0 || X >= a.G) continue;
    float gx = base_coord(X, a.G) + tx;
    float gy = base_coord(Y, a.G) + ty;
    Taps tp = make_taps(unnorm(gx, a.G), unnorm(gy, a.G));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int yy = tp.y0 + (k >> 1), xx = tp.x0 + (k & 1);
      float w = k == 0 ? tp.w00 : k == 1 ? tp.w01 : k == 2 ? tp.w10 : tp.w11;
      // zero padding of grid_sample, then the zero border of the agent view around the paste
      if (yy >= a.lo && yy < hi && xx >= a.lo && xx < hi && yy < a.G && xx < a.G) {
        const f32x4 q = eb[((size_t)(yy - a.lo) * a.E + (xx - a.lo)) * C4 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += q[j] * w;
      }
    }
    size_t o = ((size_t)Y * a.G + X) * C4 + c;
    f32x4 g = gb[o];
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = v[j] > g[j] ? v[j] : g[j];
    gb[o] = g;
  }
}

// map_fuse_kernel for an ego map that arrives as ROTATED NCHW PLANES (bev_scatter_rotate_kernel).  A workgroup owns FP
// consecutive window pixels of one window row and all channels, in two phases:
//   1. work item = (channel, pixel) with the PIXEL fastest: the four taps are 4-byte loads that run along a plane row (coalesced,
//      independent — several items in flight per thread), the translated value goes to LDS as t[pixel][channel];
//   2. work item = (pixel, 4 channels) with the channel quad fastest, as in map_fuse_kernel: the value comes back from LDS as one
//      16-byte read and the global map is read-modify-written in 16-byte pieces of consecutive addresses.
// Same arithmetic per element (taps in k order).  (First form of this kernel: the (FR + 2) x (FC + 4) patch of every plane in LDS,
// then gathers out of it — 65 KB per workgroup, two workgroups per CU, load and compute phases that could not overlap: 334 us at
// cfg4 against 144 us of map_fuse_kernel.)
constexpr int FP = 64;
__global__ __launch_bounds__(256) void map_fuse_planes_kernel(const float* __restrict__ ego, float* __restrict__ gm,
                                                              const float* __restrict__ gps, MapArgs a, int tiles_x) {
  extern __shared__ float tl[];           // [FP][C + 1]... pitch C + 4 keeps the 16-byte reads aligned
  const int b = blockIdx.y;
  const int wy = blockIdx.x / tiles_x, tx_ = blockIdx.x - wy * tiles_x;
  const int WN = a.E + 4;
  Pose ps = grid_cell(gps, b, a.G, a.cmax, a.cmin, a.gsz);
  const float tx = -(ps.gy - a.halfG) / a.halfG;
  const float ty = -(ps.gx - a.halfG) / a.halfG;
  const int wy0 = a.lo + (int)(ps.gx - a.halfG) - 2;
  const int wx0 = a.lo + (int)(ps.gy - a.halfG) - 2;
  const int C4 = a.C >> 2;
  const int E2 = a.E * a.E;
  const int pitch = a.C + 4;
  const int hi = a.lo + a.E;
  const int Y = wy0 + wy;
  if (Y < 0 || Y >= a.G) return;          // (whole workgroup: the row is outside the global map)
  const float* eb = ego + (size_t)b * a.C * E2;
  // ---- phase 1
  {
    const int p = threadIdx.x & (FP - 1), cq = threadIdx.x >> 6;     // pixel of the tile, channel phase (256 / FP = 4)
    const int wx = tx_ * FP + p, X = wx0 + wx;
    const bool pok = wx < WN && X >= 0 && X < a.G;
    float gx = base_coord(pok ? X : 0, a.G) + tx;
    float gy = base_coord(Y, a.G) + ty;
    Taps tp = make_taps(unnorm(gx, a.G), unnorm(gy, a.G));
    int off[4];
    float w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = tp.y0 + (k >> 1), xx = tp.x0 + (k & 1);
      w[k] = k == 0 ? tp.w00 : k == 1 ? tp.w01 : k == 2 ? tp.w10 : tp.w11;
      // zero padding of grid_sample, then the zero border of the agent view around the paste
      const bool ok = pok && yy >= a.lo && yy < hi && xx >= a.lo && xx < hi && yy < a.G && xx < a.G;
      off[k] = ok ? (yy - a.lo) * a.E + (xx - a.lo) : -1;
    }
    constexpr int U = 4;
    for (int c0 = cq; c0 < a.C; c0 += 4 * U) {
      float q[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + 4 * u;
#pragma unroll
        for (int k = 0; k < 4; ++k) q[u][k] = (c < a.C && off[k] >= 0) ? eb[(size_t)c * E2 + off[k]] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + 4 * u;
        if (c >= a.C) break;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (off[k] >= 0) v += q[u][k] * w[k];
        tl[p * pitch + c] = v;
      }
    }
  }
  __syncthreads();
  // ---- phase 2
  f32x4* gb = reinterpret_cast<f32x4*>(gm + (size_t)b * a.G * a.G * a.C);
  constexpr int V = 4;
  const int nitems = FP * C4;
  for (int i0 = threadIdx.x; i0 < nitems; i0 += 256 * V) {
    bool ok[V];
    size_t o[V];
    f32x4 g[V];
    int li[V];
#pragma unroll
    for (int u = 0; u < V; ++u) {
      const int i = i0 + 256 * u;
      const int c = i % C4, p = i / C4;
      const int wx = tx_ * FP + p, X = wx0 + wx;
      ok[u] = i < nitems && wx < WN && X >= 0 && X < a.G;
      o[u] = ((size_t)Y * a.G + X) * C4 + c;
      li[u] = p * pitch + 4 * c;
      if (ok[u]) g[u] = gb[o[u]];
    }
#pragma unroll
    for (int u = 0; u < V; ++u) {
      if (!ok[u]) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tl + li[u]);
      f32x4 gg = g[u];
#pragma unroll
      for (int j = 0; j < 4; ++j) gg[j] = v[j] > gg[j] ? v[j] : gg[j];
      gb[o[u]] = gg;
    }
  }
}

// translate the global map back to the agent and crop the centre E x E (NHWC scratch)
__global__ __launch_bounds__(256) void map_crop_kernel(const float* __restrict__ gm, const float* __restrict__ gps,
                                                       MapArgs a, float* __restrict__ crop) {
  const int b = blockIdx.y;
  Pose ps = grid_cell(gps, b, a.G, a.cmax, a.cmin, a.gsz);
  const float tx = (ps.gy - a.halfG) / a.halfG;
  const float ty = (ps.gx - a.halfG) / a.halfG;
  const int C4 = a.C >> 2;
  const f32x4* gb = reinterpret_cast<const f32x4*>(gm + (size_t)b * a.G * a.G * a.C);
  f32x4* cb = reinterpret_cast<f32x4*>(crop + (size_t)b * a.E * a.E * a.C);
  for (int64_t i = (int64_t)bev_block() * blockDim.x + threadIdx.x; i < (int64_t)a.E * a.E * C4;
       i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C4);
    int p = (int)(i / C4);
    int y = p / a.E, x = p - y * a.E;
    float gx = base_coord(a.lo + x, a.G) + tx;
    float gy = base_coord(a.lo + y, a.G) + ty;
    Taps tp = make_taps(unnorm(gx, a.G), unnorm(gy, a.G));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int yy = tp.y0 + (k >> 1), xx = tp.x0 + (k & 1);
      float w = k == 0 ? tp.w00 : k == 1 ? tp.w01 : k == 2 ? tp.w10 : tp.w11;
      if (yy >= 0 && yy < a.G && xx >= 0 && xx < a.G) {
        const f32x4 q = gb[((size_t)yy * a.G + xx) * C4 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += q[j] * w;
      }
    }
    cb[i] = v;
  }
}

// map_crop_kernel + rotate_nhwc_kernel in one pass (round 3): a rotation tap needs the cropped map at an integer pixel, which is
// itself four taps of the global map — the item (output pixel, 4 channels) computes its four crop values in registers (16 loads of
// 16 bytes, all independent, neighbours' loads hitting the same lines in L1 / L2) instead of the crop going to memory (205 MB out,
// 205 MB back at cfg4) between two launches.  Same arithmetic per value as the two kernels: crop = sum over k of q_k w_k, output =
// sum over the rotation taps (nw, ne, sw, se) of crop w.
__device__ __forceinline__ f32x4 retrieve_item_regs(const f32x4* __restrict__ gb, const MapArgs& a, float tx, float ty, Rot r,
                                                    int x, int y, int c) {
  const int C4 = a.C >> 2, E = a.E;
  const Taps rt = rot_taps(x, y, E, r);
  f32x4 q[4][4];
  float w[4][4];
  bool rok[4], cok[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {          // rotation tap k = crop pixel (cy, cx)
    const int cy = rt.y0 + (k >> 1), cx = rt.x0 + (k & 1);
    rok[k] = cy >= 0 && cy < E && cx >= 0 && cx < E;
    float gx = base_coord(a.lo + (rok[k] ? cx : 0), a.G) + tx;
    float gy = base_coord(a.lo + (rok[k] ? cy : 0), a.G) + ty;
    const Taps tp = make_taps(unnorm(gx, a.G), unnorm(gy, a.G));
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int yy = tp.y0 + (m >> 1), xx = tp.x0 + (m & 1);
      w[k][m] = m == 0 ? tp.w00 : m == 1 ? tp.w01 : m == 2 ? tp.w10 : tp.w11;
      cok[k][m] = rok[k] && yy >= 0 && yy < a.G && xx >= 0 && xx < a.G;
      if (cok[k][m]) q[k][m] = gb[((size_t)yy * a.G + xx) * C4 + c];
    }
  }
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!rok[k]) continue;
    f32x4 cv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (cok[k][m]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cv[j] += q[k][m][j] * w[k][m];
      }
    const float wr = k == 0 ? rt.w00 : k == 1 ? rt.w01 : k == 2 ? rt.w10 : rt.w11;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += cv[j] * wr;
  }
  return v;
}

// the same value, one rotation tap at a time (4 loads in flight instead of 16: the tiled kernel's rare route, few registers)
__device__ __noinline__ f32x4 retrieve_item_seq(const f32x4* __restrict__ gb, const MapArgs& a, float tx, float ty, Rot r, int x, int y,
                                                int c) {
  const int C4 = a.C >> 2, E = a.E;
  const Taps rt = rot_taps(x, y, E, r);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int cy = rt.y0 + (k >> 1), cx = rt.x0 + (k & 1);
    if (!(cy >= 0 && cy < E && cx >= 0 && cx < E)) continue;
    const float gx = base_coord(a.lo + cx, a.G) + tx;
    const float gy = base_coord(a.lo + cy, a.G) + ty;
    const Taps tp = make_taps(unnorm(gx, a.G), unnorm(gy, a.G));
    f32x4 cv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int yy = tp.y0 + (m >> 1), xx = tp.x0 + (m & 1);
      const float w = m == 0 ? tp.w00 : m == 1 ? tp.w01 : m == 2 ? tp.w10 : tp.w11;
      if (yy >= 0 && yy < a.G && xx >= 0 && xx < a.G) {
        const f32x4 q = gb[((size_t)yy * a.G + xx) * C4 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) cv[j] += q[j] * w;
      }
    }
    const float wr = k == 0 ? rt.w00 : k == 1 ? rt.w01 : k == 2 ? rt.w10 : rt.w11;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += cv[j] * wr;
  }
  return v;
}

__global__ __launch_bounds__(256) void map_retrieve_fused_kernel(const float* __restrict__ gm, const float* __restrict__ gps,
                                                                 const float* __restrict__ heading, MapArgs a,
                                                                 float* __restrict__ out) {
  const int b = blockIdx.y;
  Pose ps = grid_cell(gps, b, a.G, a.cmax, a.cmin, a.gsz);
  const float tx = (ps.gy - a.halfG) / a.halfG;
  const float ty = (ps.gx - a.halfG) / a.halfG;
  const int C4 = a.C >> 2, E = a.E;
  const float t = heading[b];
  const Rot r{cosf(t), sinf(t)};
  const f32x4* gb = reinterpret_cast<const f32x4*>(gm + (size_t)b * a.G * a.G * a.C);
  f32x4* ob = reinterpret_cast<f32x4*>(out + (size_t)b * E * E * a.C);
  for (int64_t i = (int64_t)bev_block() * blockDim.x + threadIdx.x; i < (int64_t)E * E * C4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const int p = (int)(i / C4);
    const int y = p / E, x = p - y * E;
    ob[i] = retrieve_item_regs(gb, a, tx, ty, r, x, y, c);
  }
}

// map_crop_kernel + rotate_nhwc_kernel in one pass through LDS (round 5).  The register form above pays 16 gathers of 16 bytes per
// item through the texture path (cfg4: 311 us against 219 for the two launches, which is why it only ran at small batches); here a
// workgroup owns an 8 x 8 tile of output pixels (and a slice of <= 40 channels), works out which global-map pixels the crop pixels
// under its rotation taps touch (<= 14 x 14: 7 sqrt 2 + 2 crop pixels, one more for their own taps, one for a floor that slips),
// stages that box once — rows of the box, pixels outside the map as zeros — and takes all 16 taps of an item from LDS.  The
// tap geometry is computed once per (pixel, rotation tap) — 256 records, one per thread: box offset, four crop weights, rotation
// weight — instead of once per (pixel, 4 channels) item (base_coord's IEEE divisions were the VALU time of the gather kernels),
// from the same expressions, and each value keeps the two kernels' summation order: bit-identical.  Where the two kernels SKIP a
// tap (outside the map, outside the crop) the record points at zeros with zero weights: an accumulator that starts at +0 is never
// -0, so adding +-0 leaves it unchanged.  A tile whose geometry does not fit the box (cannot happen for finite inputs: the spans
// above are bounds; an infinite gps does it) runs the register form instead.
// Latency: a tile is little work behind a chain of dependent steps (pose -> taps -> box extents -> box -> taps from LDS), and LDS
// holds few tiles per CU, so the chain is kept short: every thread derives its record from the pose alone (no tables, no LDS
// round trips), the extents meet through one wave reduction (DPP) and 16 words of LDS — LDS atomics from 64 lanes on one word
// took 20 000 of a workgroup's 32 000 cycles — and the box's loads are all in flight before the first LDS store.
constexpr int RT = 8, RBOX = 14, RBIG = 0x10000000;
struct Axis { int p0; float a, b; };  // floor(i), (floor(i) + 1) - i, i - floor(i): one axis of make_taps
__device__ __forceinline__ Axis crop_axis(int j, float t, const MapArgs& a) {
  const float i = unnorm(base_coord(a.lo + j, a.G) + t, a.G);
  const float f0 = floorf(i);
  Axis r;
  r.p0 = (int)f0;
  r.a = (f0 + 1.0f) - i;
  r.b = i - f0;
  return r;
}
// min / max over the lane's row of 16 (DPP: no LDS crossbar), in every lane of the row
template <bool MAX>
__device__ __forceinline__ int row_ext(int v) {
#define WSMG_EXT_STEP(ctrl)                                                      \
  {                                                                              \
    const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false);      \
    v = MAX ? (o > v ? o : v) : (o < v ? o : v);                                 \
  }
  WSMG_EXT_STEP(0xB1)    // quad_perm [1, 0, 3, 2]
  WSMG_EXT_STEP(0x4E)    // quad_perm [2, 3, 0, 1]
  WSMG_EXT_STEP(0x141)   // row_half_mirror
  WSMG_EXT_STEP(0x140)   // row_mirror
#undef WSMG_EXT_STEP
  return v;
}
// lane K of the caller's quad, in every lane of the quad
template <int K>
__device__ __forceinline__ int quad_get(int v) { return __builtin_amdgcn_update_dpp(v, v, K * 0x55, 0xf, 0xf, false); }
template <int K>
__device__ __forceinline__ float quad_getf(float v) { return __int_as_float(quad_get<K>(__float_as_int(v))); }

__global__ __launch_bounds__(256, 4) void map_retrieve_tiled_kernel(const float* __restrict__ gm, const float* __restrict__ gps,
                                                                 const float* __restrict__ heading, MapArgs a, int tiles_x,
                                                                 int nsplit, int S4, unsigned magicS4, float* __restrict__ out,
                                                                 unsigned* __restrict__ trace) {
  extern __shared__ float boxf[];  // [bh][bw][4 S4] floats: the box; then bw + 2 pixels of zeros
  __shared__ int s_ext[16][4];
  // diagnostic build (`tracing` in the launcher): cycles of every 64th workgroup at its phase boundaries
  const bool tr = trace && threadIdx.x == 0 && (blockIdx.x & 63) == 0 && blockIdx.y == 0 && (blockIdx.x >> 6) < 64;
  unsigned* const trw = trace + (blockIdx.x >> 6) * 8;
  const unsigned long long t00 = tr ? __builtin_readcyclecounter() : 0;
#define RTRACE(i) if (tr) trw[i] = (unsigned)(__builtin_readcyclecounter() - t00)
  const int b = blockIdx.y, tid = threadIdx.x;
  const int blk = bev_block();
  const int tile = blk / nsplit, cs0 = (blk - tile * nsplit) * S4;   // first f32x4 of this workgroup's channel slice
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  Pose ps = grid_cell(gps, b, a.G, a.cmax, a.cmin, a.gsz);
  const float tx = (ps.gy - a.halfG) / a.halfG;
  const float ty = (ps.gx - a.halfG) / a.halfG;
  const int C4 = a.C >> 2, E = a.E, G = a.G;
  const f32x4* gb = reinterpret_cast<const f32x4*>(gm + (size_t)b * G * G * a.C);
  f32x4* ob = reinterpret_cast<f32x4*>(out + (size_t)b * E * E * a.C);
  f32x4* box = reinterpret_cast<f32x4*>(boxf);
  const float th = heading[b];
  const Rot r{cosf(th), sinf(th)};

  // ---- this thread's record: pixel tid / 4 of the tile, rotation tap tid % 4 = crop pixel (cy, cx)
  const int pl_r = tid >> 2, k_r = tid & 3;
  const int ry = tyi * RT + (pl_r >> 3), rx = txi * RT + (pl_r & 7);
  const bool live = ry < E && rx < E;
  const Taps rt = rot_taps(live ? rx : 0, live ? ry : 0, E, r);
  const int cy = rt.y0 + (k_r >> 1), cx = rt.x0 + (k_r & 1);
  const bool rok = live && cy >= 0 && cy < E && cx >= 0 && cx < E;
  const Axis ax = crop_axis(rok ? cx : 0, tx, a), ay = crop_axis(rok ? cy : 0, ty, a);
  // a coordinate that is not finite and small (an infinite / NaN gps): the floors saturate and the weights are NaN while the two
  // kernels skip the taps — not this route's case
  const bool sane = fabsf(ax.a) <= 2.0f && fabsf(ax.b) <= 2.0f && fabsf(ay.a) <= 2.0f && fabsf(ay.b) <= 2.0f && ax.p0 > -RBIG &&
                    ax.p0 < RBIG && ay.p0 > -RBIG && ay.p0 < RBIG;
  const bool use = rok && sane;
  {  // (a record that is not sane stretches the box beyond RBOX: the tile takes the register form)
    const int xlo = row_ext<false>(use ? ax.p0 : (rok ? -2 * RBIG : RBIG)), xhi = row_ext<true>(use ? ax.p0 + 1 : -RBIG);
    const int ylo = row_ext<false>(use ? ay.p0 : RBIG), yhi = row_ext<true>(use ? ay.p0 + 1 : -RBIG);
    if ((tid & 15) == 0) {
      int* e = s_ext[tid >> 4];
      e[0] = xlo; e[1] = xhi; e[2] = ylo; e[3] = yhi;
    }
  }
  RTRACE(0);
  __syncthreads();
  RTRACE(1);
  int gx_lo = RBIG, gx_hi = -RBIG, gy_lo = RBIG, gy_hi = -RBIG;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    gx_lo = s_ext[w][0] < gx_lo ? s_ext[w][0] : gx_lo; gx_hi = s_ext[w][1] > gx_hi ? s_ext[w][1] : gx_hi;
    gy_lo = s_ext[w][2] < gy_lo ? s_ext[w][2] : gy_lo; gy_hi = s_ext[w][3] > gy_hi ? s_ext[w][3] : gy_hi;
  }
  int bw = 2, bh = 0;                         // no tap of the tile lands in the crop: zeros only
  bool slow = false;                          // uniform
  if (gx_lo != RBIG) {
    slow = gx_lo < -RBIG || gx_hi - gx_lo + 1 > RBOX || gy_hi - gy_lo + 1 > RBOX;
    bw = gx_hi - gx_lo + 1; bh = gy_hi - gy_lo + 1;
  } else {
    gx_lo = gy_lo = 0;
  }
  if (slow) {  // the register form for this tile
    for (int i = tid; i < RT * RT * S4; i += 256) {
      const int pl = (int)__umulhi((unsigned)i, magicS4), c = cs0 + i - pl * S4;
      const int qy = tyi * RT + (pl >> 3), qx = txi * RT + (pl & 7);
      if (qy < E && qx < E) ob[((size_t)qy * E + qx) * C4 + c] = retrieve_item_seq(gb, a, tx, ty, r, qx, qy, c);
    }
    return;
  }
  // the record stays in registers: the four records of a pixel sit in one quad
  int off = bh * bw;
  f32x4 w = {0.f, 0.f, 0.f, 0.f};
  float wr = 0.f;
  if (use) {
    off = (ay.p0 - gy_lo) * bw + (ax.p0 - gx_lo);
    w[0] = ax.a * ay.a; w[1] = ax.b * ay.a; w[2] = ax.a * ay.b; w[3] = ax.b * ay.b;
    wr = k_r == 0 ? rt.w00 : k_r == 1 ? rt.w01 : k_r == 2 ? rt.w10 : rt.w11;
  }
  // ---- stage the box: rows of the box (this slice's channels of each pixel), out-of-map pixels and the tail as zeros; a batch's
  // loads all before its LDS stores (one load, one store per trip paid the memory latency per trip), branch-free
  {
    const unsigned rowq = (unsigned)(bw * S4), mrow = 0xFFFFFFFFu / rowq + 1u;
    const unsigned nbox = (unsigned)bh * rowq, nq = nbox + (unsigned)(bw + 2) * S4;
    constexpr int UB = 9;   // (14 x 14 + 16) pixels x 10 / 256 = 8.3: one batch
    for (unsigned i0 = tid; i0 < nq; i0 += 256 * UB) {
      f32x4 q[UB];
      bool ok[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const unsigned i = i0 + 256 * u;
        const unsigned row = __umulhi(i, mrow), rem = i - row * rowq;
        const unsigned px = __umulhi(rem, magicS4), c = rem - px * S4;
        const int yy = gy_lo + (int)row, xx = gx_lo + (int)px;
        ok[u] = i < nbox && yy >= 0 && yy < G && xx >= 0 && xx < G;
        q[u] = gb[ok[u] ? ((size_t)yy * G + xx) * C4 + cs0 + c : (size_t)0];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (i0 + 256 * u < nq) box[i0 + 256 * u] = ok[u] ? q[u] : z;
      }
    }
  }
  RTRACE(2);
  __syncthreads();
  RTRACE(3);
  // ---- thread = (its record's pixel, channel groups tid % 4, + 4, ...): the pixel's four records come from the quad (DPP, as
  // they are used: held for the whole loop they cost 24 registers and the kernel its fourth workgroup per CU)
  const int dn = bw * S4;
  const int offS = off * S4;
  for (int c = k_r; c < S4 + k_r; c += 4) {   // (uniform trip count: the quad broadcasts need the whole quad)
    const bool on = live && c < S4;
    const int cc = c < S4 ? c : 0;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#define WSMG_TAP(K)                                                                                                        \
    {                                                                                                                        \
      const f32x4* p = box + quad_get<K>(offS) + cc;                                                                         \
      const f32x4 q0 = p[0], q1 = p[S4], q2 = p[dn], q3 = p[dn + S4];                                                        \
      const float w0 = quad_getf<K>(w[0]), w1 = quad_getf<K>(w[1]), w2 = quad_getf<K>(w[2]), w3 = quad_getf<K>(w[3]);        \
      const float wk = quad_getf<K>(wr);                                                                                     \
      f32x4 cv = {0.f, 0.f, 0.f, 0.f};                                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) cv[j] += q0[j] * w0;                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) cv[j] += q1[j] * w1;                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) cv[j] += q2[j] * w2;                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) cv[j] += q3[j] * w3;                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) v[j] += cv[j] * wk;                                                      \
    }
    WSMG_TAP(0) WSMG_TAP(1) WSMG_TAP(2) WSMG_TAP(3)
#undef WSMG_TAP
    if (on) ob[((size_t)ry * E + rx) * C4 + cs0 + c] = v;
  }
  RTRACE(4);
#undef RTRACE
}

MapArgs map_args(int B, int C, int E, int G, float resolution) {
  MapArgs a;
  a.B = B; a.C = C; a.E = E; a.G = G;
  a.lo = G / 2 - E / 2;  // G//2 - floor(E/2)
  double cmin = -(double)G * (double)resolution / 2, cmax = (double)G * (double)resolution / 2;
  a.cmax = (float)cmax;
  a.cmin = (float)cmin;
  a.gsz = (float)((cmax - cmin) / G);
  a.halfG = (float)(G / 2);
  return a;
}

// widest adaptive_max_pool1d window of Cf -> C channels, capped at 4 (scatter_plane's WU)
int scatter_window(int Cf, int C) {
  int m = 1;
  for (int c = 0; c < C; ++c) {
    const int ws = (int)(((int64_t)c * Cf) / C), we = (int)((((int64_t)(c + 1)) * Cf + C - 1) / C);
    if (we - ws > m) m = we - ws;
  }
  return m > 4 ? 4 : m;
}

int sgrid(int64_t n, int cap = 4096) {
  int64_t g = wsmg_cdiv(n, 256);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int wsmg_bev_index(const float* depth, int B, int Hd, int Wd, float depth_scale, int Hf, int Wf, int E,
                              float local_scale, int32_t* lin_idx, wsmg_stream_t stream) {
  if (B <= 0 || Hd <= 0 || Wd <= 0 || Hf <= 0 || Wf <= 0 || E <= 0 || Hf > Hd || Wf > Wd) return WSMG_EINVAL;
  IndexArgs a;
  a.depth = depth; a.lin = lin_idx;
  a.B = B; a.Hd = Hd; a.Wd = Wd; a.Hf = Hf; a.Wf = Wf; a.E = E;
  a.depth_scale = depth_scale;
  a.local_scale = local_scale;
  a.half = (float)((E - 1) / 2.0);
  // get_camera_matrix(imh, imw, 90): cx=imh/2, cy=imw/2, fx=(imh/2)/tan(45deg), fy=(imw/2)/tan(45deg) (float64 -> f32)
  const double tn = tan(45.0 * 3.14159265358979323846 / 180.0);
  a.cx = (float)(Hd / 2.0); a.cy = (float)(Wd / 2.0);
  a.fx = (float)((Hd / 2.0) / tn); a.fy = (float)((Wd / 2.0) / tn);
  a.K = (float)((double)Wd / (double)Wf);
  if (B > 65535 || (int64_t)Hf * Wf >= (1ll << 31)) return WSMG_EINVAL;
  hipLaunchKernelGGL(bev_index_kernel, dim3(sgrid((int64_t)Hf * Wf, 1024), (unsigned)B), dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

// wsmg_bev_index + the compacted list of valid sources (round 6): clist [B][Hf*Wf] uint32, cnt [B][ceil(Hf*Wf / 8192)] int32, both
// written in full by this launch (nothing to zero); needs Hf*Wf <= 65536 and E*E <= 65536.
extern "C" int wsmg_bev_index_compact(const float* depth, int B, int Hd, int Wd, float depth_scale, int Hf, int Wf, int E,
                                      float local_scale, int32_t* lin_idx, uint32_t* clist, int32_t* cnt, wsmg_stream_t stream) {
  if (B <= 0 || Hd <= 0 || Wd <= 0 || Hf <= 0 || Wf <= 0 || E <= 0 || Hf > Hd || Wf > Wd || B > 65535) return WSMG_EINVAL;
  if ((int64_t)Hf * Wf > 65536 || (int64_t)E * E > 65536 || !clist || !cnt || !lin_idx) return WSMG_EINVAL;
  IndexArgs a;
  a.depth = depth; a.lin = lin_idx;
  a.B = B; a.Hd = Hd; a.Wd = Wd; a.Hf = Hf; a.Wf = Wf; a.E = E;
  a.depth_scale = depth_scale;
  a.local_scale = local_scale;
  a.half = (float)((E - 1) / 2.0);
  const double tn = tan(45.0 * 3.14159265358979323846 / 180.0);
  a.cx = (float)(Hd / 2.0); a.cy = (float)(Wd / 2.0);
  a.fx = (float)((Hd / 2.0) / tn); a.fy = (float)((Wd / 2.0) / tn);
  a.K = (float)((double)Wd / (double)Wf);
  hipLaunchKernelGGL(bev_index_compact_kernel, dim3((unsigned)wsmg_cdiv((int64_t)Hf * Wf, CB), (unsigned)B), dim3(1024), 0, wsmg_s(stream), a,
                     clist, cnt);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_bev_scatter_max(const float* feat, const int32_t* lin_idx, int B, int Cf, int Hf, int Wf, int C,
                                    int E, float* out, wsmg_stream_t stream) {
  if (B <= 0 || Cf <= 0 || C <= 0 || C > Cf || E <= 0 || B > 65535) return WSMG_EINVAL;
  const int E2 = E * E;
  const size_t plane = (size_t)E2 * sizeof(unsigned);
  if (plane > 160 * 1024) return WSMG_EINVAL;
  // two planes per workgroup when that still leaves >= 2 workgroups per CU's LDS and enough workgroups
  int CG = (2 * plane <= 80 * 1024 && (int64_t)B * C >= 1024) ? 2 : 1;
  size_t lds = plane * CG;
  dim3 grid((unsigned)wsmg_cdiv(C, CG), (unsigned)B);
  const int wu = scatter_window(Cf, C);
#define WSMG_SCATTER(WU_)                                                                                                          \
  {                                                                                                                                \
    static bool attr_set = false;                                                                                                  \
    if (!attr_set) {                                                                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bev_scatter_kernel<WU_>),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                  \
      if (e != hipSuccess) return (int)e;                                                                                          \
      attr_set = true;                                                                                                             \
    }                                                                                                                              \
    hipLaunchKernelGGL(bev_scatter_kernel<WU_>, grid, dim3(1024), lds, wsmg_s(stream), feat, lin_idx, Cf, Hf * Wf, C, E2, CG, out); \
  }
  if (wu == 1) WSMG_SCATTER(1) else if (wu == 2) WSMG_SCATTER(2) else if (wu == 3) WSMG_SCATTER(3) else WSMG_SCATTER(4)
#undef WSMG_SCATTER
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_bev_rotate(const float* in, const float* heading, float sign, int B, int C, int E, float* out,
                               wsmg_stream_t stream) {
  if (B <= 0 || C <= 0 || C > 64 || E <= 1 || B > 65535) return WSMG_EINVAL;
  const int nb8 = (E + 7) / 8;
  dim3 grid((unsigned)(nb8 * nb8), (unsigned)B);
  hipLaunchKernelGGL(rotate_nchw_to_nhwc_kernel, grid, dim3(256), 0, wsmg_s(stream), in, heading, sign, C, E, out);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_map_fuse(const float* ego_rot, float* global_map, const float* gps, const float* masks, int B,
                             int C, int E, int G, float resolution, wsmg_stream_t stream) {
  if (B <= 0 || C <= 0 || C % 4 || E <= 1 || G < E || B > 65535) return WSMG_EINVAL;
  MapArgs a = map_args(B, C, E, G, resolution);
  int64_t n4 = (int64_t)G * G * C / 4;
  hipLaunchKernelGGL(map_reset_kernel, dim3(sgrid(n4, 1024), B), dim3(256), 0, wsmg_s(stream), global_map, masks, n4);
  int64_t n = (int64_t)(E + 4) * (E + 4) * (C / 4);
  hipLaunchKernelGGL(map_fuse_kernel, dim3(sgrid(n), B), dim3(256), 0, wsmg_s(stream), ego_rot, global_map, gps, a);
  WSMG_RETURN_LAUNCH();
}

static int bev_scatter_rotate_impl(const float* feat, const int32_t* lin_idx, const uint32_t* clist, const int32_t* cnt, const float* heading,
                                   float sign, int B, int Cf, int Hf, int Wf, int C, int E, float* out_planes, wsmg_stream_t stream) {
  if (B <= 0 || Cf <= 0 || C <= 0 || C > Cf || E <= 1 || B > 65535) return WSMG_EINVAL;
  if (clist && (!cnt || (int64_t)Hf * Wf > 65536 || (int64_t)E * E > 65536)) return WSMG_EINVAL;
  const int nblk = (int)wsmg_cdiv((int64_t)Hf * Wf, CB);
  const int E2 = E * E;
  const size_t plane = (size_t)E2 * sizeof(unsigned);
  if (plane > 160 * 1024) return WSMG_EINVAL;
  int CG = (2 * plane <= 80 * 1024 && (int64_t)B * C >= 1024) ? 2 : 1;
  dim3 grid((unsigned)wsmg_cdiv(C, CG), (unsigned)B);
  const int wu = scatter_window(Cf, C);
  // the base-coordinate table rides in whatever LDS the planes leave (E = 200: 3 840 spare bytes of the 160 KB, 800 needed)
  const int table = plane * CG + (size_t)E * sizeof(float) <= 160 * 1024 ? 1 : 0;
  const size_t lds = plane * CG + (table ? (size_t)E * sizeof(float) : 0);
  const unsigned magicE = (unsigned)((1ull << 32) / (unsigned)E + 1);      // p / E == umulhi(p, magicE) for p < E * E <= 40 960
#define WSMG_SCATTER(WU_)                                                                                                          \
  {                                                                                                                                \
    static bool attr_set = false;                                                                                                  \
    if (!attr_set) {                                                                                                               \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bev_scatter_rotate_kernel<WU_>),                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                  \
      if (e != hipSuccess) return (int)e;                                                                                          \
      attr_set = true;                                                                                                             \
    }                                                                                                                              \
    hipLaunchKernelGGL(bev_scatter_rotate_kernel<WU_>, grid, dim3(1024), lds, wsmg_s(stream), feat, lin_idx, heading, sign,        \
                       Cf, Hf * Wf, C, E, CG, magicE, table, out_planes, clist, cnt, nblk);                                        \
  }
  if (wu == 1) WSMG_SCATTER(1) else if (wu == 2) WSMG_SCATTER(2) else if (wu == 3) WSMG_SCATTER(3) else WSMG_SCATTER(4)
#undef WSMG_SCATTER
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_bev_scatter_rotate(const float* feat, const int32_t* lin_idx, const float* heading, float sign, int B, int Cf,
                                       int Hf, int Wf, int C, int E, float* out_planes, wsmg_stream_t stream) {
  return bev_scatter_rotate_impl(feat, lin_idx, nullptr, nullptr, heading, sign, B, Cf, Hf, Wf, C, E, out_planes, stream);
}
// the same from wsmg_bev_index_compact's list of valid sources (bit-identical planes)
extern "C" int wsmg_bev_scatter_rotate_compact(const float* feat, const uint32_t* clist, const int32_t* cnt, const float* heading, float sign,
                                               int B, int Cf, int Hf, int Wf, int C, int E, float* out_planes, wsmg_stream_t stream) {
  if (!clist || !cnt) return WSMG_EINVAL;
  return bev_scatter_rotate_impl(feat, nullptr, clist, cnt, heading, sign, B, Cf, Hf, Wf, C, E, out_planes, stream);
}

extern "C" int wsmg_map_fuse_planes(const float* ego_rot_planes, float* global_map, const float* gps, const float* masks, int B,
                                    int C, int E, int G, float resolution, wsmg_stream_t stream) {
  if (B <= 0 || C <= 0 || C % 4 || C > 64 || E <= 1 || G < E || B > 65535) return WSMG_EINVAL;
  MapArgs a = map_args(B, C, E, G, resolution);
  int64_t n4 = (int64_t)G * G * C / 4;
  hipLaunchKernelGGL(map_reset_kernel, dim3(sgrid(n4, 1024), B), dim3(256), 0, wsmg_s(stream), global_map, masks, n4);
  const int tiles_x = wsmg_cdiv(E + 4, FP), tiles_y = E + 4;
  const size_t lds = (size_t)FP * (C + 4) * sizeof(float);
  hipLaunchKernelGGL(map_fuse_planes_kernel, dim3((unsigned)(tiles_x * tiles_y), (unsigned)B), dim3(256), lds, wsmg_s(stream),
                     ego_rot_planes, global_map, gps, a, tiles_x);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_map_retrieve_fused(const float* global_map, const float* gps, const float* compass, int B, int C, int E, int G,
                                       float resolution, float* out, wsmg_stream_t stream) {
  if (B <= 0 || C <= 0 || C % 4 || E <= 1 || G < E || B > 65535) return WSMG_EINVAL;
  MapArgs a = map_args(B, C, E, G, resolution);
  int64_t n = (int64_t)E * E * (C / 4);
  hipLaunchKernelGGL(map_retrieve_fused_kernel, dim3(sgrid(n), B), dim3(256), 0, wsmg_s(stream), global_map, gps, compass, a, out);
  WSMG_RETURN_LAUNCH();
}

// channel slices of wsmg_map_retrieve_tiled: the fewest that keep a slice at <= 40 channels and divide C / 4 (a box of 14 x 14 + 16
// pixels x 40 channels and the records are 40 KB: four workgroups per CU); 0: none does with a box that fits LDS
static int retrieve_slices(int C) {
  const int C4 = C / 4;
  for (int n = 1; n <= C4; ++n)
    if (C4 % n == 0 && C4 / n <= 10) return n;
  return 0;
}

extern "C" int wsmg_map_retrieve_tiled(const float* global_map, const float* gps, const float* compass, int B, int C, int E, int G,
                                       float resolution, float* out, wsmg_stream_t stream) {
  if (B <= 0 || C <= 0 || C % 4 || E <= 1 || G < E || B > 65535) return WSMG_EINVAL;
  const int nsplit = retrieve_slices(C);
  if (nsplit <= 0) return WSMG_EINVAL;
  const int S4 = C / 4 / nsplit;
  const size_t lds = (size_t)(RBOX * RBOX + RBOX + 2) * S4 * 16;
  MapArgs a = map_args(B, C, E, G, resolution);
  const int tiles = (E + RT - 1) / RT;
  if ((int64_t)tiles * tiles * nsplit > 0x7fffffff) return WSMG_EINVAL;
  // i / S4 as a multiply-high: exact while i * S4 < 2^32 (i < 16 * 14 * S4)
  const unsigned magic = 0xFFFFFFFFu / (unsigned)S4 + 1u;
  static unsigned* trace_dev = nullptr;
  constexpr bool tracing = false;   // diagnostic build: true = phase stamps on stderr (synchronises)
  if (tracing && !trace_dev && hipMalloc((void**)&trace_dev, 64 * 8 * sizeof(unsigned)) != hipSuccess) trace_dev = nullptr;
  const int nblk = tiles * tiles * nsplit;
  hipLaunchKernelGGL(map_retrieve_tiled_kernel, dim3((unsigned)nblk, (unsigned)B), dim3(256), lds, wsmg_s(stream),
                     global_map, gps, compass, a, tiles, nsplit, S4, magic, out, tracing ? trace_dev : nullptr);
  if (tracing && trace_dev) {  // diagnostic only: synchronises
    unsigned h[64 * 8];
    const int n = nblk / 64 < 64 ? nblk / 64 : 64;
    if (n > 0 && hipStreamSynchronize(wsmg_s(stream)) == hipSuccess && hipMemcpy(h, trace_dev, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      double m[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < 5; ++j) m[j] += h[i * 8 + j] / (double)n;
      fprintf(stderr, "map_retrieve_tiled trace (cycles since the workgroup started, mean of %d): records %.0f, extents met %.0f, box "
                      "issued %.0f, box in LDS %.0f, end %.0f\n", n, m[0], m[1], m[2], m[3], m[4]);
    }
  }
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_map_retrieve(const float* global_map, const float* gps, const float* compass, int B, int C, int E,
                                 int G, float resolution, float* scratch, float* out, wsmg_stream_t stream) {
  if (B <= 0 || C <= 0 || C % 4 || E <= 1 || G < E || B > 65535) return WSMG_EINVAL;
  MapArgs a = map_args(B, C, E, G, resolution);
  int64_t n = (int64_t)E * E * (C / 4);
  hipLaunchKernelGGL(map_crop_kernel, dim3(sgrid(n), B), dim3(256), 0, wsmg_s(stream), global_map, gps, a, scratch);
  hipLaunchKernelGGL(rotate_nhwc_kernel, dim3(sgrid(n), B), dim3(256), 0, wsmg_s(stream), scratch, compass, 1.0f, C, E,
                     out);
  WSMG_RETURN_LAUNCH();
}
