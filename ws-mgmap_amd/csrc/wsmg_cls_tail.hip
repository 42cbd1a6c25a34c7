// The tail of the semantic classifier (mg_map_policy.py:78-86, 191-199; policy.py:61-66 of the reference) as ONE pass per direction
// over its 32-channel, 2S x 2S (48 x 48) activations:
//
//     BatchNorm(batch statistics) + ReLU  ->  Conv2d 1 x 1 (32 -> 27)  ->  { cross-entropy against the nearest-resized semantic
//     ground truth (the prediction monitor),  AvgPool2d(2) (the input of map_classified_linear),  the logits themselves }
//
// At B = 512 each of these tensors is 75 MB in bf16 and every step is HBM-bound: BatchNorm apply (read + write), the 1 x 1
// convolution (read + write, 48-72 TFLOP/s of a 2 500 TFLOP/s engine: 2.4 GFLOP on 150 MB), the cross-entropy (read), the
// average pool (read) — 470 MB forward; backward the loss gradient, the pool's gradient and their sum, the convolution's
// backward-data and weight gradient and BatchNorm's reduction: 900 MB.  Here the forward reads the convolution output once
// (75 MB) and writes the logits and the pooled map (94 MB); the backward reads it once more with the pooled gradient (94 MB) and
// writes BatchNorm's incoming gradient (75 MB) together with the per-sample partial sums of everything that is reduced over
// pixels: BatchNorm's two per-channel sums, the 1 x 1 weight gradient, its bias gradient.
//
// One 512-thread workgroup per sample; a wave takes patches of 2 rows x 16 columns = 32 pixels (a 2 x 2 pooling window never
// leaves a patch).  The 1 x 1 convolution is two v_mfma_f32_32x32x16_bf16 per patch, computed TRANSPOSED — D[class][pixel] =
// sum_k W[class][k] a[pixel][k] — so that the B operand of a lane is 16 contiguous bytes of its pixel's NHWC channel run,
// straight from global memory, and a lane (pixel r, half h) ends up with 16 of its pixel's 32 class logits in registers: the
// soft-max is in-lane plus one exchange with lane r ^ 32.  Backward: d a = W^T d logits is the same shape with the class
// axis as K (the K slots follow the accumulator layout, so d logits feeds the MFMA without a shuffle), and the weight
// gradient contracts over the 32 pixels of the patch, both operands transposed through LDS.
//
// Tensors leave as 16-byte stores after a per-wave LDS transpose (80-byte pitch: conflict-free ds_read_b128).
// Sums over pixels are accumulated per lane, reduced over lanes, waves (in wave order) and samples (wsmg_cls_tail_finish_kernel,
// in sample order, float64): bit-identical from run to run.
#include "wsmg_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int CH = 32;          // channels of the activation = padded classes
constexpr int PITCH = 80;       // LDS tile pitch (bytes): 64 + 16
constexpr int TILE = 32 * PITCH;
constexpr int WAVES = 8;

__device__ __forceinline__ unsigned short f2bf(float f) {
  bf16_t b = (bf16_t)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ void unpack8(const u32x4 v, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    o[2 * j] = __uint_as_float(v[j] << 16);
    o[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
  }
}
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(v[j]);
  return o;
}
// class / channel index of accumulator register g in lane half h (C layout of the 32 x 32 MFMA: rows)
__device__ __forceinline__ int row_of(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

struct TailArgs {
  const bf16_t* y2;        // [B][H][W][32] convolution output (pre-BatchNorm)
  const float* gamma;      // [32]
  const float* beta;
  const float* mean;       // [32] batch mean / 1 / sqrt(var + eps) (wsmg_bn_stats_finalize)
  const float* invstd;
  const float* w6;         // [classes][32] float32 (the 1 x 1 convolution's OIHW parameter)
  const float* b6;         // [classes]
  const float* gt;         // [B][Hg][Wg] float32 semantic ground truth (class ids), or null: no loss
  int B, H, W, Hg, Wg, classes;
  float sy, sx;            // Hg / H, Wg / W in float32: torch's nearest-neighbour source index is floorf(dst * scale)
  // forward outputs
  bf16_t* sem;             // [B][H][W][32] logits (channels >= classes are 0)
  bf16_t* pooled;          // [B][H/2][W/2][32]
  float* ce_rows;          // [B] mean cross-entropy of the sample, or null
  // backward
  const float* g_rows;     // [B] gradient of ce_rows, or null
  const bf16_t* dpooled;   // [B][H/2][W/2][32] gradient of pooled, or null
  bf16_t* dbn;             // [B][H][W][32] gradient of the BatchNorm + ReLU output's PRE-ReLU value (masked)
  float* part;             // [B][PART] per-sample partial sums: BatchNorm (2 x 32), weight gradient (32 x 32), bias gradient (32)
};
constexpr int PART = 2 * CH + CH * CH + CH;

// per-channel constants live in LDS (every lane of a half reads the same address: broadcast, no conflicts); registers are
// what these kernels run out of.  cst[0] = scale = invstd * gamma, cst[1] = shift = beta - mean * scale, cst[2] = mean,
// cst[3] = invstd, cst[4] = the 1 x 1 convolution's bias (0 for the padded classes)
__device__ __forceinline__ void fill_const(const float* mean, const float* invstd, const float* gamma, const float* beta, const float* b6,
                                           int classes, float (*cst)[CH], int tid) {
  if (tid < CH) {
    const float sc = invstd[tid] * gamma[tid];
    cst[0][tid] = sc;
    cst[1][tid] = beta[tid] - mean[tid] * sc;
    cst[2][tid] = mean[tid];
    cst[3][tid] = invstd[tid];
    cst[4][tid] = tid < classes ? b6[tid] : 0.f;
  }
}
// relu(batchnorm(y2)) of this lane's two 8-channel pieces (channels 8h .. 8h+7 and 16+8h .. 16+8h+7) as the two bf16 B fragments
__device__ __forceinline__ void act_frags(const u32x4 (&raw)[2], const float (*cst)[CH], int h, bf16x8 (&frag)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float v[8];
    unpack8(raw[ks], v);
    const int k0 = 16 * ks + 8 * h;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(&cst[0][k0]), s1 = *reinterpret_cast<const f32x4*>(&cst[0][k0 + 4]);
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(&cst[1][k0]), t1 = *reinterpret_cast<const f32x4*>(&cst[1][k0 + 4]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float u0 = v[s] * s0[s] + t0[s], u1 = v[4 + s] * s1[s] + t1[s];
      v[s] = u0 > 0.f ? u0 : 0.f;
      v[4 + s] = u1 > 0.f ? u1 : 0.f;
    }
    frag[ks] = pack8(v);
  }
}
// A operand of the logits product: W6[class r][16 ks + 8 h + s]
__device__ __forceinline__ void w6_frags(const TailArgs& a, int r, int h, bf16x8 (&wa)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = r < a.classes ? a.w6[r * CH + 16 * ks + 8 * h + s] : 0.f;
    wa[ks] = pack8(v);
  }
}

struct Patch {
  int y0, x0;        // first row / column of the patch
  int64_t pix;       // global pixel index of this lane's pixel
};
__device__ __forceinline__ Patch patch_of(const TailArgs& a, int b, int p, int r) {
  const int ppr = a.W >> 4;
  const int py = p / ppr, px = p - py * ppr;
  Patch q;
  q.y0 = 2 * py; q.x0 = 16 * px;
  q.pix = ((int64_t)b * a.H + q.y0 + (r >> 4)) * a.W + q.x0 + (r & 15);
  return q;
}
__device__ __forceinline__ void load_raw(const TailArgs& a, int64_t pix, int h, u32x4 (&raw)[2]) {
  const u32x4* p = reinterpret_cast<const u32x4*>(a.y2 + pix * CH);
  raw[0] = p[h];
  raw[1] = p[2 + h];
}
__device__ __forceinline__ int label_of(const TailArgs& a, int b, const Patch& q, int r) {
  const int y = q.y0 + (r >> 4), x = q.x0 + (r & 15);
  int syi = (int)floorf((float)y * a.sy), sxi = (int)floorf((float)x * a.sx);
  syi = syi < a.Hg - 1 ? syi : a.Hg - 1;
  sxi = sxi < a.Wg - 1 ? sxi : a.Wg - 1;
  return (int)a.gt[((int64_t)b * a.Hg + syi) * a.Wg + sxi];      // .long() of the float map: truncation
}
// this wave's accumulator tile (lane = pixel, 4 consecutive channels per 8-byte piece) -> [pixel][32] bf16 at dst, 16-byte stores
__device__ __forceinline__ void store_tile(unsigned char* lds, const float (&v)[16], int r, int h, int lane, bf16_t* dst, const Patch& q,
                                           int b, int H, int W) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x2 w;
    w[0] = (unsigned)f2bf(v[4 * j]) | ((unsigned)f2bf(v[4 * j + 1]) << 16);
    w[1] = (unsigned)f2bf(v[4 * j + 2]) | ((unsigned)f2bf(v[4 * j + 3]) << 16);
    *reinterpret_cast<u32x2*>(lds + r * PITCH + 16 * j + 8 * h) = w;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int qq = lane >> 2, pc = lane & 3;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(lds + (qq + 16 * half) * PITCH + 16 * pc);
    const int64_t pix = ((int64_t)b * H + q.y0 + half) * W + q.x0 + qq;
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(dst) + pix * (CH * 2) + 16 * pc) = t;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512) void cls_tail_fwd_kernel(TailArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[WAVES * TILE];
  __shared__ float wsum[WAVES];
  __shared__ __attribute__((aligned(16))) float cst[5][CH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.x;
  unsigned char* const lds = lds_all + wave * TILE;
  fill_const(a.mean, a.invstd, a.gamma, a.beta, a.b6, a.classes, cst, tid);
  bf16x8 wa[2];
  w6_frags(a, r, h, wa);
  __syncthreads();
  const int npatch = (a.H >> 1) * (a.W >> 4);
  float loss = 0.f;
  u32x4 raw[2], nraw[2];
  if (wave < npatch) load_raw(a, patch_of(a, b, wave, r).pix, h, raw);
  for (int p = wave; p < npatch; p += WAVES) {
    const Patch q = patch_of(a, b, p, r);
    if (p + WAVES < npatch) load_raw(a, patch_of(a, b, p + WAVES, r).pix, h, nraw);
    const int tlab = a.gt ? label_of(a, b, q, r) : -1;      // (a 4-byte gather: issued before the arithmetic that hides it)
    bf16x8 fb[2];
    act_frags(raw, cst, h, fb);
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], fb[1], acc, 0, 0, 0);
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(&cst[4][8 * j + 4 * h]);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[4 * j + i] = acc[4 * j + i] + bq[i];
    }
    if (a.gt) {   // cross-entropy of this pixel: 16 classes in this lane, 16 in lane ^ 32
      float mx = -INFINITY;
#pragma unroll
      for (int g = 0; g < 16; ++g) if (row_of(g, h) < a.classes) mx = fmaxf(mx, v[g]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float s = 0.f, vt = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (row_of(g, h) < a.classes) s += __expf(v[g] - mx);
        if (row_of(g, h) == tlab) vt = v[g];
      }
      s += __shfl_xor(s, 32, 64);
      vt += __shfl_xor(vt, 32, 64);
      // a label outside [0, classes): F.cross_entropy of the reference (policy.py:61-66) faults on it; here the sample's loss row
      // becomes NaN — loud in the loss and in every gradient behind it — instead of a finite, wrong number
      if (h == 0) loss += ((unsigned)tlab < (unsigned)a.classes) ? (mx + logf(s)) - vt : __builtin_nanf("");
    }
    // the logits as the unfused path stores them (bf16), and their 2 x 2 average
    float vr[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) vr[g] = bf2f(f2bf(v[g]));
    store_tile(lds, vr, r, h, lane, a.sem, q, b, a.H, a.W);
    {
      float ps[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float t2 = vr[g] + __shfl_xor(vr[g], 1, 64);
        t2 += __shfl_xor(t2, 16, 64);
        ps[g] = t2 * 0.25f;
      }
      if ((r & 17) == 0) {     // even column of the patch's first row: one pooled pixel
        const int64_t pp = ((int64_t)b * (a.H >> 1) + (q.y0 >> 1)) * (a.W >> 1) + ((q.x0 + r) >> 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x2 w;
          w[0] = (unsigned)f2bf(ps[4 * j]) | ((unsigned)f2bf(ps[4 * j + 1]) << 16);
          w[1] = (unsigned)f2bf(ps[4 * j + 2]) | ((unsigned)f2bf(ps[4 * j + 3]) << 16);
          *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(a.pooled) + pp * (CH * 2) + 16 * j + 8 * h) = w;
        }
      }
    }
    raw[0] = nraw[0]; raw[1] = nraw[1];
  }
  if (a.ce_rows) {
    loss = wave_sum(loss);
    if (lane == 0) wsum[wave] = loss;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) t += wsum[w];
      a.ce_rows[b] = t / (float)(a.H * a.W);
    }
  }
}

// sum over the 32 lanes that share h (xor 1, 2, 4, 8, 16): every lane ends with the total
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(512) void cls_tail_bwd_kernel(TailArgs a) {
  // per wave: [0, TILE) store staging, [TILE, 2 TILE) d logits^T [class][pixel], [2 TILE, 3 TILE) activations^T [channel][pixel]
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[WAVES * 3 * TILE];
  __shared__ __attribute__((aligned(16))) float cst[5][CH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int b = blockIdx.x;
  unsigned char* const lds = lds_all + wave * 3 * TILE;
  unsigned char* const dlT = lds + TILE;
  unsigned char* const acT = lds + 2 * TILE;
  fill_const(a.mean, a.invstd, a.gamma, a.beta, a.b6, a.classes, cst, tid);
  bf16x8 wa[2], wt[2];
  w6_frags(a, r, h, wa);
  // A operand of d a = W^T d logits: W6[class of K slot][channel r]; the K slots follow the accumulator layout of d logits:
  // slice ks, slot s -> register g = 8 ks + s of the lane half h -> class row_of(g, h)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int m = row_of(8 * ks + s, h);
      v[s] = m < a.classes ? a.w6[m * CH + r] : 0.f;
    }
    wt[ks] = pack8(v);
  }
  const float coef = a.g_rows ? a.g_rows[b] / (float)(a.H * a.W) : 0.f;
  __syncthreads();
  const int npatch = (a.H >> 1) * (a.W >> 4);
  float s_dy[16], s_dyx[16], s_db[16];
  f32x16 accw;
#pragma unroll
  for (int g = 0; g < 16; ++g) { s_dy[g] = 0.f; s_dyx[g] = 0.f; s_db[g] = 0.f; accw[g] = 0.f; }
  u32x4 raw[2], nraw[2];
  if (wave < npatch) load_raw(a, patch_of(a, b, wave, r).pix, h, raw);
  for (int p = wave; p < npatch; p += WAVES) {
    const Patch q = patch_of(a, b, p, r);
    if (p + WAVES < npatch) load_raw(a, patch_of(a, b, p + WAVES, r).pix, h, nraw);
    // the loads this patch needs later — the pooled map's gradient, y2 once more in the accumulator layout (same cache lines as
    // `raw`), the label — are issued first: with one workgroup per CU nothing else hides their latency
    u32x2 dpw[4], y2w[4];
    {
      const int64_t pp = ((int64_t)b * (a.H >> 1) + (q.y0 >> 1)) * (a.W >> 1) + ((q.x0 + (r & 15)) >> 1);
      const unsigned char* src = reinterpret_cast<const unsigned char*>(a.dpooled) + pp * (CH * 2) + 8 * h;
      const unsigned char* ysrc = reinterpret_cast<const unsigned char*>(a.y2) + q.pix * (CH * 2) + 8 * h;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dpw[j] = a.dpooled ? *reinterpret_cast<const u32x2*>(src + 16 * j) : u32x2{0u, 0u};
        y2w[j] = *reinterpret_cast<const u32x2*>(ysrc + 16 * j);
      }
    }
    const int tlab = a.g_rows ? label_of(a, b, q, r) : -1;
    bf16x8 fb[2];
    act_frags(raw, cst, h, fb);
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[0], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[1], fb[1], acc, 0, 0, 0);
    // d logits = coef (softmax - onehot) + d pooled / 4
    float dl[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) dl[g] = 0.f;
    if (a.g_rows) {
      float v[16], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(&cst[4][8 * j + 4 * h]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[4 * j + i] = acc[4 * j + i] + bq[i];
          if (row_of(4 * j + i, h) < a.classes) mx = fmaxf(mx, v[4 * j + i]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) { v[g] = row_of(g, h) < a.classes ? __expf(v[g] - mx) : 0.f; s += v[g]; }
      s += __shfl_xor(s, 32, 64);
      const float inv = 1.f / s;
#pragma unroll
      for (int g = 0; g < 16; ++g)
        if (row_of(g, h) < a.classes) dl[g] = coef * (v[g] * inv - (row_of(g, h) == tlab ? 1.f : 0.f));
    }
    if (a.dpooled) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x2 w = dpw[j];
        const float d0 = __uint_as_float(w[0] << 16), d1 = __uint_as_float(w[0] & 0xffff0000u);
        const float d2 = __uint_as_float(w[1] << 16), d3 = __uint_as_float(w[1] & 0xffff0000u);
        if (row_of(4 * j, h) < a.classes) dl[4 * j] += 0.25f * d0;
        if (row_of(4 * j + 1, h) < a.classes) dl[4 * j + 1] += 0.25f * d1;
        if (row_of(4 * j + 2, h) < a.classes) dl[4 * j + 2] += 0.25f * d2;
        if (row_of(4 * j + 3, h) < a.classes) dl[4 * j + 3] += 0.25f * d3;
      }
    }
    // bf16, as the MFMA consumes it; the bias gradient sums the same rounded values
    unsigned short dlb[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) { dlb[g] = f2bf(dl[g]); s_db[g] += bf2f(dlb[g]); }
    bf16x8 bl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int s = 0; s < 8; ++s) bl[ks][s] = (short)dlb[8 * ks + s];
    f32x16 da;
#pragma unroll
    for (int g = 0; g < 16; ++g) da[g] = 0.f;
    da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[0], bl[0], da, 0, 0, 0);
    da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[1], bl[1], da, 0, 0, 0);
    // ReLU mask + BatchNorm's two sums, in the accumulator layout: channels row_of(g, h) of pixel r — re-read y2 in that layout
    float dbn[16];
    {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x2 w = y2w[j];
        const float yv[4] = {__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16),
                             __uint_as_float(w[1] & 0xffff0000u)};
        const f32x4 sc = *reinterpret_cast<const f32x4*>(&cst[0][8 * j + 4 * h]), sh = *reinterpret_cast<const f32x4*>(&cst[1][8 * j + 4 * h]);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(&cst[2][8 * j + 4 * h]), is = *reinterpret_cast<const f32x4*>(&cst[3][8 * j + 4 * h]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float xh = (yv[i] - mu[i]) * is[i];
          const bool on = yv[i] * sc[i] + sh[i] > 0.f;      // the forward's own expression: the same mask
          const float d = on ? da[4 * j + i] : 0.f;
          const float dr = bf2f(f2bf(d));          // what BatchNorm's apply pass will read back
          dbn[4 * j + i] = dr;
          s_dy[4 * j + i] += dr;
          s_dyx[4 * j + i] += dr * xh;
        }
      }
    }
    store_tile(lds, dbn, r, h, lane, a.dbn, q, b, a.H, a.W);
    // weight gradient: dW[class][k] += sum over the patch's pixels of d logits[class][pixel] act[pixel][k]; both operands K(pixel)-major
#pragma unroll
    for (int g = 0; g < 16; ++g) *reinterpret_cast<unsigned short*>(dlT + row_of(g, h) * PITCH + 2 * r) = dlb[g];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int s = 0; s < 8; ++s) *reinterpret_cast<unsigned short*>(acT + (16 * ks + 8 * h + s) * PITCH + 2 * r) = (unsigned short)fb[ks][s];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(dlT + r * PITCH + 16 * h), a1 = *reinterpret_cast<const bf16x8*>(dlT + r * PITCH + 32 + 16 * h);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(acT + r * PITCH + 16 * h), b1 = *reinterpret_cast<const bf16x8*>(acT + r * PITCH + 32 + 16 * h);
      accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, accw, 0, 0, 0);
      accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, accw, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw[0] = nraw[0]; raw[1] = nraw[1];
  }
  // ---- reductions: lanes, then waves in wave order, into this sample's partial record
  __syncthreads();
  float* const red = reinterpret_cast<float*>(lds_all);     // [WAVES][PART] (the tiles are dead)
  float* const mine = red + wave * PART;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const float t0 = half_sum(s_dy[g]), t1 = half_sum(s_dyx[g]), t2 = half_sum(s_db[g]);
    if (r == 0) {
      mine[row_of(g, h)] = t0;
      mine[CH + row_of(g, h)] = t1;
      mine[2 * CH + CH * CH + row_of(g, h)] = t2;
    }
    mine[2 * CH + row_of(g, h) * CH + r] = accw[g];      // accumulator tile: row = class, column (lane) = channel
  }
  __syncthreads();
  for (int i = tid; i < PART; i += 512) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += red[w * PART + i];
    a.part[(size_t)b * PART + i] = t;
  }
}

// sums of the per-sample partial records (float64): d gamma, d beta of the BatchNorm, dW [classes][32], db [classes].  Round 6: a
// workgroup = 16 outputs x 16 groups of samples — every thread adds its group's samples in sample order (8 loads in flight), then
// the 16 group sums of an output are added in group order: a fixed order, so bit-identical from run to run.  (One thread per output
// walking all B samples was 512 dependent-latency steps: 43 us at B = 512, on the critical path of the backward pass.)
constexpr int FIN_OUT = 16, FIN_GRP = 16;
__global__ __launch_bounds__(FIN_OUT * FIN_GRP) void cls_tail_finish_kernel(const float* __restrict__ part, int B, int classes,
                                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                          float* __restrict__ dw6, float* __restrict__ db6) {
  __shared__ double sh[FIN_GRP][FIN_OUT];
  const int o = threadIdx.x % FIN_OUT, grp = threadIdx.x / FIN_OUT;
  const int i = blockIdx.x * FIN_OUT + o;
  const int per = (B + FIN_GRP - 1) / FIN_GRP;
  const int b0 = grp * per, b1 = b0 + per < B ? b0 + per : B;
  double s = 0.0;
  if (i < PART) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(b + j) * PART + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; b < b1; ++b) s += part[(size_t)b * PART + i];
  }
  sh[grp][o] = s;
  __syncthreads();
  if (grp != 0 || i >= PART) return;
  s = 0.0;
#pragma unroll
  for (int g2 = 0; g2 < FIN_GRP; ++g2) s += sh[g2][o];
  if (i < CH) dbeta[i] = (float)s;
  else if (i < 2 * CH) dgamma[i - CH] = (float)s;
  else if (i < 2 * CH + CH * CH) { const int m = (i - 2 * CH) / CH; if (m < classes) dw6[i - 2 * CH] = (float)s; }
  else { const int m = i - 2 * CH - CH * CH; if (m < classes) db6[m] = (float)s; }
}

int check_tail(int B, int H, int W, int classes) {
  if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 15) || classes <= 0 || classes > CH) return WSMG_EINVAL;
  if ((int64_t)B * H * W >= (1ll << 31)) return WSMG_EINVAL;
  return 0;
}

}  // namespace

extern "C" long long wsmg_cls_tail_workspace_floats(int B) { return B > 0 ? (long long)B * PART : 0; }

extern "C" int wsmg_cls_tail_fwd_bf16(const void* y2, const float* gamma, const float* beta, const float* mean, const float* invstd,
                                      const float* w6, const float* b6, int classes, const float* gt, int Hg, int Wg, int B, int H, int W,
                                      void* sem, void* pooled, float* ce_rows, wsmg_stream_t stream) {
  if (int e = check_tail(B, H, W, classes)) return e;
  if (!y2 || !gamma || !beta || !mean || !invstd || !w6 || !b6 || !sem || !pooled) return WSMG_EINVAL;
  if ((gt != nullptr) != (ce_rows != nullptr) || (gt && (Hg <= 0 || Wg <= 0))) return WSMG_EINVAL;
  TailArgs a{};
  a.y2 = (const bf16_t*)y2; a.gamma = gamma; a.beta = beta; a.mean = mean; a.invstd = invstd; a.w6 = w6; a.b6 = b6; a.gt = gt;
  a.B = B; a.H = H; a.W = W; a.Hg = Hg; a.Wg = Wg; a.classes = classes;
  a.sy = gt ? (float)Hg / (float)H : 0.f; a.sx = gt ? (float)Wg / (float)W : 0.f;
  a.sem = (bf16_t*)sem; a.pooled = (bf16_t*)pooled; a.ce_rows = ce_rows;
  hipLaunchKernelGGL(cls_tail_fwd_kernel, dim3((unsigned)B), dim3(512), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_cls_tail_bwd_bf16(const void* y2, const float* gamma, const float* beta, const float* mean, const float* invstd,
                                      const float* w6, const float* b6, int classes, const float* gt, int Hg, int Wg, const float* g_rows,
                                      const void* dpooled, int B, int H, int W, void* dbn, float* workspace, long long workspace_floats,
                                      float* dgamma, float* dbeta, float* dw6, float* db6, wsmg_stream_t stream) {
  if (int e = check_tail(B, H, W, classes)) return e;
  if (!y2 || !gamma || !beta || !mean || !invstd || !w6 || !b6 || !dbn || !workspace || !dgamma || !dbeta || !dw6 || !db6) return WSMG_EINVAL;
  if (workspace_floats < (long long)B * PART) return WSMG_ENOMEM;
  if (g_rows && (!gt || Hg <= 0 || Wg <= 0)) return WSMG_EINVAL;
  TailArgs a{};
  a.y2 = (const bf16_t*)y2; a.gamma = gamma; a.beta = beta; a.mean = mean; a.invstd = invstd; a.w6 = w6; a.b6 = b6; a.gt = gt;
  a.B = B; a.H = H; a.W = W; a.Hg = Hg; a.Wg = Wg; a.classes = classes;
  a.sy = gt ? (float)Hg / (float)H : 0.f; a.sx = gt ? (float)Wg / (float)W : 0.f;
  a.g_rows = g_rows; a.dpooled = (const bf16_t*)dpooled; a.dbn = (bf16_t*)dbn; a.part = workspace;
  hipLaunchKernelGGL(cls_tail_bwd_kernel, dim3((unsigned)B), dim3(512), 0, wsmg_s(stream), a);
  hipLaunchKernelGGL(cls_tail_finish_kernel, dim3((PART + FIN_OUT - 1) / FIN_OUT), dim3(FIN_OUT * FIN_GRP), 0, wsmg_s(stream), workspace, B,
                     classes, dgamma, dbeta, dw6, db6);
  WSMG_RETURN_LAUNCH();
}
