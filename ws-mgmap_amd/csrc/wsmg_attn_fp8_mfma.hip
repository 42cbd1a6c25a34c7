// BASELINE configs[4] on the matrix cores: cross-attention with OCP e4m3 storage where the contraction IS a matrix product.
//
// The update path attends T x N rows over N shared instruction sets (mg_map_policy.py:173-178,229-232: `_attn(q, k, v, mask)`,
// row b uses the keys / values of instruction inverse[b]): per set u the R_u queries that use it form Q_u [R_u x 256] and
//     S_u = Q_u K_u^T  [R_u x L]         (the contraction over the 256 channels)
//     A_u = softmax((S_u - 1e8 mask) / 16)
//     O_u = A_u V_u    [R_u x 256]       (the contraction over the L tokens)
// are real GEMMs, unlike the single-query form of wsmg_attn_fp8.hip (a GEMV per row).  Storage: q, k, v as e4m3 bytes with one
// float scale per tensor.
//
//   * S on v_mfma_f32_32x32x16_fp8_fp8: both operands are the stored bytes (8 consecutive channels of a query / of a token per
//     lane: 8-byte loads straight from the row-major tensors), products exact, float32 accumulation.
//   * softmax in float32 (LDS-resident 32 x L tile, 8 lanes per row).
//   * O on v_mfma_f32_32x32x16_bf16: the attention weights as a bf16 PAIR (hi = bf16(p), lo = bf16(p - hi): 16 significant bits),
//     the values converted e4m3 -> bf16 exactly (every e4m3 number is a bf16 number).  An e4m3 attention weight would carry
//     6 % rounding error per token; the pair costs a second MFMA on a 32 x L x 256 product.
//
// One workgroup = one set and one tile of 32 of its rows (rows are addressed through a list grouped by set).  On inputs that are
// exactly e4m3 numbers the result equals the reference's float32 `_attn` up to summation order — golden g5f.
#include <stdio.h>

#include "wsmg_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
constexpr int AC = 256;          // channels
constexpr int LMAX = 224;        // tokens per set at most (a multiple of 32; the reference pads instructions to 200)

struct F8mArgs {
  const uint8_t* q;      // [B][256] e4m3
  const uint8_t* k;      // [U][L][256]
  const uint8_t* v;      // [U][L][256]
  const float* q_scale;  // device scalars
  const float* k_scale;
  const float* v_scale;
  const int* lengths;    // [U] valid tokens per set, or null
  const int* row_ids;    // [B] row indices grouped by set
  const int* set_start;  // [U + 1]
  float scale;           // 1 / 16
  int B, U, L;
  float* out;            // [B][256]
  float* attn;           // [B][L]
};

__device__ __forceinline__ float e4m3_to_f32(uint8_t c) {
  return __builtin_amdgcn_cvt_f32_fp8((int)c, 0);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
  bf16_t b = (bf16_t)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }

__global__ __launch_bounds__(256) void attn_fp8_mfma_fwd_kernel(F8mArgs a) {
  __shared__ __attribute__((aligned(16))) float S[32][LMAX + 4];
  __shared__ __attribute__((aligned(16))) uint8_t V8[2][32][AC + 16];      // two chunks of 32 tokens x 256 value bytes (+16: rows 4 banks apart)
  __shared__ int rows[32];
  const int u = blockIdx.x, tile = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int s0 = a.set_start[u], nrows_set = a.set_start[u + 1] - s0;
  const int first = tile * 32;
  if (first >= nrows_set) return;
  const int nr = nrows_set - first < 32 ? nrows_set - first : 32;
  if (tid < 32) rows[tid] = tid < nr ? a.row_ids[s0 + first + tid] : -1;
  __syncthreads();
  const int len = a.lengths ? a.lengths[u] : a.L;
  const int LP = (a.L + 31) & ~31;
  const float sq = *a.q_scale, sk = *a.k_scale, sv = *a.v_scale;

  // ---- S = Q K^T on the fp8 matrix pipe.  A: lane (r, h) = query r, channels 16 ks + 8 h .. + 7 (8 bytes); B: token r likewise.
  const int my_row = rows[r];
  const uint8_t* qp = a.q + (size_t)(my_row < 0 ? 0 : my_row) * AC + 8 * h;
  long qa[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) qa[ks] = my_row < 0 ? 0l : *reinterpret_cast<const long*>(qp + 16 * ks);
  for (int tt = wave; tt * 32 < LP; tt += 4) {
    const int tok = tt * 32 + r;
    const uint8_t* kp = a.k + ((size_t)u * a.L + (tok < a.L ? tok : 0)) * AC + 8 * h;
    // Four independent accumulators over 64 channels each, added in float32 afterwards: the matrix pipe aligns the 16 products of
    // an instruction to its accumulator and TRUNCATES what falls below (measured: one 256-deep chain sat 3e-5 from the reference
    // on logits of magnitude 700, the error all of one sign); four short chains and three rounded adds cut that four-fold.
    f32x16 acc4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc4[c][g] = 0.f;
    long kb[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) kb[ks] = tok < a.L ? *reinterpret_cast<const long*>(kp + 16 * ks) : 0l;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc4[ks >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(qa[ks], kb[ks], acc4[ks >> 2], 0, 0, 0);
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = (acc4[0][g] + acc4[1][g]) + (acc4[2][g] + acc4[3][g]);
    // D: lane (r = token column, h), register g -> query row (g & 3) + 8 (g >> 2) + 4 h
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int m = (g & 3) + 8 * (g >> 2) + 4 * h;
      // the reference's (q . k - 1e8 mask) / 16 in float32; tokens past L do not exist (weight exactly 0)
      float lg = acc[g] * (sq * sk);
      if (tok >= len) lg -= 1e8f;
      S[m][tok] = tok < a.L ? lg * a.scale : -INFINITY;
    }
  }
  __syncthreads();

  // ---- softmax over the tokens: 8 lanes per row, float32; attention weights to global, P stays in LDS
  {
    const int row = tid >> 3, sub = tid & 7;
    // (rows of the tile that hold no query — 24 of 32 at B = 64 over 8 sets — are skipped: their logits are never read as results, and
    //  the softmax is 3.7 of the kernel's 30 us there)
    if (rows[row] >= 0) {
    float mx = -INFINITY;
    for (int l = sub; l < LP; l += 8) mx = fmaxf(mx, S[row][l]);
    mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    float sum = 0.f;
    for (int l = sub; l < LP; l += 8) {
      const float e = expf(S[row][l] - mx);
      S[row][l] = e;
      sum += e;
    }
    sum += __shfl_xor(sum, 4, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 1, 64);
    const float inv = 1.f / sum;
    const int gr = rows[row];
    for (int l = sub; l < LP; l += 8) {
      const float p = S[row][l] * inv;
      S[row][l] = p;
      if (gr >= 0 && l < a.L) a.attn[(size_t)gr * a.L + l] = p;
    }
    }
  }
  __syncthreads();

  // ---- O = P V on the bf16 matrix pipe, P as a (hi, lo) bf16 pair.  Wave w owns channels 64 w .. 64 w + 63 (two 32-wide tiles).
  // The values arrive in chunks of 32 tokens (8 KB of bytes): one 16-byte global load per thread and chunk, double-buffered in LDS;
  // a B fragment is 8 tokens of ONE channel — 8 byte reads down a column of the chunk (the first version read those bytes from
  // global memory: 160 dependent-latency byte loads per lane, 20 of the kernel's 27 us at B = 64).
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) o[t][g] = 0.f;
  const int nchunk = LP / 32;
  auto stage = [&](int c, int buf) {      // tokens 32 c .. 32 c + 31 of set u -> V8[buf]; thread = (token tid / 8, 32-byte piece... 16 B each, 2 passes)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int idx = tid + 256 * half;            // 512 pieces of 16 bytes
      const int tk = idx >> 4, piece = idx & 15;
      const int l = 32 * c + tk;
      u32x4v val = {0u, 0u, 0u, 0u};
      if (l < a.L) val = *reinterpret_cast<const u32x4v*>(a.v + ((size_t)u * a.L + l) * AC + 16 * piece);
      *reinterpret_cast<u32x4v*>(&V8[buf][tk][16 * piece]) = val;
    }
  };
  stage(0, 0);
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk) stage(c + 1, (c + 1) & 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ks = 2 * c + kk;
      bf16x8 phi, plo;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float p = S[r][16 * ks + 8 * h + s];
        const unsigned short hi = f2bf(p);
        phi[s] = (short)hi;
        plo[s] = (short)f2bf(p - bf2f(hi));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ch = 64 * wave + 32 * t + r;
        bf16x8 vb;
#pragma unroll
        for (int s = 0; s < 8; ++s) vb[s] = (short)f2bf(e4m3_to_f32(V8[c & 1][16 * kk + 8 * h + s][ch]));   // exact: e4m3 has 4 significant bits
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(phi, vb, o[t], 0, 0, 0);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(plo, vb, o[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int m = (g & 3) + 8 * (g >> 2) + 4 * h;
      const int gr = rows[m];
      if (gr >= 0) a.out[(size_t)gr * AC + 64 * wave + 32 * t + r] = o[t][g] * sv;
    }
}

// ---- operands of one call in two launches (round 3; the host side took ~25 stock launches for them: three abs / amax /
// divide / clamp chains, three quantisations, argsort + bincount + cumsum — 185 of the op's 200 us at B = 64).
// prep 1: |x| maxima of the three tensors (non-negative floats order like their bit patterns: atomicMax on the words).
struct PrepArgs {
  const float* x[3];
  int64_t n4[3];          // float4 pieces of q, k, v
  unsigned* amax;         // [3], zero on entry
  float fixed[3];         // > 0: the caller's scale for that tensor (its maximum is not taken)
  uint8_t* codes[3];
  float* scales;          // [3] out
  const int64_t* inverse; // [B]
  int* order;             // [B] out: rows grouped by set (order inside a set is arbitrary: every row's result is its own)
  int* start;             // [U + 1] out
  int B, U;
};
__global__ __launch_bounds__(256) void attn_fp8_amax_kernel(PrepArgs a) {
  const int64_t tot = a.n4[0] + a.n4[1] + a.n4[2];
  float m[3] = {0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const int t = i < a.n4[0] ? 0 : i < a.n4[0] + a.n4[1] ? 1 : 2;
    if (a.fixed[t] > 0.f) continue;
    const int64_t k = i - (t == 0 ? 0 : t == 1 ? a.n4[0] : a.n4[0] + a.n4[1]);
    const f32x4 v = reinterpret_cast<const f32x4*>(a.x[t])[k];
    const float mm = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    // a NaN anywhere must reach the scale, as torch's amax propagates it (fmaxf drops NaNs: test the elements)
    const bool bad = v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3];
    m[t] = (bad || m[t] != m[t]) ? __builtin_nanf("") : fmaxf(m[t], mm);
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float v = m[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float w = __shfl_xor(v, o, 64);
      v = (v != v || w != w) ? __builtin_nanf("") : fmaxf(v, w);
    }
    m[t] = v;
  }
  // one atomic per workgroup and tensor (one per wave was 5 120 same-address atomics at U = 64, L = 160: 200 us of serialisation)
  __shared__ float wm[4][3];
  if ((threadIdx.x & 63) == 0) { wm[threadIdx.x >> 6][0] = m[0]; wm[threadIdx.x >> 6][1] = m[1]; wm[threadIdx.x >> 6][2] = m[2]; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = wm[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float o = wm[w][threadIdx.x];
      v = (v != v || o != o) ? __builtin_nanf("") : fmaxf(v, o);
    }
    if (v > 0.f || v != v) atomicMax(a.amax + threadIdx.x, v != v ? 0x7fc00000u : __float_as_uint(v));
  }
}
// prep 2: the three quantisations (scale = max(amax * float32(1 / 448), 1e-30) — the value `(x.abs().amax() / 448.0).clamp_min(1e-30)`
// has in torch, which multiplies by the reciprocal of a scalar divisor — or the caller's; codes as wsmg_quantize_e4m3_dev makes them) and,
// in one extra workgroup, the grouping of the rows by instruction set.
__global__ __launch_bounds__(256) void attn_fp8_quant_group_kernel(PrepArgs a, int qblocks) {
  __shared__ int cnt[1025];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x == qblocks) {     // grouping: histogram, exclusive prefix, placement
    for (int u = tid; u <= a.U; u += 256) cnt[u] = 0;
    __syncthreads();
    auto set_of = [&](int b) {        // (an index outside [0, U) is a caller error; clamped so that it cannot leave the arrays)
      const int64_t u = a.inverse[b];
      return (int)(u < 0 ? 0 : u >= a.U ? a.U - 1 : u);
    };
    for (int b = tid; b < a.B; b += 256) atomicAdd(&cnt[set_of(b) + 1], 1);
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int u = 0; u <= a.U; ++u) { run += cnt[u]; cnt[u] = run; a.start[u] = run; }   // cnt[u] = first slot of set u
    }
    __syncthreads();
    for (int b = tid; b < a.B; b += 256) a.order[atomicAdd(&cnt[set_of(b)], 1)] = b;
    if (tid < 3) {
      const float am = __uint_as_float(a.amax[tid]);
      a.scales[tid] = a.fixed[tid] > 0.f ? a.fixed[tid] : (am != am ? am : fmaxf(am * (1.0f / 448.0f), 1e-30f));
    }
    return;
  }
  const int64_t tot = a.n4[0] + a.n4[1] + a.n4[2];
  const int64_t i = (int64_t)blockIdx.x * 256 + tid;
  if (i >= tot) return;
  const int t = i < a.n4[0] ? 0 : i < a.n4[0] + a.n4[1] ? 1 : 2;
  const int64_t k = i - (t == 0 ? 0 : t == 1 ? a.n4[0] : a.n4[0] + a.n4[1]);
  const float am = __uint_as_float(a.amax[t]);
  const float sc = a.fixed[t] > 0.f ? a.fixed[t] : (am != am ? am : fmaxf(am * (1.0f / 448.0f), 1e-30f));
  const float inv_scale = (float)(1.0 / (double)sc);
  f32x4 v = reinterpret_cast<const f32x4*>(a.x[t])[k];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j] * inv_scale, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
  reinterpret_cast<unsigned*>(a.codes[t])[k] = (unsigned)w;
}

// ---- round 5: the whole operator in ONE launch (it was |x| maxima + quantise / group + this kernel = 36 us at
// B = 64, of which the attention itself 15).  From float32 q / k / v and `inverse`:
//   phase A (skipped when the caller fixes all three scales): every workgroup — the U x tiles attention workgroups and `nhelp` helper
//     workgroups that do nothing else — takes a slice of the three tensors, one atomicMax per workgroup and tensor into this
//     launch's slot of a persistent workspace, then one arrival on a monotonic counter; the attention workgroups spin (bounded) until
//     all have arrived.  Only the arrivals are waited for, never a workgroup's residency: helpers run and leave, so the launch
//     cannot deadlock on co-residency as long as the attention workgroups (<= 128: the host's condition) do not fill the GPU.
//   phase B: the rows of this workgroup's (set, tile) from a ranked scan of `inverse` (rank = position among the set's rows in row
//     order); no row list in memory.
//   phase C: attn_fp8_mfma_fwd_kernel's arithmetic, operands quantised on the fly with the quantiser's exact expression — the codes,
//     and so every result bit, equal the three-launch route's (tests/test_gpu_round5.py).
struct F8fArgs {
  const float* q;          // [B][256]
  const float* k;          // [U][L][256]
  const float* v;          // [U][L][256]
  const int64_t* inverse;  // [B]
  const int* lengths;      // [U] or null
  float fixed[3];          // > 0: the caller's scale of q / k / v
  unsigned* host_status;   // the process-wide persistent-kernel status word (host-mapped): bit 16 = this kernel's barrier timed out
  float scale;
  int B, U, L, tiles, nmain;
  unsigned* ws;            // [0..2] / [4..6]: |x| maxima of the launch with epoch parity 0 / 1; [8]: arrivals (monotonic)
  unsigned target;         // value of ws[8] once every workgroup of THIS launch has arrived
  int parity, need_amax;
  float* scales_out;       // [3] or null: the scales used (diagnostics / tests)
  float* out;
  float* attn;
  float* trace;            // diagnostic build (`tracing` in the launcher): microseconds since the start of workgroup 0 at its phase boundaries
};

__device__ __forceinline__ long quant8(const float* p, float inv_scale) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = fminf(fmaxf(x[j] * inv_scale, -448.f), 448.f);
  int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w0, true);
  int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[4], x[5], 0, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[6], x[7], w1, true);
  return (long)(((unsigned long)(unsigned)w1 << 32) | (unsigned long)(unsigned)w0);
}
__device__ __forceinline__ float scale_of(float am, float fixed) {
  return fixed > 0.f ? fixed : (am != am ? am : fmaxf(am * (1.0f / 448.0f), 1e-30f));
}

// quantiser's expression on 8 values already in registers (the codes wsmg_attn_fp8_prep writes)
__device__ __forceinline__ long quant8r(const f32x4 a, const f32x4 b, float inv_scale) {
  float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = fminf(fmaxf(x[j] * inv_scale, -448.f), 448.f);
  int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], w0, true);
  int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[4], x[5], 0, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x[6], x[7], w1, true);
  return (long)(((unsigned long)(unsigned)w1 << 32) | (unsigned long)(unsigned)w0);
}

// Order inside an attention workgroup (second form of this kernel; the first — maxima, barrier, then every operand loaded and
// quantised where it was needed — took 32 us at B = 64, as long as the three launches it replaced: the value chunks alone were five
// dependent load -> convert -> LDS -> barrier rounds, 12 us): the rows of the (set, tile) first — they need `inverse` only —, then ALL
// float32 loads that do not depend on a scale are ISSUED (the tile's queries, the set's values: up to 350 registers per lane, one
// wave per SIMD has 512), then the maxima pass and the grid barrier run while those loads are in flight; after the barrier the
// registers are quantised (queries to MFMA operands, values to the LDS image of all chunks at once) and only the keys are loaded
// behind it.  The arithmetic per element is unchanged.
constexpr int NCHMAX = LMAX / 32;      // 7 value chunks of 32 tokens at most

__global__ __launch_bounds__(256) void attn_fp8_mfma_fused_kernel(F8fArgs a) {
  // (dynamic: 29 KB of logits + 60 KB of value codes are more than the 64 KB a kernel may declare statically without an attribute)
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
  float (*S)[LMAX + 4] = reinterpret_cast<float (*)[LMAX + 4]>(fused_lds);
  uint8_t (*V8)[32][AC + 16] = reinterpret_cast<uint8_t (*)[32][AC + 16]>(fused_lds + 32 * (LMAX + 4) * 4);
  __shared__ int rows[32];
  __shared__ int wtot[4];
  __shared__ float wm[4][3];
  __shared__ float sc3[3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const unsigned long long t_start = wall_clock64();
  auto stamp = [&](int i) { if (a.trace && wg == 0 && tid == 0) a.trace[i] = (float)(wall_clock64() - t_start) * 0.01f; };
  const bool main_wg = wg < a.nmain;
  const int u = main_wg ? wg % a.U : 0, tile = main_wg ? wg / a.U : 0;
  const int first = tile * 32;
  const int r = lane & 31, h = lane >> 5;
  const int LP = (a.L + 31) & ~31;
  const int nchunk = LP / 32;
  // ---- the rows of this workgroup's (set, tile): rank = position among the set's rows in row order
  int running = 0;
  if (main_wg) {
    if (tid < 32) rows[tid] = -1;
    __syncthreads();
    for (int base = 0; base < a.B; base += 256) {
      const int b = base + tid;
      bool mine = false;
      if (b < a.B) {
        const int64_t su = a.inverse[b];
        mine = (int)(su < 0 ? 0 : su >= a.U ? a.U - 1 : su) == u;
      }
      const unsigned long long bal = __ballot(mine);
      const int before = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wtot[wave] = __popcll(bal);
      __syncthreads();
      int off = running;
      for (int w = 0; w < wave; ++w) off += wtot[w];
      const int rank = off + before;
      if (mine && rank >= first && rank < first + 32) rows[rank - first] = b;
      running += wtot[0] + wtot[1] + wtot[2] + wtot[3];
      __syncthreads();
    }
  }
  const bool active = main_wg && first < running;      // (uniform) this workgroup has rows to attend for
  stamp(0);
  // ---- issue the scale-independent loads: this lane's query channels, this thread's pieces of the set's values
  f32x4 qraw[16][2];
  f32x4 vraw[NCHMAX][2][4];
  const int my_row = active ? rows[r] : -1;
  if (active) {
    const float* qp = a.q + (size_t)(my_row < 0 ? 0 : my_row) * AC + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      qraw[ks][0] = *reinterpret_cast<const f32x4*>(qp + 16 * ks);
      qraw[ks][1] = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
    }
#pragma unroll
    for (int c = 0; c < NCHMAX; ++c)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int idx = tid + 256 * half;
        const int tk = idx >> 4, piece = idx & 15;
        const int l = 32 * c + tk;
        const bool ok = c < nchunk && l < a.L;
        const float* vp = a.v + ((size_t)u * a.L + (ok ? l : 0)) * AC + 16 * piece;
#pragma unroll
        for (int j = 0; j < 4; ++j) vraw[c][half][j] = ok ? *reinterpret_cast<const f32x4*>(vp + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  stamp(1);
  // ---- phase A: the |x| maxima (every workgroup) and the grid barrier (attention workgroups)
  if (a.need_amax) {
    const int64_t n4q = (int64_t)a.B * AC / 4, n4k = (int64_t)a.U * a.L * AC / 4;
    const int64_t tot = n4q + 2 * n4k;
    float m[3] = {0.f, 0.f, 0.f};
    constexpr int UA = 4;
    for (int64_t i0 = (int64_t)wg * 256 + tid; i0 < tot; i0 += (int64_t)gridDim.x * 256 * UA) {
      f32x4 vv[UA];
      int tt[UA];
#pragma unroll
      for (int uu = 0; uu < UA; ++uu) {
        const int64_t i = i0 + (int64_t)uu * gridDim.x * 256;
        const int t = i < n4q ? 0 : i < n4q + n4k ? 1 : 2;
        tt[uu] = i < tot ? t : -1;
        const int64_t kk = i - (t == 0 ? 0 : t == 1 ? n4q : n4q + n4k);
        const float* src = t == 0 ? a.q : t == 1 ? a.k : a.v;
        vv[uu] = i < tot ? reinterpret_cast<const f32x4*>(src)[kk] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int uu = 0; uu < UA; ++uu) {
        const f32x4 v = vv[uu];
        const float mm = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        const bool bad = v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (tt[uu] == t && !(a.fixed[t] > 0.f)) m[t] = (bad || m[t] != m[t]) ? __builtin_nanf("") : fmaxf(m[t], mm);
      }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      float v = m[t];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o, 64);
        v = (v != v || w != w) ? __builtin_nanf("") : fmaxf(v, w);
      }
      m[t] = v;
    }
    if (lane == 0) { wm[wave][0] = m[0]; wm[wave][1] = m[1]; wm[wave][2] = m[2]; }
    __syncthreads();
    unsigned* const slot = a.ws + 4 * a.parity;
    if (tid < 3) {
      float v = wm[0][tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float o = wm[w][tid];
        v = (v != v || o != o) ? __builtin_nanf("") : fmaxf(v, o);
      }
      if (v > 0.f || v != v) __hip_atomic_fetch_max(slot + tid, v != v ? 0x7fc00000u : __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_fetch_add(a.ws + 8, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(2);
    if (!main_wg) return;           // a helper: done
    if (!active && wg != 0) return; // no rows for this (set, tile): it has arrived, nothing else to do (workgroup 0 cleans up below)
    if (tid == 0) {
      unsigned spins = 0;
      bool ok = true;
      while ((int)(__hip_atomic_load(a.ws + 8, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - a.target) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 24)) { ok = false; break; }     // (seconds: a workgroup of this launch never ran — results become NaN)
      }
      // a timeout is REPORTED, not only visible as NaN: the host's next status check raises
      if (!ok && a.host_status) __hip_atomic_fetch_or(a.host_status, 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float am = __uint_as_float(__hip_atomic_load(slot + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        sc3[t] = ok ? scale_of(am, a.fixed[t]) : __builtin_nanf("");
      }
      if (wg == 0) {       // the other slot is the next launch's: hand it over clean (launches on one stream do not overlap)
        unsigned* const other = a.ws + 4 * (a.parity ^ 1);
        other[0] = 0u; other[1] = 0u; other[2] = 0u;
        if (a.scales_out) { a.scales_out[0] = sc3[0]; a.scales_out[1] = sc3[1]; a.scales_out[2] = sc3[2]; }
      }
    }
    __syncthreads();
  } else {
    if (!main_wg) return;
    if (tid < 3) sc3[tid] = a.fixed[tid];
    if (wg == 0 && tid < 3 && a.scales_out) a.scales_out[tid] = a.fixed[tid];
    __syncthreads();
  }
  if (!active) return;
  stamp(3);
  const float sq = sc3[0], sk = sc3[1], sv = sc3[2];
  const float iq = (float)(1.0 / (double)sq), ik = (float)(1.0 / (double)sk), iv = (float)(1.0 / (double)sv);
  const int len = a.lengths ? a.lengths[u] : a.L;

  // ---- quantise what is in registers: the queries to MFMA operands, the values to the LDS image of all chunks
  long qa[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) qa[ks] = my_row < 0 ? 0l : quant8r(qraw[ks][0], qraw[ks][1], iq);
#pragma unroll
  for (int c = 0; c < NCHMAX; ++c) {
    if (c < nchunk) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int idx = tid + 256 * half;
        const int tk = idx >> 4, piece = idx & 15;
        const int l = 32 * c + tk;
        u32x4v val = {0u, 0u, 0u, 0u};
        if (l < a.L) {
          const long lo = quant8r(vraw[c][half][0], vraw[c][half][1], iv), hi = quant8r(vraw[c][half][2], vraw[c][half][3], iv);
          val[0] = (unsigned)(lo & 0xffffffffl); val[1] = (unsigned)((unsigned long)lo >> 32);
          val[2] = (unsigned)(hi & 0xffffffffl); val[3] = (unsigned)((unsigned long)hi >> 32);
        }
        *reinterpret_cast<u32x4v*>(&V8[c][tk][16 * piece]) = val;
      }
    }
  }

  stamp(4);
  // ---- S = Q K^T on the fp8 matrix pipe (the keys are loaded and quantised here)
  for (int tt = wave; tt * 32 < LP; tt += 4) {
    const int tok = tt * 32 + r;
    const float* kp = a.k + ((size_t)u * a.L + (tok < a.L ? tok : 0)) * AC + 8 * h;
    f32x16 acc4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc4[c][g] = 0.f;
    long kb[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) kb[ks] = tok < a.L ? quant8(kp + 16 * ks, ik) : 0l;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) acc4[ks >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(qa[ks], kb[ks], acc4[ks >> 2], 0, 0, 0);
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = (acc4[0][g] + acc4[1][g]) + (acc4[2][g] + acc4[3][g]);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int m = (g & 3) + 8 * (g >> 2) + 4 * h;
      float lg = acc[g] * (sq * sk);
      if (tok >= len) lg -= 1e8f;
      S[m][tok] = tok < a.L ? lg * a.scale : -INFINITY;
    }
  }
  __syncthreads();          // S and the value image are complete
  stamp(5);
  {
    const int row = tid >> 3, sub = tid & 7;
    // (rows of the tile that hold no query — 24 of 32 at B = 64 over 8 sets — are skipped: their logits are never read as results, and
    //  the softmax is 3.7 of the kernel's 30 us there)
    if (rows[row] >= 0) {
    float mx = -INFINITY;
    for (int l = sub; l < LP; l += 8) mx = fmaxf(mx, S[row][l]);
    mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    float sum = 0.f;
    for (int l = sub; l < LP; l += 8) {
      const float e = expf(S[row][l] - mx);
      S[row][l] = e;
      sum += e;
    }
    sum += __shfl_xor(sum, 4, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 1, 64);
    const float inv = 1.f / sum;
    const int gr = rows[row];
    for (int l = sub; l < LP; l += 8) {
      const float p = S[row][l] * inv;
      S[row][l] = p;
      if (gr >= 0 && l < a.L) a.attn[(size_t)gr * a.L + l] = p;
    }
    }
  }
  __syncthreads();
  stamp(6);
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) o[t][g] = 0.f;
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ks = 2 * c + kk;
      bf16x8 phi, plo;
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const float p = S[r][16 * ks + 8 * h + s_];
        const unsigned short hi = f2bf(p);
        phi[s_] = (short)hi;
        plo[s_] = (short)f2bf(p - bf2f(hi));
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ch = 64 * wave + 32 * t + r;
        bf16x8 vb;
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) vb[s_] = (short)f2bf(e4m3_to_f32(V8[c][16 * kk + 8 * h + s_][ch]));
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(phi, vb, o[t], 0, 0, 0);
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(plo, vb, o[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int m = (g & 3) + 8 * (g >> 2) + 4 * h;
      const int gr = rows[m];
      if (gr >= 0) a.out[(size_t)gr * AC + 64 * wave + 32 * t + r] = o[t][g] * sv;
    }
  stamp(7);
}

}  // namespace

extern "C" int wsmg_attn_fp8_prep(const float* q, const float* k_sets, const float* v_sets, const int64_t* inverse, int B, int U, int L,
                                  int C, float q_scale, float k_scale, float v_scale, uint8_t* q_codes, uint8_t* k_codes,
                                  uint8_t* v_codes, float* scales, int* row_ids, int* set_start, unsigned* amax_ws,
                                  wsmg_stream_t stream) {
  if (!q || !k_sets || !v_sets || !inverse || !q_codes || !k_codes || !v_codes || !scales || !row_ids || !set_start || !amax_ws)
    return WSMG_EINVAL;
  if (B <= 0 || U <= 0 || U > 1024 || L <= 0 || C <= 0 || (C & 3)) return WSMG_EINVAL;
  PrepArgs a;
  a.x[0] = q; a.x[1] = k_sets; a.x[2] = v_sets;
  a.n4[0] = (int64_t)B * C / 4; a.n4[1] = a.n4[2] = (int64_t)U * L * C / 4;
  a.amax = amax_ws;
  a.fixed[0] = q_scale; a.fixed[1] = k_scale; a.fixed[2] = v_scale;
  a.codes[0] = q_codes; a.codes[1] = k_codes; a.codes[2] = v_codes;
  a.scales = scales; a.inverse = inverse; a.order = row_ids; a.start = set_start; a.B = B; a.U = U;
  const int64_t tot = a.n4[0] + a.n4[1] + a.n4[2];
  if (!(q_scale > 0.f && k_scale > 0.f && v_scale > 0.f)) {
    int64_t g = wsmg_cdiv(tot, 256 * 4);
    if (g > 512) g = 512;
    hipLaunchKernelGGL(attn_fp8_amax_kernel, dim3((unsigned)g), dim3(256), 0, wsmg_s(stream), a);
  }
  const int64_t qb = wsmg_cdiv(tot, 256);
  if (qb >= (1ll << 31) - 1) return WSMG_EINVAL;
  hipLaunchKernelGGL(attn_fp8_quant_group_kernel, dim3((unsigned)qb + 1), dim3(256), 0, wsmg_s(stream), a, (int)qb);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_attn_fp8_mfma_fwd(const uint8_t* q_codes, const float* q_scale, const uint8_t* k_codes, const float* k_scale,
                                      const uint8_t* v_codes, const float* v_scale, const int* lengths, const int* row_ids,
                                      const int* set_start, float scale, int B, int U, int L, int C, float* out, float* attn,
                                      wsmg_stream_t stream) {
  if (!q_codes || !q_scale || !k_codes || !k_scale || !v_codes || !v_scale || !row_ids || !set_start || !out || !attn) return WSMG_EINVAL;
  if (B <= 0 || U <= 0 || L <= 0 || L > LMAX || C != AC) return WSMG_EINVAL;
  F8mArgs a{q_codes, k_codes, v_codes, q_scale, k_scale, v_scale, lengths, row_ids, set_start, scale, B, U, L, out, attn};
  hipLaunchKernelGGL(attn_fp8_mfma_fwd_kernel, dim3((unsigned)U, (unsigned)wsmg_cdiv(B, 32)), dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

// co-residency budget of the fused launch's grid barrier, from the device's own CU count
static int fused_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  return cus;
}
static int fused_main_limit() { const int h = fused_cus() / 2; return h < 128 ? h : 128; }
static int fused_total_limit() { const int t = fused_cus() - 32; return t < 224 ? (t > 1 ? t : 1) : 224; }

extern "C" int wsmg_attn_fp8_mfma_fused(const float* q, const float* k_sets, const float* v_sets, const int64_t* inverse, const int* lengths,
                                        float q_scale, float k_scale, float v_scale, float scale, int B, int U, int L, int C,
                                        unsigned* workspace, unsigned arrivals_before, int epoch, float* scales_out, float* out, float* attn,
                                        wsmg_stream_t stream) {
  if (!q || !k_sets || !v_sets || !inverse || !workspace || !out || !attn) return WSMG_EINVAL;
  if (B <= 0 || U <= 0 || U > 1024 || L <= 0 || L > LMAX || C != AC) return WSMG_EINVAL;
  const int tiles = (int)wsmg_cdiv(B, 32);
  const int64_t nmain = (int64_t)U * tiles;
  const int need = !(q_scale > 0.f && k_scale > 0.f && v_scale > 0.f);
  // the spinning workgroups (one per CU: ~90 KB of LDS each) must leave room for the helpers — half the CUs THIS device has (a
  // partitioned or CU-masked device has fewer than 256), never more than 128 (see the kernel)
  if (need && nmain > fused_main_limit()) return WSMG_EINVAL;
  if (nmain > (1 << 20)) return WSMG_EINVAL;
  int nhelp = 0;
  if (need) {     // ~8 KB of float32 per workgroup in the |x| pass, at most one workgroup per CU in all
    const int64_t tot4 = ((int64_t)B * C + 2ll * U * L * C) / 4;
    int64_t want = wsmg_cdiv(tot4, 256 * 4);
    if (want > fused_total_limit()) want = fused_total_limit();
    nhelp = want > nmain ? (int)(want - nmain) : 0;
  }
  F8fArgs a;
  a.q = q; a.k = k_sets; a.v = v_sets; a.inverse = inverse; a.lengths = lengths;
  a.fixed[0] = q_scale; a.fixed[1] = k_scale; a.fixed[2] = v_scale;
  a.host_status = wsmgi_rnn_status_dev();
  a.scale = scale; a.B = B; a.U = U; a.L = L; a.tiles = tiles; a.nmain = (int)nmain;
  a.ws = workspace; a.target = arrivals_before + (unsigned)(nmain + nhelp); a.parity = epoch & 1; a.need_amax = need;
  a.scales_out = scales_out; a.out = out; a.attn = attn;
  static float* trace_dev = nullptr;
  constexpr bool tracing = false;   // diagnostic build: true = phase stamps on stderr (synchronises)
  if (tracing && !trace_dev && hipMalloc((void**)&trace_dev, 16 * sizeof(float)) != hipSuccess) trace_dev = nullptr;
  a.trace = tracing ? trace_dev : nullptr;
  constexpr int FUSED_LDS = 32 * (LMAX + 4) * 4 + NCHMAX * 32 * (AC + 16);
  static bool attr = false;
  if (!attr) {
    hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8_mfma_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS);
    if (e0 != hipSuccess) return (int)e0;
    attr = true;
  }
  hipLaunchKernelGGL(attn_fp8_mfma_fused_kernel, dim3((unsigned)(nmain + nhelp)), dim3(256), FUSED_LDS, wsmg_s(stream), a);
  if (a.trace) {      // diagnostic only: synchronises
    float h[8];
    if (hipStreamSynchronize(wsmg_s(stream)) == hipSuccess && hipMemcpy(h, a.trace, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "fp8 fused trace (us since workgroup 0 started): rows %.2f, loads issued %.2f, arrived %.2f, barrier passed %.2f, quantised %.2f, "
                      "S done %.2f, softmax done %.2f, end %.2f\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

/* workgroups wsmg_attn_fp8_mfma_fused launches for this shape when it has to take the maxima itself (= the arrivals one launch adds
 * to its workspace's counter); 0 when it would refuse the shape */
extern "C" int wsmg_attn_fp8_mfma_fused_arrivals(int B, int U, int L, int C) {
  if (B <= 0 || U <= 0 || U > 1024 || L <= 0 || L > LMAX || C != AC) return 0;
  const int64_t nmain = (int64_t)U * wsmg_cdiv(B, 32);
  if (nmain > fused_main_limit()) return 0;
  const int64_t tot4 = ((int64_t)B * C + 2ll * U * L * C) / 4;
  int64_t want = wsmg_cdiv(tot4, 256 * 4);
  if (want > fused_total_limit()) want = fused_total_limit();
  return (int)(want > nmain ? want : nmain);
}
