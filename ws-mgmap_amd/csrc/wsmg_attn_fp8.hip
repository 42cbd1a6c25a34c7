// fp8 (OCP e4m3) text attention, BASELINE configs[4] ("CMA cross-attention in fp8 MFMA, instruction len=160, batch=64"),
// forward AND backward.  Reference arithmetic: the state -> instruction attention of the policy,
// mg_map_policy.py:126-127 (state_text_k_layer = Conv1d(256, 256, 1)), :173-178 (_attn), :229-232 (call site):
//
//   k_l = W_k x_l + b_k ;  logits_l = (q . k_l - 1e8 mask_l) / 16 ;  a = softmax(logits) ;  out = sum_l a_l x_l
//
// with x (the instruction embedding, keys' input AND values) stored as e4m3 bytes + one float scale.
//
// Decomposition used here (same design as the bf16 map attention, wsmg_attn.hip):
//   * ONE query per row makes q . (W_k x_l + b_k) = (W_k^T q) . x_l + q . b_k, and the last term is the same for every token,
//     so it cancels in the softmax: the key projection — 99 % of the operator's FLOPs in the reference — collapses into
//     the [B,256] x [256,256] product q_f = q W_k.  That product is the operator's only real contraction and runs on the
//     matrix cores (attn_fp8_fold_kernel: v_mfma_f32_16x16x4_f32, exact float32 products — the query stays float32:
//     an e4m3 query would put 6 % noise on every logit for no bandwidth gain, the query is 1 KB per row).
//   * the rest is GEMV-shaped (1 flop per token byte): HBM / latency bound.  At B = 64 one workgroup per row is 64
//     workgroups on 256 CUs, so a row is SPLIT over NS workgroups (token chunks): each reads its chunk once into LDS,
//     takes local softmax statistics and a partial weighted sum, and the LAST workgroup of a row to finish (agent-scope
//     release / acquire ticket, cdna_hip_programming.md Guideline 16) combines the NS partials.
//   * backward (attn_fp8_bwd_kernel): one pass over the row's bytes gives d logits, d q_f and the straight-through
//     gradient of the de-quantised tokens; dq = d q_f W_k^T and dW_k = q^T d q_f are the same MFMA kernel / one GEMM.
#include <stdlib.h>

#include "wsmg_common.h"

namespace {

constexpr int AC = 256;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 cvt4_e4m3(unsigned w) {
  auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
  auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// ----------------------------------------------------------------------------- fold: out[B][256] = in[B][256] @ W  (or W^T)
// W [256][256] row-major (Conv1d weight [C_out][C_in]).  transpose = 0: out[b][j] = sum_o in[b][o] W[o][j]   (q_f = q W_k);
// transpose = 1: out[b][o] = sum_j in[b][j] W[o][j]   (dq = d q_f W_k^T).  One workgroup = 16 rows x 64 columns, one wave =
// one 16 x 16 tile over K = 256 in 64 steps of v_mfma_f32_16x16x4_f32 (float32 in, float32 accumulate: bit-for-bit an fmaf
// chain per output).
__global__ __launch_bounds__(256) void attn_fp8_fold_kernel(const float* __restrict__ in, const float* __restrict__ w, int B,
                                                            int transpose, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 16, c0 = blockIdx.y * 64 + wave * 16;
  const int lr = lane & 15, lk = lane >> 4;
  const int row = r0 + lr;
  const float* ap = in + (size_t)(row < B ? row : 0) * AC + lk;
  const float amask = row < B ? 1.f : 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // all 128 operand loads of the wave are issued before the first MFMA (the launch is a latency chain, not a bandwidth
  // problem: with 8 k-steps in flight it took 14.6 us for B = 64)
  float av[AC / 4], bv[AC / 4];
#pragma unroll
  for (int ks = 0; ks < AC / 4; ++ks) {
    const int k = 4 * ks + lk;
    av[ks] = ap[4 * ks];
    bv[ks] = transpose ? w[(size_t)(c0 + lr) * AC + k] : w[(size_t)k * AC + c0 + lr];
  }
#pragma unroll
  for (int ks = 0; ks < AC / 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks] * amask, bv[ks], acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int orow = r0 + 4 * lk + j;
    if (orow < B) out[(size_t)orow * AC + c0 + lr] = acc[j];
  }
}

// ----------------------------------------------------------------------------- forward, split over the token axis
constexpr int F8_CHUNK = 224;         // tokens per workgroup at most (56 KB of dynamic LDS: a whole row when the batch fills the chip)
constexpr int F8_PART = AC + 2;       // partial record: 256 weighted sums, local max, local sum

struct Fp8FwdArgs {
  const float* qf;        // [B][256]  q W_k
  const uint8_t* x;       // [B][L][256] e4m3
  const float* x_scale;   // device scalar: real value = byte value * x_scale
  const int* lengths;     // [B] or null
  float scale;            // 1/16
  int L, chunk, ns;
  float* out;             // [B][256]
  float* attn;            // [B][L]
  float* part;            // [B][ns][F8_PART] scratch
  unsigned* ticket;       // [B], zero before the first launch; the last workgroup of a row resets its word
};

// dot of this 16-lane group's token with the query: lane holds 16 channels of the query; x bytes from LDS
__device__ __forceinline__ float dot16(const uint8_t* tok, const float (&qv)[16], int l16) {
  const u32x4 raw = *reinterpret_cast<const u32x4*>(tok + l16 * 16);
  float d = 0.f;
#pragma unroll
  for (int w4 = 0; w4 < 4; ++w4) {
    const f32x4 v = cvt4_e4m3(raw[w4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) d = fmaf(qv[4 * w4 + j], v[j], d);
  }
  // 16-lane butterfly (xor 8, 4, 2, 1): DPP row operations
  d += __shfl_xor(d, 8, 64);
  d += __shfl_xor(d, 4, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 1, 64);
  return d;
}

__global__ __launch_bounds__(256) void attn_fp8_fwd_kernel(Fp8FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t xs[];   // chunk * 256 token bytes
  __shared__ float lg[F8_CHUNK];
  __shared__ __attribute__((aligned(16))) float psum[4][AC];
  __shared__ float red[8];
  __shared__ int is_last;
  const int sp = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, grp = lane >> 4;
  const int t0 = sp * a.chunk;
  const int nt = (a.L - t0) < a.chunk ? (a.L - t0) : a.chunk;   // tokens of this chunk (>= 1 by construction)
  const int len = a.lengths ? a.lengths[b] : a.L;
  const float xsc = *a.x_scale;
  // the only HBM read of the tokens: 16 bytes per thread and trip
  const u32x4* src = reinterpret_cast<const u32x4*>(a.x + ((size_t)b * a.L + t0) * AC);
  for (int i = tid; i < nt * (AC / 16); i += 256) reinterpret_cast<u32x4*>(xs)[i] = src[i];
  float qv[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.qf + (size_t)b * AC + l16 * 16)[j];
    qv[4 * j] = v[0]; qv[4 * j + 1] = v[1]; qv[4 * j + 2] = v[2]; qv[4 * j + 3] = v[3];
  }
  __syncthreads();
  // ---- logits: 16 lanes per token, 16 tokens per trip of the workgroup
  for (int i0 = 0; i0 < nt; i0 += 16) {
    const int i = i0 + wave * 4 + grp;
    float d = dot16(xs + (size_t)(i < nt ? i : nt - 1) * AC, qv, l16);
    if (i < nt && l16 == 0) {
      d = d * xsc;
      if (t0 + i >= len) d = d - 1e8f;          // the reference's additive mask (mg_map_policy.py:175)
      lg[i] = d * a.scale;
    }
  }
  __syncthreads();
  // ---- local softmax statistics (nt <= 64: one wave's worth)
  float v = tid < nt ? lg[tid] : -INFINITY;
  float mx = wave_max(v);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float e = tid < nt ? expf(v - mx) : 0.f;
  if (tid < nt && a.ns > 1)   // raw scaled logit: the row's last workgroup normalises
    __hip_atomic_store(&a.attn[(size_t)b * a.L + t0 + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float sm = wave_sum(e);
  if (lane == 0) red[4 + wave] = sm;
  if (tid < nt) lg[tid] = e;
  __syncthreads();
  sm = red[4] + red[5] + red[6] + red[7];
  // ---- partial weighted sum: lane = 4 channels, wave w takes tokens w, w + 4, ...
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < nt; i += 4) {
    const float w = lg[i];
    const f32x4 xv = cvt4_e4m3(reinterpret_cast<const unsigned*>(xs + (size_t)i * AC)[lane]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fmaf(w, xv[j], acc[j]);
  }
  reinterpret_cast<f32x4*>(&psum[wave][0])[lane] = acc;
  __syncthreads();
  const float mine = psum[0][tid] + psum[1][tid] + psum[2][tid] + psum[3][tid];
  if (a.ns == 1) {   // the whole row is here: normalise and leave
    const float inv1 = 1.f / sm;
    a.out[(size_t)b * AC + tid] = mine * inv1 * xsc;
    if (tid < nt) a.attn[(size_t)b * a.L + tid] = lg[tid] * inv1;
    return;
  }
  float* pr = a.part + ((size_t)b * a.ns + sp) * F8_PART;
  __hip_atomic_store(&pr[tid], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    __hip_atomic_store(&pr[AC], mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&pr[AC + 1], sm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // (agent-scope stores / loads for everything that crosses workgroups: write-through, so no release fence — an L2
  // write-back per workgroup — and no acquire are needed: cdna_hip_programming.md Guideline 16, the sc1 form.  With the fence
  // pair the B = 4096 launch took 395 us against 110 us for one workgroup per row.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(&a.ticket[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t == (unsigned)a.ns - 1u);
    if (is_last) __hip_atomic_store(&a.ticket[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
  }
  __syncthreads();
  if (!is_last) return;
  const float* pb = a.part + (size_t)b * a.ns * F8_PART;
  float M = -INFINITY;
  for (int s = 0; s < a.ns; ++s) M = fmaxf(M, __hip_atomic_load(&pb[s * F8_PART + AC], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  float S = 0.f, o = 0.f;
  for (int s = 0; s < a.ns; ++s) {
    const float f = expf(__hip_atomic_load(&pb[s * F8_PART + AC], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - M);
    S = fmaf(__hip_atomic_load(&pb[s * F8_PART + AC + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), f, S);
    o = fmaf(__hip_atomic_load(&pb[s * F8_PART + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), f, o);
  }
  const float inv = 1.f / S;
  a.out[(size_t)b * AC + tid] = o * inv * xsc;
  for (int i = tid; i < a.L; i += 256) {
    float* ap = a.attn + (size_t)b * a.L + i;
    *ap = expf(__hip_atomic_load(ap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - M) * inv;
  }
}

// ----------------------------------------------------------------------------- forward in ONE launch: fold + attention per row
// BASELINE configs[4] as SURVEY 8d defines it (B = 64 rows, each with its OWN L = 160 token set: 82 KB per row): rounds 3-5 ran it as
// the fold launch + the split forward launch (18.9 us).  Here one workgroup of 512 threads owns a row from the float32 query to the
// context: (1) the row's token bytes are requested first and stay in flight in registers; (2) the query fold q_f = q W_k — a
// [1,256] x [256,256] product per row, W_k (256 KB) read through L2 — as two float32 fmaf chains per output column (threads j and
// 256 + j take the two halves of the reduction); (3) the tokens land in LDS, the logits, the softmax and the weighted sum are
// attn_fp8_fwd_kernel's (whole row, ns = 1), with eight waves instead of four.  No second launch, no workspace, no ticket.
constexpr int ROW_MAX_L = 224;        // 56 KB of dynamic LDS for the row's bytes
struct Fp8RowArgs {
  const float* q;         // [B][256]
  const float* w;         // [256][256] Conv1d weight [C_out][C_in]: q_f[j] = sum_o q[o] w[o][j]
  const uint8_t* x;       // [B][L][256] e4m3
  const float* x_scale;
  const int* lengths;     // [B] or null
  float scale;
  int L;
  float* qf;              // [B][256] or null (the backward pass wants it)
  float* out;             // [B][256]
  float* attn;            // [B][L]
};

__global__ __launch_bounds__(512) void attn_fp8_row_kernel(Fp8RowArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t xs[];   // L * 256 token bytes
  __shared__ float lg[ROW_MAX_L];
  __shared__ __attribute__((aligned(16))) float qs[AC];
  __shared__ __attribute__((aligned(16))) float qfs[2][AC];
  __shared__ __attribute__((aligned(16))) float psum[8][AC];
  __shared__ float red[16];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, grp = lane >> 4;
  const int L = a.L;
  const int len = a.lengths ? a.lengths[b] : L;
  // (1) the row's bytes: up to 7 x 16 bytes per thread, in flight while the fold runs
  constexpr int TRIPS = ROW_MAX_L * (AC / 16) / 512;
  const u32x4* src = reinterpret_cast<const u32x4*>(a.x + (size_t)b * L * AC);
  const int npiece = L * (AC / 16);
  u32x4 tok[TRIPS];
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int i = tid + 512 * t;
    tok[t] = i < npiece ? src[i] : u32x4{0u, 0u, 0u, 0u};
  }
  if (tid < AC) qs[tid] = a.q[(size_t)b * AC + tid];
  const float xsc = *a.x_scale;
  __syncthreads();
  // (2) fold: column j, reduction half `hf`
  {
    const int j = tid & (AC - 1), hf = tid >> 8;
    const float* wp = a.w + (size_t)(hf * (AC / 2)) * AC + j;
    float acc = 0.f;
#pragma unroll 1
    for (int o0 = 0; o0 < AC / 2; o0 += 16) {
      float wv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wv[u] = wp[(size_t)(o0 + u) * AC];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fmaf(qs[hf * (AC / 2) + o0 + u], wv[u], acc);
    }
    qfs[hf][j] = acc;
  }
#pragma unroll
  for (int t = 0; t < TRIPS; ++t) {
    const int i = tid + 512 * t;
    if (i < npiece) reinterpret_cast<u32x4*>(xs)[i] = tok[t];
  }
  __syncthreads();
  if (tid < AC) {
    const float v = qfs[0][tid] + qfs[1][tid];
    qfs[0][tid] = v;
    if (a.qf) a.qf[(size_t)b * AC + tid] = v;
  }
  __syncthreads();
  float qv[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = reinterpret_cast<const f32x4*>(&qfs[0][l16 * 16])[j];
    qv[4 * j] = v[0]; qv[4 * j + 1] = v[1]; qv[4 * j + 2] = v[2]; qv[4 * j + 3] = v[3];
  }
  // (3) logits: 16 lanes per token, 32 tokens per trip of the workgroup
  for (int i0 = 0; i0 < L; i0 += 32) {
    const int i = i0 + wave * 4 + grp;
    float d = dot16(xs + (size_t)(i < L ? i : L - 1) * AC, qv, l16);
    if (i < L && l16 == 0) {
      d = d * xsc;
      if (i >= len) d = d - 1e8f;          // the reference's additive mask (mg_map_policy.py:175)
      lg[i] = d * a.scale;
    }
  }
  __syncthreads();
  const float v = tid < L ? lg[tid] : -INFINITY;
  float mx = wave_max(v);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
  const float e = tid < L ? expf(v - mx) : 0.f;
  float sm = wave_sum(e);
  if (lane == 0) red[8 + wave] = sm;
  if (tid < L) lg[tid] = e;
  __syncthreads();
  sm = ((red[8] + red[9]) + (red[10] + red[11])) + ((red[12] + red[13]) + (red[14] + red[15]));
  f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < L; i += 8) {
    const float w = lg[i];
    const f32x4 xv = cvt4_e4m3(reinterpret_cast<const unsigned*>(xs + (size_t)i * AC)[lane]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc4[j] = fmaf(w, xv[j], acc4[j]);
  }
  reinterpret_cast<f32x4*>(&psum[wave][0])[lane] = acc4;
  __syncthreads();
  const float inv1 = 1.f / sm;
  if (tid < AC) {
    const float mine = ((psum[0][tid] + psum[1][tid]) + (psum[2][tid] + psum[3][tid])) + ((psum[4][tid] + psum[5][tid]) + (psum[6][tid] + psum[7][tid]));
    a.out[(size_t)b * AC + tid] = mine * inv1 * xsc;
  }
  if (tid < L) a.attn[(size_t)b * L + tid] = lg[tid] * inv1;
}

// ----------------------------------------------------------------------------- backward (one workgroup per row)
constexpr int F8_MAX_L = 224;   // 56 KB of LDS for the row's bytes
struct Fp8BwdArgs {
  const float* qf;       // [B][256]
  const uint8_t* x;      // [B][L][256]
  const float* x_scale;
  const float* attn;     // [B][L] saved weights
  const float* dout;     // [B][256]
  const float* dattn;    // [B][L] or null
  float scale;
  int L;
  float* dqf;            // [B][256]
  float* dx;             // [B][L][256]  gradient of the DE-QUANTISED tokens (straight-through), or null
};

__global__ __launch_bounds__(256) void attn_fp8_bwd_kernel(Fp8BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t xs[];   // L * 256 bytes
  __shared__ float dl[F8_MAX_L];
  __shared__ __attribute__((aligned(16))) float psum[4][AC];
  __shared__ float red[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, grp = lane >> 4;
  const float xsc = *a.x_scale;
  const u32x4* src = reinterpret_cast<const u32x4*>(a.x + (size_t)b * a.L * AC);
  for (int i = tid; i < a.L * (AC / 16); i += 256) reinterpret_cast<u32x4*>(xs)[i] = src[i];
  float dv[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.dout + (size_t)b * AC + l16 * 16)[j];
    dv[4 * j] = v[0]; dv[4 * j + 1] = v[1]; dv[4 * j + 2] = v[2]; dv[4 * j + 3] = v[3];
  }
  __syncthreads();
  // da_i = dout . x_i (+ dattn_i);  s = sum_i a_i da_i
  float part = 0.f;
  for (int i0 = 0; i0 < a.L; i0 += 16) {
    const int i = i0 + wave * 4 + grp;
    float d = dot16(xs + (size_t)(i < a.L ? i : a.L - 1) * AC, dv, l16);
    if (i < a.L && l16 == 0) {
      d = d * xsc + (a.dattn ? a.dattn[(size_t)b * a.L + i] : 0.f);
      dl[i] = d;
      part = fmaf(a.attn[(size_t)b * a.L + i], d, part);
    }
  }
  part = wave_sum(part);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  const float s = red[0] + red[1] + red[2] + red[3];
  for (int i = tid; i < a.L; i += 256) dl[i] = a.attn[(size_t)b * a.L + i] * (dl[i] - s) * a.scale;   // d(q_f . x_i) incl. 1/16
  __syncthreads();
  // d q_f = x_scale * sum_i dl_i code_i ;  dx_i = dl_i q_f + a_i dout   (lane = 4 channels, wave w takes tokens w, w+4, ...)
  const f32x4 qf4 = reinterpret_cast<const f32x4*>(a.qf + (size_t)b * AC)[lane];
  const f32x4 do4 = reinterpret_cast<const f32x4*>(a.dout + (size_t)b * AC)[lane];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < a.L; i += 4) {
    const float g = dl[i];
    const f32x4 xv = cvt4_e4m3(reinterpret_cast<const unsigned*>(xs + (size_t)i * AC)[lane]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = fmaf(g, xv[j], acc[j]);
    if (a.dx) {
      const float ai = a.attn[(size_t)b * a.L + i];
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaf(g, qf4[j], ai * do4[j]);
      reinterpret_cast<f32x4*>(a.dx + ((size_t)b * a.L + i) * AC)[lane] = o;
    }
  }
  reinterpret_cast<f32x4*>(&psum[wave][0])[lane] = acc;
  __syncthreads();
  a.dqf[(size_t)b * AC + tid] = (psum[0][tid] + psum[1][tid] + psum[2][tid] + psum[3][tid]) * xsc;
}

// float32 -> e4m3 (round to nearest even, saturating at +-448), 4 values per thread; inverse scale from device memory
__global__ __launch_bounds__(256) void quantize_e4m3_dev_kernel(const float* __restrict__ x, int64_t n4, const float* __restrict__ scale,
                                                                unsigned* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float inv_scale = (float)(1.0 / (double)*scale);   // as the host quantiser: float32(1 / scale) from a float64 quotient
  f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j] * inv_scale, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
  y[i] = (unsigned)w;
}

}  // namespace

extern "C" int wsmg_attn_fp8_fold(const float* in, const float* w, int B, int C, int transpose, float* out, wsmg_stream_t stream) {
  if (C != AC || B <= 0) return WSMG_EINVAL;
  hipLaunchKernelGGL(attn_fp8_fold_kernel, dim3((unsigned)wsmg_cdiv(B, 16), 4), dim3(256), 0, wsmg_s(stream), in, w, B, transpose, out);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_attn_fp8_splits(int B, int L) {
  // enough workgroups for the chip (256 CUs) while a chunk keeps >= 32 tokens; a chunk holds at most F8_CHUNK
  int ns = 1;
  while (ns < 8 && (int64_t)B * ns < 256 && wsmg_cdiv(L, ns * 2) >= 32) ns *= 2;
  while (wsmg_cdiv(L, ns) > F8_CHUNK && ns < 64) ++ns;
  return ns;
}

extern "C" int64_t wsmg_attn_fp8_workspace_bytes(int B, int L) {
  return ((int64_t)B * wsmg_attn_fp8_splits(B, L) * F8_PART + 64) * 4;
}

extern "C" int wsmg_attn_fp8_fwd(const float* q_folded, const uint8_t* x_e4m3, const float* x_scale, const int* lengths, float scale,
                                 int B, int L, int C, float* out, float* attn, void* workspace, unsigned* ticket,
                                 wsmg_stream_t stream) {
  if (C != AC || B <= 0 || L <= 0 || B > 65535 || !workspace || !ticket) return WSMG_EINVAL;
  if (wsmg_cdiv(L, wsmg_attn_fp8_splits(B, L)) > F8_CHUNK) return WSMG_EINVAL;
  const int ns = wsmg_attn_fp8_splits(B, L);
  const int chunk = (int)wsmg_cdiv(L, ns);
  Fp8FwdArgs a{q_folded, x_e4m3, x_scale, lengths, scale, L, chunk, (int)wsmg_cdiv(L, chunk), out, attn, (float*)workspace, ticket};
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       F8_CHUNK * AC);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(attn_fp8_fwd_kernel, dim3((unsigned)a.ns, (unsigned)B), dim3(256), (size_t)chunk * AC, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_attn_fp8_row_fwd(const float* q, const float* w_k, const uint8_t* x_e4m3, const float* x_scale, const int* lengths,
                                     float scale, int B, int L, int C, float* q_folded, float* out, float* attn, wsmg_stream_t stream) {
  if (C != AC || B <= 0 || L <= 0 || L > ROW_MAX_L || !q || !w_k || !x_e4m3 || !x_scale || !out || !attn) return WSMG_EINVAL;
  Fp8RowArgs a{q, w_k, x_e4m3, x_scale, lengths, scale, L, q_folded, out, attn};
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fp8_row_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       ROW_MAX_L * AC);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  hipLaunchKernelGGL(attn_fp8_row_kernel, dim3((unsigned)B), dim3(512), (size_t)L * AC, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_attn_fp8_bwd(const float* q_folded, const uint8_t* x_e4m3, const float* x_scale, const float* attn,
                                 const float* dout, const float* dattn, float scale, int B, int L, int C, float* dq_folded, float* dx,
                                 wsmg_stream_t stream) {
  if (C != AC || B <= 0 || L <= 0 || L > F8_MAX_L) return WSMG_EINVAL;
  Fp8BwdArgs a{q_folded, x_e4m3, x_scale, attn, dout, dattn, scale, L, dq_folded, dx};
  hipLaunchKernelGGL(attn_fp8_bwd_kernel, dim3((unsigned)B), dim3(256), (size_t)L * AC, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_quantize_e4m3_dev(const float* x, int64_t n, const float* scale, uint8_t* y, wsmg_stream_t stream) {
  if (n <= 0 || (n & 3) != 0) return WSMG_EINVAL;
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(quantize_e4m3_dev_kernel, dim3((unsigned)wsmg_cdiv(n4, 256)), dim3(256), 0, wsmg_s(stream), x, n4, scale,
                     reinterpret_cast<unsigned*>(y));
  WSMG_RETURN_LAUNCH();
}
