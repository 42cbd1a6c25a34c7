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

}  // namespace

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
