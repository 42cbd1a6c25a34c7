// Unique rows of an instruction-token matrix in ONE launch.
//
// A teacher-forcing batch repeats every episode's instruction over its T time steps (the reference encodes all T x N rows:
// instruction_encoder.py:68-93, called from mg_map_policy.py:182); the policy encodes each distinct row once.  Finding the distinct
// rows was ~25 stock launches (row hash, 1-D unique = sort + scan + scatter, gather, compare, two host read-backs) — 0.5 ms of
// dependent small kernels in front of the ONE host synchronisation of a forward pass, during which the host cannot enqueue
// anything: with the update's recurrent core pipelined over streams the host was left with < 1 ms of lead over the GPU there.
// Here: one workgroup.
//   1. a wave per row: 64-bit polynomial hash of the row (coalesced reads), hashes to LDS;
//   2. a thread per row b: the first row j <= b with the same hash AND the same tokens is its representative (the scan over
//      hashes is LDS broadcast reads; the token comparison runs only on a hash match);
//   3. block prefix sum over "b is its own representative" -> rank; inverse[b] = rank[rep[b]];
//   4. representatives copy their row to uniq[rank] and count its non-zero tokens.
// Unique rows come out in order of first appearance.  meta[0] = U, meta[1] = longest length, meta[2 + u] = length of row u: the
// one small read-back the caller needs (the packed LSTM is launched over U sequences of at most meta[1] steps).
#include "wsmg_common.h"

namespace {

constexpr int DD_THREADS = 1024;
constexpr int DD_MAX_ROWS = 4096;     // LDS: 8 B hash + 4 B rep per row

template <class T>
__device__ __forceinline__ long long tok(const T* p, size_t i) { return (long long)p[i]; }

template <class T>
__global__ __launch_bounds__(DD_THREADS) void instruction_dedup_kernel(const T* __restrict__ tokens, int B, int L, long long* __restrict__ uniq,
                                                                       long long* __restrict__ inverse, long long* __restrict__ meta) {
  __shared__ unsigned long long hsh[DD_MAX_ROWS];
  __shared__ int rep[DD_MAX_ROWS];
  __shared__ int wsum[DD_THREADS / 64];
  __shared__ int carry, lmax;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry = 0; lmax = 0; }
  // 1. hashes
  for (int b = wave; b < B; b += DD_THREADS / 64) {
    unsigned long long h = 0;
    for (int l = lane; l < L; l += 64)
      h += (unsigned long long)tok(tokens, (size_t)b * L + l) * (((unsigned long long)(l + 1) * 0x9E3779B97F4A7C15ull) | 1ull);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    if (lane == 0) hsh[b] = h;
  }
  __syncthreads();
  // 2. representatives
  for (int b = tid; b < B; b += DD_THREADS) {
    const unsigned long long h = hsh[b];
    int r = b;
    for (int j = 0; j < b; ++j) {
      if (hsh[j] != h) continue;
      // (no early exit inside the comparison: the loads of a row pair are independent and stay in flight together; a loop that
      //  stops at the first difference waits out one memory round trip per token)
      int diff = 0;
      const T* __restrict__ pj = tokens + (size_t)j * L;
      const T* __restrict__ pb = tokens + (size_t)b * L;
#pragma unroll 8
      for (int l = 0; l < L; ++l) diff |= (pj[l] != pb[l]);
      if (!diff) { r = j; break; }
    }
    rep[b] = r;
  }
  __syncthreads();
  // 3. rank of every representative = exclusive prefix sum of "is its own representative", in chunks of DD_THREADS rows
  for (int base = 0; base < B; base += DD_THREADS) {
    const int b = base + tid;
    const int first = (b < B && rep[b] == b) ? 1 : 0;
    int v = first;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    const int rank = off + v - first;
    if (first) {
      // 4. the representative's row and its length
      int n = 0;
      for (int l = 0; l < L; ++l) {
        const long long t = tok(tokens, (size_t)b * L + l);
        uniq[(size_t)rank * L + l] = t;
        n += t != 0;
      }
      meta[2 + rank] = n;
      atomicMax(&lmax, n);
      hsh[b] = (unsigned long long)rank;      // (the hash of a representative is not read again: its slot carries the rank)
    }
    __syncthreads();
    if (tid == 0) {
      int s = carry;
      for (int w = 0; w < DD_THREADS / 64; ++w) s += wsum[w];
      carry = s;
    }
    __syncthreads();
  }
  for (int b = tid; b < B; b += DD_THREADS) inverse[b] = (long long)hsh[rep[b]];
  if (tid == 0) { meta[0] = carry; meta[1] = lmax; }
}

}  // namespace

// tokens [B][L] float32 (is_f32: integer-valued floats, as the trainer hands them over after .float()) or int64; uniq [B][L] int64
// (rows 0 .. U-1 written), inverse [B] int64, meta [2 + B] int64.  B <= 4096.
extern "C" int wsmg_instruction_dedup(const void* tokens, int is_f32, int B, int L, long long* uniq, long long* inverse, long long* meta,
                                      wsmg_stream_t stream) {
  if (B <= 0 || B > DD_MAX_ROWS || L <= 0 || !tokens || !uniq || !inverse || !meta) return WSMG_EINVAL;
  if (is_f32)
    hipLaunchKernelGGL(instruction_dedup_kernel<float>, dim3(1), dim3(DD_THREADS), 0, wsmg_s(stream), (const float*)tokens, B, L, uniq, inverse, meta);
  else
    hipLaunchKernelGGL(instruction_dedup_kernel<long long>, dim3(1), dim3(DD_THREADS), 0, wsmg_s(stream), (const long long*)tokens, B, L, uniq,
                       inverse, meta);
  WSMG_RETURN_LAUNCH();
}
