// Dense layers of the recurrent core's attention stage on a CHUNK of rows (64-128 rows of the T x N batch), float32:
//
//     C = epilogue( [A0 | A1 | A2] W^T )            ("NT": W is an nn.Linear weight [N][K], the forward layers)
//     C = epilogue( [A0 | A1 | A2] W   )            ("NN": W is [K][N] — the same weights in the backward pass, dX = dY W)
//
// mg_map_policy.py:229-245 of the reference — state_text_q_layer, text_map_q_layer (+ the folded text_map_k_layer),
// second_state_compress (Linear + ReLU over cat(state, text_embedding, map_embedding)) and the second GRU's input projection — are
// five such products per chunk and direction between the two attention kernels.  Through the GEMM library each was 7-20 us at 128
// rows (plus a bias pre-fill, a concatenation, a ReLU and a threshold_backward launch): the attention stage of a chunk took
// 125 us forward / 150 us backward, longer than a chunk of either recurrence (105-120 us), so the three-stream pipeline of
// wsmgmap/recurrent.py was paced by it (profiles/r04_update_sections.txt).  Here a product is ONE launch with everything around it
// in the epilogue: bias, ReLU, the ReLU mask of the backward pass, an accumulate-into (beta = 1), the concatenation as up to
// three column segments of A, and the split of a backward product's columns into up to three output tensors.
//
// Tile = 16 rows x 16 columns per workgroup on v_mfma_f32_16x16x4f32 (float32 products and sums), the workgroup's 4-16 waves split K
// (each a contiguous share) and meet in LDS in a fixed order: deterministic.  At M = 128 a layer is 128-768 workgroups.
#include "wsmg_common.h"

namespace {

struct RowsGemmArgs {
  const float* a[3];     // column segments of A (unused: null / 0 columns)
  int lda[3], ka[3];
  const float* w;
  int ldw;               // NT: row stride of W[N][K]; NN: row stride of W[K][N]
  const float* bias;     // [N] or null
  const float* cin;      // accumulate-into source (same segmentation as C) or null
  const float* mask;     // ReLU-backward mask source [M][N] (C = mask > 0 ? C : 0) or null
  int ldmask;
  float* c[3];           // column segments of C
  int ldc[3], nc[3];
  const float* cin_seg[3];
  int ldcin[3];
  int M, N, K, relu, nn;
  // chaining to the persistent recurrences (wsmg_rnn.hip, chain_wait / chain_signal): every workgroup waits (bounded) until *wait_cnt
  // has reached wait_target before it reads anything, and adds one arrival to *signal_cnt when its tile is stored
  const unsigned* wait_cnt;
  unsigned wait_target;
  unsigned* signal_cnt;
  unsigned* status;      // process-wide status word (a wait that times out sets bit `fail_bit` and the tile becomes NaN)
  unsigned fail_bit;
  const unsigned* gate;  // or null: a word the gate launch in front of this one (chain_gate_kernel) left non-zero when ITS wait timed out
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Tile = 16 rows x 16 columns per workgroup (v_mfma_f32_16x16x4f32), WAVES waves split K: each takes a contiguous K / WAVES = 16 U
// and issues ALL its loads (U 16-deep chunks: 2 U 16-byte loads per lane; lane (r, kg) takes k0 + 4 kg .. + 3 of row r, the four values
// feed four consecutive MFMAs, A and W with the same k assignment) before its first MFMA — one memory round trip per launch.
// Everything in the k-loop is branch-free (the operand segment of a chunk is a pointer SELECT).  History of this kernel (round 5):
//   * `ok ? load : 0` guards became a branch and an `s_waitcnt vmcnt(0)` per load — a chain of dependent round trips, 10-30 us per
//     product; without the sched_barrier below hipcc sinks every load to just in front of its MFMA, same effect;
//   * 32 x 32 tiles (v_mfma_f32_32x32x2f32, 64 cycles each) put a product on 32-192 workgroups whose K / 2 dependent MFMAs share the
//     CU's four matrix pipes: K x 8 cycles = 3.9 us at K = 1024 however the waves split it, 10-15 us per product in the trace; at
//     16 x 16 the same product is 4x the workgroups (256 for the compress layer: the whole chip) with a quarter of the MFMA time each.
template <int WAVES, int U, bool NN>
__global__ __launch_bounds__(64 * WAVES) void rows_gemm_f32_kernel(RowsGemmArgs a) {
  __shared__ float red[WAVES][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int kbeg = wave * 16 * U;                   // this wave's share of the reduction (host: K == WAVES x 16 U)
  const int row = m0 + r < a.M ? m0 + r : a.M - 1;  // rows past M are computed on the last row and never stored
  const int s1 = a.ka[0], s2 = a.ka[0] + a.ka[1];
  bool timed_out = false;
  if (a.wait_cnt) {       // the operands are produced by a kernel that is still running on another stream (enqueued before this one)
    int bad = 0;
    if (tid == 0) {
      unsigned n = 0;
      while (__hip_atomic_load(a.wait_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.wait_target) {
        __builtin_amdgcn_s_sleep(32);                     // (~1 us: a poll per workgroup and microsecond, not a stream of them)
        if (++n >= (1u << 21)) { bad = 1; break; }        // (seconds)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (bad && a.status) __hip_atomic_fetch_or(a.status, a.fail_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    timed_out = __syncthreads_or(bad) != 0;
  } else if (a.gate) {
    timed_out = __syncthreads_or(tid == 0 && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) != 0;
  }
  const float* const p0 = a.a[0] + (size_t)row * a.lda[0];
  const float* const p1 = a.a[1] + (size_t)row * a.lda[1] - s1;      // (never dereferenced below s1: pointer arithmetic only)
  const float* const p2 = a.a[2] + (size_t)row * a.lda[2] - s2;
  const float* const wrow = NN ? a.w + n0 + r : a.w + (size_t)(n0 + r) * a.ldw;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 av[U], wv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = kbeg + 16 * u + 4 * kg;           // k .. k + 3 lie in one segment (host: segment widths % 16 == 0)
    const float* const ap = k < s1 ? p0 : (k < s2 ? p1 : p2);
    av[u] = ldg4(ap + k);
    if constexpr (!NN) {
      wv[u] = ldg4(wrow + k);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[u][j] = wrow[(size_t)(k + j) * a.ldw];
    }
  }
  __builtin_amdgcn_sched_barrier(0);                // all loads are issued above this line
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][j], wv[u][j], acc, 0, 0, 0);
#pragma unroll
  for (int g = 0; g < 4; ++g) red[wave][g][lane] = acc[g];
  __syncthreads();
  // column segment of this tile (host: segment widths % 16 == 0, so a tile lies in one segment)
  int seg = 0, cn = n0;
  if (cn >= a.nc[0]) { cn -= a.nc[0]; seg = 1; if (cn >= a.nc[1]) { cn -= a.nc[1]; seg = 2; } }
  float* const cb = a.c[seg];
  const float* const ib = a.cin ? a.cin_seg[seg] : nullptr;
  if (tid < 256) {
    const int g = tid >> 6, l = tid & 63;
    const int orow = m0 + 4 * (l >> 4) + g, col = l & 15;   // D of v_mfma_f32_16x16x4f32: lane l, register g -> row 4 (l / 16) + g, column l % 16
    float v = red[0][g][l];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v += red[w][g][l];     // fixed order: deterministic
    if (orow < a.M) {
      if (a.bias) v += a.bias[n0 + col];
      if (ib) v += ib[(size_t)orow * a.ldcin[seg] + cn + col];
      if (a.relu) v = v > 0.f ? v : 0.f;
      if (a.mask) v = a.mask[(size_t)orow * a.ldmask + n0 + col] > 0.f ? v : 0.f;
      const float ov = timed_out ? __uint_as_float(0x7fc00000u) : v;
      float* const op = cb + (size_t)orow * a.ldc[seg] + cn + col;
      // a signalling launch writes its tile THROUGH to the memory side (agent-scope stores): a waiter on another XCD must find it
      // there, and the alternative — an agent-scope release fence (an L2 write-back) in each of the launch's 256-768 workgroups —
      // made the second GRU's input projection 23 us instead of 9 (profiles/r05_update_timeline.txt)
      if (a.signal_cnt) __hip_atomic_store(op, ov, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *op = ov;
    }
  }
  if (a.signal_cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's write-through stores have been acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.signal_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The wait of a chained product as its OWN one-workgroup launch in front of it (round 5, end).  With the wait inside the product every
// workgroup of it spins while resident — the backward pass's first product of a chunk is 768 workgroups of 16 waves: two per CU, the
// whole chip's wave slots — and the persistent recurrence it waits for needs all 32 of ITS workgroups resident to take a step: which
// of the two reaches the CUs first is decided by a few microseconds of stream timing (removing one redundant event record in
// wsmgmap/recurrent.py was enough to turn it: the recurrences timed out in 4 of 5 runs).  One spinning wave cannot starve anything.
__global__ __launch_bounds__(64) void chain_gate_kernel(const unsigned* cnt, unsigned target, unsigned* status, unsigned fail_bit,
                                                        unsigned* gate) {
  if (threadIdx.x != 0) return;
  unsigned n = 0;
  int bad = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(32);
    if (++n >= (1u << 21)) { bad = 1; break; }        // (seconds)
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (bad && status) __hip_atomic_fetch_or(status, fail_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(gate, bad ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int WAVES, int U>
void launch_rows(const RowsGemmArgs& g, dim3 grid, hipStream_t s) {
  if (g.nn) hipLaunchKernelGGL((rows_gemm_f32_kernel<WAVES, U, true>), grid, dim3(64 * WAVES), 0, s, g);
  else hipLaunchKernelGGL((rows_gemm_f32_kernel<WAVES, U, false>), grid, dim3(64 * WAVES), 0, s, g);
}

// ---- debug: hold n workgroups' worth of CUs (whole CUs: 1 024 threads and `lds_bytes` of LDS each) until *stop != 0 or max_us passed
__global__ __launch_bounds__(1024) void debug_occupy_kernel(const volatile int* stop, unsigned long long max_ticks, unsigned* arrived) {
  extern __shared__ unsigned char hold[];
  if (threadIdx.x == 0) {
    hold[0] = 1;
    atomicAdd(arrived, 1u);
    const unsigned long long t0 = wall_clock64();
    while (!*stop && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(64);
  }
  __syncthreads();
}

}  // namespace

extern "C" int wsmg_rows_gemm_f32(const float* a0, int lda0, int ka0, const float* a1, int lda1, int ka1, const float* a2, int lda2, int ka2,
                                  const float* w, int ldw, int w_is_kn, const float* bias, const float* mask, int ldmask, int relu,
                                  float* c0, int ldc0, int nc0, float* c1, int ldc1, int nc1, float* c2, int ldc2, int nc2,
                                  const float* cin0, int ldcin0, const float* cin1, int ldcin1, const float* cin2, int ldcin2,
                                  int M, const unsigned* wait_count, unsigned wait_target, unsigned* signal_count, int fail_bit,
                                  unsigned* gate_word, wsmg_stream_t stream) {
  if (wait_count && !gate_word) return WSMG_EINVAL;
  if (!a0 || !w || !c0 || M <= 0 || ka0 <= 0 || nc0 <= 0) return WSMG_EINVAL;
  if ((ka1 > 0 && !a1) || (ka2 > 0 && !a2) || (nc1 > 0 && !c1) || (nc2 > 0 && !c2) || ka1 < 0 || ka2 < 0 || nc1 < 0 || nc2 < 0) return WSMG_EINVAL;
  if (ka2 > 0 && ka1 <= 0) return WSMG_EINVAL;
  if (nc2 > 0 && nc1 <= 0) return WSMG_EINVAL;
  const int K = ka0 + ka1 + ka2, N = nc0 + nc1 + nc2;
  if (K % 64 || ka0 % 16 || ka1 % 16 || ka2 % 16 || nc0 % 16 || nc1 % 16 || nc2 % 16) return WSMG_EINVAL;
  if ((lda0 | lda1 | lda2 | ldw) & 3) return WSMG_EINVAL;       // 16-byte loads along K (NT) / rows of A
  const bool any_cin = cin0 != nullptr;
  if (any_cin && ((nc1 > 0 && !cin1) || (nc2 > 0 && !cin2))) return WSMG_EINVAL;
  RowsGemmArgs g;
  g.a[0] = a0; g.a[1] = a1; g.a[2] = a2;
  g.lda[0] = lda0; g.lda[1] = lda1; g.lda[2] = lda2;
  g.ka[0] = ka0; g.ka[1] = ka1; g.ka[2] = ka2;
  g.w = w; g.ldw = ldw; g.bias = bias; g.cin = any_cin ? cin0 : nullptr; g.mask = mask; g.ldmask = ldmask;
  g.c[0] = c0; g.c[1] = c1; g.c[2] = c2;
  g.ldc[0] = ldc0; g.ldc[1] = ldc1; g.ldc[2] = ldc2;
  g.nc[0] = nc0; g.nc[1] = nc1; g.nc[2] = nc2;
  g.cin_seg[0] = cin0; g.cin_seg[1] = cin1; g.cin_seg[2] = cin2;
  g.ldcin[0] = ldcin0; g.ldcin[1] = ldcin1; g.ldcin[2] = ldcin2;
  g.M = M; g.N = N; g.K = K; g.relu = relu; g.nn = w_is_kn;
  g.wait_cnt = wait_count; g.wait_target = wait_target; g.signal_cnt = signal_count;
  g.status = wait_count ? wsmgi_rnn_status_dev() : nullptr; g.fail_bit = (unsigned)fail_bit;
  g.gate = nullptr;
  if (wait_count) {
    // the wait is a ONE-workgroup launch in front of the product (a grid of spinning workgroups can keep the producer it waits for off
    // the CUs: DESIGN.md section 5); its verdict goes through the CALLER's word (round 6, ADVICE r05: a process-wide word per fail
    // bit let a second stream's gate overwrite a timeout before this product had read it)
    hipLaunchKernelGGL(chain_gate_kernel, dim3(1), dim3(64), 0, wsmg_s(stream), wait_count, wait_target, g.status, g.fail_bit, gate_word);
    g.wait_cnt = nullptr;
    g.gate = gate_word;
  }
  const dim3 grid((unsigned)(N / 16), (unsigned)wsmg_cdiv(M, 16));
  // (waves, chunks per wave) with waves x 16 x chunks == K: K = 256: 16 x 1, 512: 16 x 2, 1024: 16 x 4, 1536: 16 x 6 — every product of
  // the recurrent core is one round of loads; other multiples of 64: 4 waves
  const int cap = 16;
  hipStream_t st = wsmg_s(stream);
  const int c16 = K / 16;                           // 16-deep chunks in all
  if (cap >= 16 && c16 % 16 == 0 && c16 / 16 <= 8) {
    switch (c16 / 16) {
      case 1: launch_rows<16, 1>(g, grid, st); break;
      case 2: launch_rows<16, 2>(g, grid, st); break;
      case 3: launch_rows<16, 3>(g, grid, st); break;
      case 4: launch_rows<16, 4>(g, grid, st); break;
      case 6: launch_rows<16, 6>(g, grid, st); break;
      case 8: launch_rows<16, 8>(g, grid, st); break;
      default: return WSMG_EINVAL;                  // (K = 1280, 1792: not a shape of this path)
    }
  } else if (c16 % 4 == 0 && c16 / 4 <= 8) {
    switch (c16 / 4) {
      case 1: launch_rows<4, 1>(g, grid, st); break;
      case 2: launch_rows<4, 2>(g, grid, st); break;
      case 3: launch_rows<4, 3>(g, grid, st); break;
      case 4: launch_rows<4, 4>(g, grid, st); break;
      case 6: launch_rows<4, 6>(g, grid, st); break;
      case 8: launch_rows<4, 8>(g, grid, st); break;
      default: return WSMG_EINVAL;
    }
  } else {
    return WSMG_EINVAL;
  }
  WSMG_RETURN_LAUNCH();
}

// 1 if wsmg_rows_gemm_f32 has a launch form for a reduction of K (the sum of the operand segments' widths), else 0
extern "C" int wsmg_rows_gemm_supported(int K) {
  if (K <= 0 || K % 64) return 0;
  const int c16 = K / 16;
  auto form = [](int n) { return n == 1 || n == 2 || n == 3 || n == 4 || n == 6 || n == 8; };
  if (c16 % 16 == 0 && c16 / 16 <= 8) return form(c16 / 16) ? 1 : 0;
  return (c16 % 4 == 0 && c16 / 4 <= 8 && form(c16 / 4)) ? 1 : 0;
}

extern "C" int wsmg_rows_gemm_workgroups(int M, int N) { return (M <= 0 || N <= 0) ? 0 : (N / 16) * (int)wsmg_cdiv(M, 16); }

extern "C" int wsmg_debug_occupy(int n_workgroups, int lds_bytes, int max_ms, const int* stop_flag, unsigned* arrived, wsmg_stream_t stream) {
  if (n_workgroups <= 0 || n_workgroups > 256 || lds_bytes < 0 || lds_bytes > 160 * 1024 || max_ms <= 0 || max_ms > 10000 || !stop_flag || !arrived)
    return WSMG_EINVAL;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(debug_occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  // wall_clock64 ticks at 100 MHz on gfx950
  hipLaunchKernelGGL(debug_occupy_kernel, dim3((unsigned)n_workgroups), dim3(1024), (size_t)lds_bytes, wsmg_s(stream),
                     (const volatile int*)stop_flag, (unsigned long long)max_ms * 100000ull, arrived);
  WSMG_RETURN_LAUNCH();
}
