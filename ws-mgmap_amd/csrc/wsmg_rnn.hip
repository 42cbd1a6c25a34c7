// Persistent masked-GRU sequence kernels (forward + backward-through-time) for the two state
// encoders of the policy (hidden 512, batch N <= 8 per step, T <= 200 steps).
//
// Replaces the per-time-step cuDNN/MIOpen GRU launches behind habitat-lab's RNNStateEncoder
// (reference call sites mg_map_policy.py:220-227,242-249).  At N = 8 the recurrence is a chain of
// T dependent [8 x 512] x [512 x 1536] products: latency-, not FLOP-bound (about 4000 tiny
// launches per update in MIOpen).  Here ONE launch runs the whole sequence:
//   * 32 workgroups x 4 waves; each wave owns 4 hidden units (12 gate rows of W_hh, or 4 columns
//     for backward) and keeps them in REGISTERS for all T steps — no LDS or HBM weight traffic
//     inside the loop;
//   * the K = 512 (1536) reduction is split across the 64 lanes; partial sums are combined with a
//     halving butterfly (each shuffle halves the live values), so lane l ends up with its own
//     (unit, batch) outputs;
//   * h_t (resp. dGH_t) is exchanged through HBM with ONE grid-wide barrier per step: plain
//     stores -> vmcnt(0) -> workgroup barrier -> agent-scope release -> counter add; consumers
//     poll relaxed, then one agent-scope acquire (cdna_hip_programming.md Guideline 16).  Spins
//     are bounded: on timeout an error word is set and every workgroup exits (no GPU hang).
// Episode restarts are handled in-kernel: h_{t-1} is multiplied by masks[t] before every step,
// which is what the reference's split-at-zeros sequence form computes — so no host sync is needed.
// Input projections (x W_ih^T + b_ih for all T*N rows) and the weight gradients are single large
// GEMMs done by the caller.  Gate order r, z, n (PyTorch nn.GRU).
#include <stdlib.h>

#include "wsmg_common.h"

// This file must be compiled WITHOUT packed-float32 instructions (Makefile: EXTRA_wsmg_rnn): with them the gradient
// kernels returned wrong partial sums beside co-resident MFMA waves.  The define travels with the compiler flag.
#ifndef WSMG_RNN_NO_PK_FP32
#error "wsmg_rnn.hip: compile with -Xclang -target-feature -Xclang -packed-fp32-ops -DWSMG_RNN_NO_PK_FP32 (see csrc/Makefile)"
#endif
__attribute__((visibility("hidden"))) int wsmgi_rnn_no_pk_fp32() { return 1; }

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int H = 512;
constexpr int NB = 8;            // batch slots per step
constexpr int UNITS_WAVE = 4;
constexpr int WAVES = 4;
constexpr int UNITS_WG = UNITS_WAVE * WAVES;   // 16
constexpr int NWG = H / UNITS_WG;              // 32
constexpr unsigned SPIN_LIMIT = 1u << 20;   // default bound of every spin; wsmg_rnn_debug_spin_limit() lowers it (tests)

// What a kernel does when a spin timed out (or another workgroup reported one): the process-wide status word in
// host-mapped pinned memory gets the kernel's bit (system-scope store: the host reads it without synchronising),
// and the workgroup's slice of every output is filled with NaN, so nothing downstream can mistake the
// uninitialised rows for results.
__device__ __forceinline__ void rnn_fail(unsigned* host_status, unsigned bit) {
  if (threadIdx.x == 0 && host_status) __hip_atomic_fetch_or(host_status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// rows x [col0, col0 + ncols) of a row-major [rows][rowlen] float tensor <- NaN
__device__ __forceinline__ void rnn_poison(float* base, size_t rows, int rowlen, int col0, int ncols) {
  if (!base) return;
  const float nan = __uint_as_float(0x7fc00000u);
  for (size_t i = threadIdx.x; i < rows * (size_t)ncols; i += blockDim.x)
    base[(i / ncols) * rowlen + col0 + (int)(i % ncols)] = nan;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// one-per-step grid barrier among NWG co-resident workgroups; returns false on timeout
__device__ __forceinline__ bool grid_barrier(unsigned* sync, unsigned target, int tid, int* ok_lds, unsigned limit) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned n = 0;
    int good = 1;
    while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++n >= limit || __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *ok_lds = good;
  }
  __syncthreads();
  return *ok_lds != 0;
}

// ---- round 5: chaining a whole-sequence launch to kernels on OTHER streams by time chunk (wsmgmap/recurrent.py) -------------------
// The pipelined recurrent core cut each recurrence into K launches so that the attention stage of chunk k could start when the
// first recurrence had finished chunk k, and the second recurrence's chunk k when the attention stage had: every chunk launch paid
// the ~10 us prologue (W_hh into registers) and a launch gap again, 16 times per update.  Chained, a recurrence is ONE launch:
//   * a PRODUCER (in_cnt == null side) adds one arrival per workgroup to out_cnt[k] when the outputs of its chunk k are complete
//     (all threads' stores drained -> workgroup barrier -> agent-scope release -> add);
//   * a CONSUMER spins (bounded, thread 0) at the first step of chunk k until in_cnt[k] has reached in_target, then an
//     agent-scope acquire; the kernels that fill its inputs run on another stream, enqueued BEFORE it, and signal the same way
//     (wsmg_rows_gemm_f32's `signal`).
// Counters are zeroed by the host once per pass.  A waiter is always enqueued after its producers, so whatever hardware queues the
// streams share, the producer is never stuck behind its own waiter.
__device__ __forceinline__ bool chain_wait(const unsigned* cnt, unsigned target, unsigned* sync, unsigned limit, int tid) {
  int bad = 0;
  if (tid == 0) {
    unsigned n = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(32);      // (~1 us between polls: 32 waiting workgroups must not compete with the running recurrence's exchange)
      if (++n >= limit || __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  return !__syncthreads_or(bad);
}
__device__ __forceinline__ void chain_signal(unsigned* cnt, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// cross-lane exchange with lane ^ D at VALU rate where the ISA has a pattern for it: xor 1 / xor 2 =
// DPP quad_perm, xor 8 = DPP row rotate by 8; xor 4 falls back to ds_bpermute
template <int D>
__device__ __forceinline__ float lane_xor(float v) {
  const int x = __float_as_int(v);
  if constexpr (D == 1) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  else if constexpr (D == 2) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  else if constexpr (D == 8) return __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false));  // row_ror:8
  else return __shfl_xor(v, D, 64);
}
// Halving butterfly step: C live values per lane, xor distance D; afterwards C/2 live values: lanes with
// bit D clear hold the sums of the lower half, lanes with bit D set those of the upper half.  For D = 32 and
// 16 one v_permlane{32,16}_swap exchanges the halves of a register pair (gfx950), so a step costs one swap
// and one add per output instead of a ds_bpermute and two selects (the 127-exchange butterfly of the
// forward kernel measured 3.7 us per step with ds_bpermute).
template <int C, int D, int NV>
__device__ __forceinline__ void halve(float (&v)[NV], int lane) {
  if constexpr (D == 32 || D == 16) {
#pragma unroll
    for (int i = 0; i < C / 2; ++i) {
      const int lo = __float_as_int(v[i]), hi = __float_as_int(v[i + C / 2]);
      // swap: lanes with bit D set of `lo` <-> lanes with bit D clear of `hi`
      auto r = (D == 32) ? __builtin_amdgcn_permlane32_swap(lo, hi, false, false)
                         : __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
      v[i] = __int_as_float(r[0]) + __int_as_float(r[1]);
    }
  } else {
    const bool up = (lane & D) != 0;
#pragma unroll
    for (int i = 0; i < C / 2; ++i) {
      float keep = up ? v[i + C / 2] : v[i];
      float send = up ? v[i] : v[i + C / 2];
      v[i] = keep + lane_xor<D>(send);
    }
  }
}

struct GruFwdArgs {
  const float* gi;     // [T][N][3H]  x W_ih^T + b_ih
  const float* whh;    // [3H][H]
  const float* bhh;    // [3H]
  const float* h0;     // [N][H]
  const float* masks;  // [T][N]
  float* y;            // [T][N][H]   h_t
  float* sr;           // saved gates for backward, each [T][N][H]
  float* sz;
  float* sn;
  float* sghn;         // W_hn h + b_hn
  unsigned* sync;      // [1] error word (zeroed by the launcher)
  unsigned long long* xh;  // exchange [T][NWG][NB][UNITS_WG] of {value, tag} words, each written once per launch
  int T, N;
  unsigned tagbase;    // launch-unique tag bits (epoch << 10); a word is valid for step t when tag == tagbase | (t + 1)
  unsigned* status;    // host-mapped process status word (rnn_fail)
  unsigned spin;       // spin bound
  // chaining by time chunk (see chain_wait): steps per chunk (0: none), the counters gi's chunks are ready behind / this launch reports on
  int Tc;
  const unsigned* in_cnt;
  unsigned in_target;
  unsigned* out_cnt;
};

// Flag-in-data exchange (forward): every h value travels as one 8-byte {value, tag} word written with a
// single agent-scope store and read with agent-scope loads, so a reader that sees the step's tag has the
// step's value — no counter, no fences, no separate data read after a barrier: ONE memory round trip per
// step instead of three.  All 256 threads of a workgroup poll the 4096 words of the step (16 each: one
// 128-byte row of one producer workgroup and batch), drop the values into LDS, and the four waves read
// h_{t-1} from there.  Tags are launch-unique (epoch) and slots step-indexed, so stale contents of the
// image — from an older launch or an older step — can never match.
__device__ __forceinline__ bool poll_row16(const unsigned long long* src, unsigned want, unsigned* sync, float (&out)[16], unsigned limit) {
  unsigned n = 0;
  for (;;) {
    unsigned long long v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 16; ++i) ok = ok && ((unsigned)(v[i] >> 32) == want);
    if (ok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) out[i] = __uint_as_float((unsigned)v[i]);
      return true;
    }
    __builtin_amdgcn_s_sleep(1);
    if (++n >= limit || __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

template <int D>
__device__ __forceinline__ float row_xor(float v) { return lane_xor<D>(v); }
// halving butterfly inside 16-lane rows: C live values per lane, xor distance D; afterwards C/2
template <int C, int D, int NV>
__device__ __forceinline__ void halve_row(float (&v)[NV], int lane) {
  const bool up = (lane & D) != 0;
#pragma unroll
  for (int i = 0; i < C / 2; ++i) {
    float keep = up ? v[i + C / 2] : v[i];
    float send = up ? v[i] : v[i + C / 2];
    v[i] = keep + row_xor<D>(send);
  }
}

// (The four-wave forms of the two GRU kernels — rounds 1-2, then the WSMG_GRU_WAVES=4 arm of an A/B — are gone: the chained core has
//  only ever run the eight-wave forms below.)

// ---- 8-wave form of the forward kernel (round 3).  One wave per SIMD issues a vector instruction every 4 cycles, two every 2:
// the 768 FMAs per lane of the 4-wave form are the longest phase of its step (1.8 us of 5.5).  Here waves w and w + 4 share the
// 4 units of wave w and split K: wave half q owns k = 64 j + 4 kl + e for j = 4 q .. 4 q + 3 (48 weights per lane instead of 96,
// 384 FMAs); the upper half's sums cross to the lower half through LDS behind one more workgroup barrier, the lower half does the
// gate math as before.  All 512 threads poll: 8 {value, tag} words each instead of 16.  Same arithmetic per output up to the
// order of the two K halves' addition.
__device__ __forceinline__ bool poll_row8(const unsigned long long* src, unsigned want, unsigned* sync, float (&out)[8], unsigned limit) {
  unsigned n = 0;
  for (;;) {
    unsigned long long v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) ok = ok && ((unsigned)(v[i] >> 32) == want);
    if (ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) out[i] = __uint_as_float((unsigned)v[i]);
      return true;
    }
    __builtin_amdgcn_s_sleep(1);
    if (++n >= limit || __hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

__global__ __launch_bounds__(512) void gru_fwd8_kernel(GruFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float hs[2][NB][H];   // h_{t-1}, double-buffered by step parity
  __shared__ float part[WAVES][64][2];                          // the upper K half's two sums per lane
  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
  const int wave = wave8 & 3, q = wave8 >> 2;                   // unit group of the wave, K half
  const int grp = lane >> 4, kl = lane & 15;
  const int my_unit = blockIdx.x * UNITS_WG + wave * UNITS_WAVE + grp;
  float w[3][16];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = *reinterpret_cast<const f32x4*>(a.whh + (size_t)(g * H + my_unit) * H + 64 * (4 * q + j) + 4 * kl);
#pragma unroll
      for (int e = 0; e < 4; ++e) w[g][4 * j + e] = v[e];
    }
  const int my_b = kl >> 1;
  const bool worker = (q == 0) && ((kl & 1) == 0) && (my_b < a.N);
  const float br = a.bhh[my_unit], bz = a.bhh[H + my_unit], bn = a.bhh[2 * H + my_unit];
  // staging role: half `sh` of the 16 units of producer workgroup sw for batch sb
  const int sw = tid >> 4, sb = (tid >> 1) & 7, sh = tid & 1;
  const int xw = (blockIdx.x * NB + my_b) * UNITS_WG + wave * UNITS_WAVE + grp;
  for (int t = 0; t < a.T; ++t) {
    if (a.in_cnt && t % a.Tc == 0 && !chain_wait(a.in_cnt + t / a.Tc, a.in_target, a.sync, a.spin, tid)) {   // gi of this chunk is not there yet
      rnn_fail(a.status, 1u);
      rnn_poison(a.y, (size_t)a.T * a.N, H, blockIdx.x * UNITS_WG, UNITS_WG);
      return;
    }
    float (*hcur)[H] = hs[t & 1];
    float gr = 0.f, gz = 0.f, gn = 0.f;
    const size_t orow = (size_t)t * a.N + my_b;
    if (worker) {
      const float* g = a.gi + orow * 3 * H;
      gr = g[my_unit]; gz = g[H + my_unit]; gn = g[2 * H + my_unit];
    }
    {
      float row[8];
      bool good = true;
      if (sb >= a.N) {
#pragma unroll
        for (int i = 0; i < 8; ++i) row[i] = 0.f;
      } else if (t == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x4 v = *reinterpret_cast<const f32x4*>(a.h0 + (size_t)sb * H + sw * UNITS_WG + 8 * sh + 4 * i);
          row[4 * i] = v[0]; row[4 * i + 1] = v[1]; row[4 * i + 2] = v[2]; row[4 * i + 3] = v[3];
        }
      } else {
        const unsigned long long* src = a.xh + ((size_t)(t - 1) * NWG + sw) * NB * UNITS_WG + sb * UNITS_WG + 8 * sh;
        good = poll_row8(src, a.tagbase | (unsigned)t, a.sync, row, a.spin);
      }
      const float sm = sb < a.N ? a.masks[t * a.N + sb] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v = {row[4 * i] * sm, row[4 * i + 1] * sm, row[4 * i + 2] * sm, row[4 * i + 3] * sm};
        *reinterpret_cast<f32x4*>(&hcur[sb][sw * UNITS_WG + 8 * sh + 4 * i]) = v;
      }
      if (__syncthreads_or(good ? 0 : 1)) {   // timeout or error elsewhere: every thread leaves
        rnn_fail(a.status, 1u);
        rnn_poison(a.y, (size_t)a.T * a.N, H, blockIdx.x * UNITS_WG, UNITS_WG);
        return;
      }
    }
    float acc[32];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      if (b < a.N) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(&hcur[b][64 * (4 * q + j) + 4 * kl]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 = fmaf(w[0][4 * j + e], hv[e], s0);
            s1 = fmaf(w[1][4 * j + e], hv[e], s1);
            s2 = fmaf(w[2][4 * j + e], hv[e], s2);
          }
        }
      }
      acc[4 * b] = s0; acc[4 * b + 1] = s1; acc[4 * b + 2] = s2; acc[4 * b + 3] = 0.f;
    }
    halve_row<32, 8>(acc, lane);
    halve_row<16, 4>(acc, lane);
    halve_row<8, 2>(acc, lane);
    halve_row<4, 1>(acc, lane);
    if (q == 1) { part[wave][lane][0] = acc[0]; part[wave][lane][1] = acc[1]; }
    __syncthreads();
    if (q == 0) { acc[0] += part[wave][lane][0]; acc[1] += part[wave][lane][1]; }
    const float nsum = row_xor<1>(acc[0]);   // odd lane's acc[0] = n gate of the same batch
    if (worker) {
      const float ghr = acc[0] + br, ghz = acc[1] + bz, ghn = nsum + bn;
      const float r = sigmoidf_(gr + ghr);
      const float z = sigmoidf_(gz + ghz);
      const float nn = tanhf(gn + r * ghn);
      const float hprev = hcur[my_b][my_unit];   // already masked
      const float h = (1.0f - z) * nn + z * hprev;
      if (t + 1 < a.T)
        __hip_atomic_store(a.xh + (size_t)t * NWG * NB * UNITS_WG + xw,
                           ((unsigned long long)(a.tagbase | (unsigned)(t + 1)) << 32) | __float_as_uint(h),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.y[orow * H + my_unit] = h;
      a.sr[orow * H + my_unit] = r;
      a.sz[orow * H + my_unit] = z;
      a.sn[orow * H + my_unit] = nn;
      a.sghn[orow * H + my_unit] = ghn;
    }
    if (a.out_cnt && (t + 1) % a.Tc == 0) chain_signal(a.out_cnt + t / a.Tc, tid);   // y of this chunk is complete (this workgroup's part)
  }
}

struct GruBwdArgs {
  const float* dy;     // [T][N][H]   gradient w.r.t. every h_t
  const float* dhT;    // [N][H] gradient w.r.t. the final hidden state, or null
  const float* whh;    // [3H][H]
  const float* h0;     // [N][H]
  const float* masks;  // [T][N]
  const float* y;      // [T][N][H]
  const float* sr;
  const float* sz;
  const float* sn;
  const float* sghn;
  float* dgi;          // [T][N][3H]
  float* dgh;          // [T][N][3H]
  float* dh0;          // [N][H]
  unsigned* sync;      // [1] error word (zeroed by the launcher)
  unsigned long long* xp;  // exchange ring [BWD_RING][consumer WG][producer WG][NB][UNITS_WG] of {value, tag} words
  int T, N;
  unsigned tagbase;    // launch-unique tag bits; a word belongs to step t when tag == tagbase | (t + 1)
  unsigned* status;    // host-mapped process status word (rnn_fail)
  unsigned spin;       // spin bound
  // chaining by time chunk (see chain_wait): steps per chunk (0: none), the counters dy's chunks are ready behind / this launch reports on
  int Tc;
  const unsigned* in_cnt;
  unsigned in_target;
  unsigned* out_cnt;
};

// Backward exchange: PARTIAL SUMS of dh, not gate gradients.  dh_{t-1}[k] needs sum over all 3H gate rows of
// W_hh[row][k] * dgate_t[row]; the gate gradients of unit u are produced by the workgroup that owns u.  Shipping
// them to every workgroup (the first version of this kernel) is 3 words per unit = 48 KB per workgroup and step
// behind a grid barrier: three memory round trips, 9.7 us per step.  Here the producer keeps its 48 gate rows
// (r, z, n of its 16 units) of W_hh in registers — the SAME rows the forward kernel holds — multiplies them with
// its own gate gradients straight out of LDS (768 FMAs per lane, no cross-lane reduction: a thread owns two
// columns k and all 8 batch slots), and publishes its contribution to dh_{t-1}[k][b] for ALL 512 k as {value,
// tag} words, grouped by the workgroup that owns k.  A consumer polls the 32 producers' words for its 16 units
// (4096 words = 32 KB, exactly the forward's volume and code path) and adds them: ONE round trip per step.
// Ring of 4 step slots: a producer can be at step t only after every workgroup has published step t+1, i.e.
// has finished reading step t+2, so slots t+2 and older are free for reuse.
constexpr int BWD_RING = 4;
constexpr size_t XP_CONSUMER = (size_t)NWG * NB * UNITS_WG;   // words one consumer polls per step (4096)
constexpr size_t XP_SLOT = (size_t)NWG * XP_CONSUMER;         // words per ring slot (1 MB)


// ---- 8-wave form of the backward kernel (round 3; see gru_fwd8_kernel).  Thread pair (tid, tid + 256) owns the same two columns
// k = 2 c, 2 c + 1 (c = tid & 255) and splits the 48 gate ROWS: half q multiplies rows 24 q .. 24 q + 23 (24 weight pairs per lane
// instead of 48, 384 FMAs, 48 LDS reads: the workgroup's LDS read volume is unchanged, which a split by columns would double).
// The halves swap the partial sums the other one finishes through LDS — half 0 finishes batch slots 0-3, half 1 slots 4-7 — and
// each publishes its 8 words; all 512 threads poll 8 words each.
__global__ __launch_bounds__(512) void gru_bwd8_kernel(GruBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float part[NWG][NB][UNITS_WG];   // the 32 producers' partial sums for my units
  __shared__ __attribute__((aligned(16))) float dgs[3 * UNITS_WG][NB];     // my gate gradients of this step: rows r, z, n
  __shared__ __attribute__((aligned(16))) float swp[2][256][8];            // swp[q][c]: what half q hands to the other half
  const int tid = threadIdx.x;
  const int c = tid & 255, q = tid >> 8;
  f32x2 w[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    const int rw = 24 * q + i, g = rw / UNITS_WG, u = rw - g * UNITS_WG;
    w[i] = *reinterpret_cast<const f32x2*>(a.whh + (size_t)(g * H + blockIdx.x * UNITS_WG + u) * H + 2 * c);
  }
  // element-wise role (threads 0..127): unit wu of this workgroup, batch slot wb
  const int wu = tid & 15, wb = tid >> 4;
  const bool worker = tid < 128 && wb < a.N;
  const int my_unit = blockIdx.x * UNITS_WG + wu;
  // polling role: half ph of the 16 units of producer pq for batch slot pb
  const int pq = tid >> 4, pb = (tid >> 1) & 7, ph = tid & 1;
  // publishing role: columns 2 c, 2 c + 1 belong to consumer workgroup c / 8, its units (2 c) % 16 and + 1; batch slots 4 q .. 4 q + 3
  const size_t pub = ((size_t)(c >> 3) * NWG + blockIdx.x) * NB * UNITS_WG + ((2 * c) & 15);

  float direct = 0.f, mk_next = 0.f;
  for (int t = a.T - 1; t >= -1; --t) {
    if (a.in_cnt && t >= 0 && t % a.Tc == a.Tc - 1 && !chain_wait(a.in_cnt + t / a.Tc, a.in_target, a.sync, a.spin, tid)) {   // dy of this chunk
      rnn_fail(a.status, 2u);
      for (int g = 0; g < 3; ++g) {
        rnn_poison(a.dgi, (size_t)a.T * a.N, 3 * H, g * H + blockIdx.x * UNITS_WG, UNITS_WG);
        rnn_poison(a.dgh, (size_t)a.T * a.N, 3 * H, g * H + blockIdx.x * UNITS_WG, UNITS_WG);
      }
      rnn_poison(a.dh0, (size_t)a.N, H, blockIdx.x * UNITS_WG, UNITS_WG);
      return;
    }
    float dyv = 0.f, r = 0.f, z = 0.f, nn = 0.f, ghn = 0.f, hprev = 0.f, mk = 0.f;
    size_t row = 0;
    if (worker && t >= 0) {
      row = (size_t)t * a.N + wb;
      const size_t o = row * H + my_unit;
      mk = a.masks[t * a.N + wb];
      const float* hsrc = (t == 0) ? a.h0 : a.y + (size_t)(t - 1) * a.N * H;
      hprev = hsrc[(size_t)wb * H + my_unit] * mk;
      dyv = a.dy[o];
      r = a.sr[o]; z = a.sz[o]; nn = a.sn[o]; ghn = a.sghn[o];
    }
    float carry = 0.f;
    if (t == a.T - 1) {
      if (worker && a.dhT) carry = a.dhT[(size_t)wb * H + my_unit];
    } else {
      bool good = true;
      if (pb < a.N) {
        float rowv[8];
        const unsigned long long* src = a.xp + (size_t)((t + 1) % BWD_RING) * XP_SLOT + (size_t)blockIdx.x * XP_CONSUMER +
                                        ((size_t)pq * NB + pb) * UNITS_WG + 8 * ph;
        good = poll_row8(src, a.tagbase | (unsigned)(t + 2), a.sync, rowv, a.spin);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x4 v = {rowv[4 * i], rowv[4 * i + 1], rowv[4 * i + 2], rowv[4 * i + 3]};
          *reinterpret_cast<f32x4*>(&part[pq][pb][8 * ph + 4 * i]) = v;
        }
      }
      if (__syncthreads_or(good ? 0 : 1)) {   // timeout or error elsewhere: every thread leaves
        rnn_fail(a.status, 2u);
        for (int g = 0; g < 3; ++g) {
          rnn_poison(a.dgi, (size_t)a.T * a.N, 3 * H, g * H + blockIdx.x * UNITS_WG, UNITS_WG);
          rnn_poison(a.dgh, (size_t)a.T * a.N, 3 * H, g * H + blockIdx.x * UNITS_WG, UNITS_WG);
        }
        rnn_poison(a.dh0, (size_t)a.N, H, blockIdx.x * UNITS_WG, UNITS_WG);
        return;
      }
      if (worker) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < NWG; p2 += 2) { s0 += part[p2][wb][wu]; s1 += part[p2 + 1][wb][wu]; }
        carry = (direct + (s0 + s1)) * mk_next;
      }
    }
    if (t < 0) {
      if (worker) a.dh0[(size_t)wb * H + my_unit] = carry;
      break;
    }
    if (tid < 128) {
      float dr_pre = 0.f, dz_pre = 0.f, dnr = 0.f;
      if (worker) {
        const float dh = dyv + carry;
        const float dn_pre = dh * (1.0f - z) * (1.0f - nn * nn);
        dz_pre = dh * (hprev - nn) * z * (1.0f - z);
        dr_pre = dn_pre * ghn * r * (1.0f - r);
        dnr = dn_pre * r;
        float* gi = a.dgi + row * 3 * H;
        float* gh = a.dgh + row * 3 * H;
        gi[my_unit] = dr_pre; gi[H + my_unit] = dz_pre; gi[2 * H + my_unit] = dn_pre;
        gh[my_unit] = dr_pre; gh[H + my_unit] = dz_pre; gh[2 * H + my_unit] = dnr;
        direct = dh * z;
        mk_next = mk;
      }
      dgs[wu][wb] = dr_pre; dgs[UNITS_WG + wu][wb] = dz_pre; dgs[2 * UNITS_WG + wu][wb] = dnr;   // zeros for unused batch slots
    }
    __syncthreads();   // dgs complete; also: every read of part[] / swp[] of the previous step is done
    float acc0[NB], acc1[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { acc0[b] = 0.f; acc1[b] = 0.f; }
    const unsigned dgs_addr = (unsigned)(size_t)&dgs[0][0] + (unsigned)(24 * q * NB * 4);     // this half's 24 rows
#pragma unroll
    for (int c8 = 0; c8 < 24; c8 += 8) {
      f32x4 d[16];
      asm volatile("" :: "v"(acc0[0]), "v"(acc0[1]), "v"(acc0[2]), "v"(acc0[3]), "v"(acc0[4]), "v"(acc0[5]), "v"(acc0[6]), "v"(acc0[7]),
                         "v"(acc1[0]), "v"(acc1[1]), "v"(acc1[2]), "v"(acc1[3]), "v"(acc1[4]), "v"(acc1[5]), "v"(acc1[6]), "v"(acc1[7]));
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[i]) : "v"(dgs_addr), "n"((c8 * NB + 4 * i) * 4) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                     "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rw = c8 + i;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          acc0[b] = fmaf(w[rw][0], d[2 * i][b], acc0[b]);             acc1[b] = fmaf(w[rw][1], d[2 * i][b], acc1[b]);
          acc0[4 + b] = fmaf(w[rw][0], d[2 * i + 1][b], acc0[4 + b]); acc1[4 + b] = fmaf(w[rw][1], d[2 * i + 1][b], acc1[4 + b]);
        }
      }
    }
    // hand the other half the four batch slots it finishes: half 0 keeps slots 0-3, half 1 keeps 4-7
    {
      const int give = 4 * (1 - q);
      f32x4 g0 = {acc0[give], acc0[give + 1], acc0[give + 2], acc0[give + 3]};
      f32x4 g1 = {acc1[give], acc1[give + 1], acc1[give + 2], acc1[give + 3]};
      *reinterpret_cast<f32x4*>(&swp[q][c][0]) = g0;
      *reinterpret_cast<f32x4*>(&swp[q][c][4]) = g1;
    }
    __syncthreads();
    const f32x4 o0 = *reinterpret_cast<const f32x4*>(&swp[1 - q][c][0]), o1 = *reinterpret_cast<const f32x4*>(&swp[1 - q][c][4]);
    unsigned long long* dst = a.xp + (size_t)(t % BWD_RING) * XP_SLOT + pub;
    const unsigned long long tag = (unsigned long long)(a.tagbase | (unsigned)(t + 1)) << 32;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const int b = 4 * q + bb;
      if (b < a.N) {
        // (the two halves' sums are added lower half first: the value does not depend on which thread adds)
        const float v0 = q == 0 ? acc0[b] + o0[bb] : o0[bb] + acc0[b];
        const float v1 = q == 0 ? acc1[b] + o1[bb] : o1[bb] + acc1[b];
        __hip_atomic_store(dst + b * UNITS_WG, tag | __float_as_uint(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + b * UNITS_WG + 1, tag | __float_as_uint(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (a.out_cnt && t % a.Tc == 0) chain_signal(a.out_cnt + t / a.Tc, tid);   // dgi / dgh of this chunk are complete (this workgroup's part)
  }
}

}  // namespace

// workspace layout: [0, 256) barrier words (zeroed per call) | [256, ...) exchange image (128-B aligned
// when the workspace is; rows of one workgroup never share a cache line with another workgroup's)
// CU ownership (round 2) and what replaced it (round 3).  With a bf16 implicit-GEMM conv (MFMA) workgroup co-resident on the
// same SIMDs, gru_bwd_kernel returned wrong partial sums in isolated lanes — always the low half of a v_pk_fma_f32 pair; memory
// contents and the shuffle reduction were verified identical — on every run, and never when it owned the CU.  Round 2 shipped
// the workaround: requesting (almost) the CU's whole 160 KiB LDS makes co-residency with any LDS-using workgroup impossible.
// Round 3 removed the trigger instead: this file is compiled WITHOUT packed-float32 instructions (csrc/Makefile:
// -target-feature -packed-fp32-ops; the compiler had SLP-packed 425 of them) — tools/stress_rnn.py under the bf16 conv load,
// no CU claim: 177 mismatching tensors in 60 repeats with them, 0 in 200 without, same step times (2.03 vs 2.06 ms per
// GRU + LSTM forward + backward).  Whether the packed forms hit a hardware hazard beside MFMA-heavy waves or a missing
// dependency stall in their scheduling cannot be told from the ISA; the measured facts are in DESIGN.md ("RNN kernels: CU
// ownership").  (The claim itself, and its two attribute calls per launch, were removed in round 4.)
// launch-unique tag bits for the flag-in-data exchange (22-bit epoch above the 10-bit step number)
// Process-wide status word of the persistent kernels in host-mapped pinned memory: bit 0 gru_fwd, 1 gru_bwd, 2 lstm_fwd,
// 3 lstm_bwd timed out.  The host polls it without synchronising (wsmg_rnn_status).
static unsigned* g_status_host = nullptr;
static unsigned* g_status_dev = nullptr;
static unsigned g_spin = SPIN_LIMIT;
static unsigned* rnn_status_dev() {
  if (!g_status_host) {
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return nullptr;
    *(volatile unsigned*)h = 0u;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return nullptr; }
    g_status_host = (unsigned*)h;
    g_status_dev = (unsigned*)d;
  }
  return g_status_dev;
}
extern "C" int wsmg_rnn_status(int clear) {
  if (!g_status_host) return 0;
  unsigned v = __atomic_load_n(g_status_host, __ATOMIC_ACQUIRE);
  if (clear && v) __atomic_and_fetch(g_status_host, ~v, __ATOMIC_ACQ_REL);
  return (int)v;
}
extern "C" int wsmg_rnn_debug_inject(unsigned bits) {
  if (!rnn_status_dev()) return 0;
  return (int)__atomic_or_fetch(g_status_host, bits, __ATOMIC_ACQ_REL);
}
extern "C" int wsmg_rnn_debug_spin_limit(unsigned limit) {
  g_spin = limit ? limit : SPIN_LIMIT;
  return 0;
}

static unsigned next_tagbase() {
  static unsigned epoch = 0;
  return (__atomic_add_fetch(&epoch, 1u, __ATOMIC_RELAXED) & 0x3FFFFFu) << 10;
}

// 256 B of control words + the larger of: forward image (per step 4096 (batch, unit) {value, tag} words = 32 KB),
// backward ring (BWD_RING step slots of 32 consumers x 4096 words = 4 MB)
extern "C" int64_t wsmg_gru_workspace_bytes(int T) {
  const int64_t fwd = (int64_t)T * NWG * NB * UNITS_WG * 8, bwd = (int64_t)BWD_RING * XP_SLOT * 8;
  return 256 + (fwd > bwd ? fwd : bwd);
}

static int gru_fwd_launch(const float* gi, const float* w_hh, const float* b_hh, const float* h0, const float* masks, int T, int N, int hidden,
                          float* y, float* save_r, float* save_z, float* save_n, float* save_ghn, void* sync_ws, hipStream_t s, bool clear,
                          int Tc = 0, const unsigned* in_cnt = nullptr, unsigned in_target = 0, unsigned* out_cnt = nullptr) {
  if (hidden != H || T <= 0 || N <= 0 || N > NB) return WSMG_EINVAL;
  if (((uintptr_t)sync_ws & 127) != 0) return WSMG_EINVAL;
  if (T > 1023) return WSMG_EINVAL;
  hipError_t e = hipSuccess;
  // the control words AND the {value, tag} image are cleared: tags are launch-unique within a process, but device
  // memory handed to a new process can still hold a previous process's image with the same epoch numbers
  if (clear && (e = hipMemsetAsync(sync_ws, 0, 256 + (size_t)T * NWG * NB * UNITS_WG * 8, s)) != hipSuccess) return (int)e;
  GruFwdArgs a{gi, w_hh, b_hh, h0, masks, y, save_r, save_z, save_n, save_ghn, (unsigned*)sync_ws,
               (unsigned long long*)((char*)sync_ws + 256), T, N, next_tagbase(), rnn_status_dev(), g_spin, Tc, in_cnt, in_target, out_cnt};
  hipLaunchKernelGGL(gru_fwd8_kernel, dim3(NWG), dim3(512), 0, s, a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_gru_fwd(const float* gi, const float* w_hh, const float* b_hh, const float* h0, const float* masks,
                            int T, int N, int hidden, float* y, float* save_r, float* save_z, float* save_n,
                            float* save_ghn, void* sync_ws, wsmg_stream_t stream) {
  return gru_fwd_launch(gi, w_hh, b_hh, h0, masks, T, N, hidden, y, save_r, save_z, save_n, save_ghn, sync_ws, wsmg_s(stream), true);
}

// The same launch on a workspace the CALLER owns for good: zeroed once when it was allocated and never used by anything but this
// process's wsmg_gru_*_owned calls.  Tags are launch-unique within a process, so words left by earlier launches can never match:
// the per-launch clear (one more launch in front of every one of the 16 chunk launches of the pipelined update) is not needed.
// After a reported timeout (wsmg_rnn_status) the owner must zero the workspace again: the error word in it is sticky.
extern "C" int wsmg_gru_fwd_owned(const float* gi, const float* w_hh, const float* b_hh, const float* h0, const float* masks,
                                  int T, int N, int hidden, float* y, float* save_r, float* save_z, float* save_n,
                                  float* save_ghn, void* sync_ws, wsmg_stream_t stream) {
  return gru_fwd_launch(gi, w_hh, b_hh, h0, masks, T, N, hidden, y, save_r, save_z, save_n, save_ghn, sync_ws, wsmg_s(stream), false);
}

static int gru_bwd_launch(const float* dy, const float* dhT, const float* w_hh, const float* h0, const float* masks, const float* y,
                          const float* save_r, const float* save_z, const float* save_n, const float* save_ghn, int T, int N, int hidden,
                          float* dgi, float* dgh, float* dh0, void* sync_ws, hipStream_t s, bool clear,
                          int Tc = 0, const unsigned* in_cnt = nullptr, unsigned in_target = 0, unsigned* out_cnt = nullptr) {
  if (hidden != H || T <= 0 || N <= 0 || N > NB) return WSMG_EINVAL;
  if (((uintptr_t)sync_ws & 127) != 0) return WSMG_EINVAL;
  if (T > 1023) return WSMG_EINVAL;
  hipError_t e = hipSuccess;
  // control words and the ring of {value, tag} words are cleared (see wsmg_gru_fwd)
  if (clear && (e = hipMemsetAsync(sync_ws, 0, 256 + (size_t)BWD_RING * XP_SLOT * 8, s)) != hipSuccess) return (int)e;
  GruBwdArgs a{dy, dhT, w_hh, h0, masks, y, save_r, save_z, save_n, save_ghn, dgi, dgh, dh0, (unsigned*)sync_ws,
               (unsigned long long*)((char*)sync_ws + 256), T, N, next_tagbase(), rnn_status_dev(), g_spin, Tc, in_cnt, in_target, out_cnt};
  hipLaunchKernelGGL(gru_bwd8_kernel, dim3(NWG), dim3(512), 0, s, a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_gru_bwd(const float* dy, const float* dhT, const float* w_hh, const float* h0, const float* masks,
                            const float* y, const float* save_r, const float* save_z, const float* save_n,
                            const float* save_ghn, int T, int N, int hidden, float* dgi, float* dgh, float* dh0,
                            void* sync_ws, wsmg_stream_t stream) {
  return gru_bwd_launch(dy, dhT, w_hh, h0, masks, y, save_r, save_z, save_n, save_ghn, T, N, hidden, dgi, dgh, dh0, sync_ws, wsmg_s(stream), true);
}

extern "C" int wsmg_gru_bwd_owned(const float* dy, const float* dhT, const float* w_hh, const float* h0, const float* masks,
                                  const float* y, const float* save_r, const float* save_z, const float* save_n,
                                  const float* save_ghn, int T, int N, int hidden, float* dgi, float* dgh, float* dh0,
                                  void* sync_ws, wsmg_stream_t stream) {
  return gru_bwd_launch(dy, dhT, w_hh, h0, masks, y, save_r, save_z, save_n, save_ghn, T, N, hidden, dgi, dgh, dh0, sync_ws, wsmg_s(stream), false);
}

// Whole-sequence launches chained by time chunk to kernels on other streams (round 5; see chain_wait): steps_per_chunk divides T;
// in_count (may be NULL): one counter per chunk that must reach in_target before the chunk's inputs (gi, resp. dy) are read;
// out_count (may be NULL): one counter per chunk this launch adds its 32 workgroups' arrivals to when the chunk's outputs (y and
// the saved gates, resp. dgi / dgh) are complete.  Counters are the caller's, zeroed before the pass.  Workspace as *_owned.
extern "C" int wsmg_gru_fwd_chain(const float* gi, const float* w_hh, const float* b_hh, const float* h0, const float* masks,
                                  int T, int N, int hidden, float* y, float* save_r, float* save_z, float* save_n,
                                  float* save_ghn, void* sync_ws, int steps_per_chunk, const unsigned* in_count, unsigned in_target,
                                  unsigned* out_count, wsmg_stream_t stream) {
  if (steps_per_chunk <= 0 || T % steps_per_chunk) return WSMG_EINVAL;
  return gru_fwd_launch(gi, w_hh, b_hh, h0, masks, T, N, hidden, y, save_r, save_z, save_n, save_ghn, sync_ws, wsmg_s(stream), false,
                        steps_per_chunk, in_count, in_target, out_count);
}
extern "C" int wsmg_gru_bwd_chain(const float* dy, const float* dhT, const float* w_hh, const float* h0, const float* masks,
                                  const float* y, const float* save_r, const float* save_z, const float* save_n,
                                  const float* save_ghn, int T, int N, int hidden, float* dgi, float* dgh, float* dh0,
                                  void* sync_ws, int steps_per_chunk, const unsigned* in_count, unsigned in_target, unsigned* out_count,
                                  wsmg_stream_t stream) {
  if (steps_per_chunk <= 0 || T % steps_per_chunk) return WSMG_EINVAL;
  return gru_bwd_launch(dy, dhT, w_hh, h0, masks, y, save_r, save_z, save_n, save_ghn, T, N, hidden, dgi, dgh, dh0, sync_ws, wsmg_s(stream), false,
                        steps_per_chunk, in_count, in_target, out_count);
}
extern "C" int wsmg_gru_chain_workgroups(void) { return NWG; }
// (wsmg_rows_gemm_f32's wait reports a timeout through the same process-wide status word)
__attribute__((visibility("hidden"))) unsigned* wsmgi_rnn_status_dev() { return rnn_status_dev(); }

// =================================================================================================
// Persistent packed bidirectional LSTM (instruction encoder, hidden 128 per direction, U <= 8 unique
// instructions per launch).  Replaces the cuDNN/MIOpen packed-sequence LSTM behind nn.LSTM at
// instruction_encoder.py:80-92 (about 10 tiny launches per token step and direction in MIOpen).
// Same structure as the GRU kernels: per direction 8 workgroups x 4 waves, each wave keeps the
// 16 gate rows (i,f,g,o of 4 hidden units) of W_hh in registers, lanes split K = 128, one bounded
// grid barrier per token step per direction; both directions run concurrently in one launch.
// Packed-sequence semantics: row b is active at step t iff t < len[b]; inactive steps freeze the
// state and emit 0 (forward direction: after the end; reverse direction: before its first token).
// Gate order i, f, g, o (PyTorch nn.LSTM).  gi = x W_ih^T + b_ih for both directions is one GEMM
// done by the caller; so are dW_hh / dW_ih.
namespace {

constexpr int LH = 128;            // hidden per direction
constexpr int L_NWG = LH / UNITS_WG;   // 8 workgroups per direction

struct LstmFwdArgs {
  const float* gi;     // [U][L][2][4*LH]
  const float* whh;    // [2][4*LH][LH]
  const float* bhh;    // [2][4*LH]
  const int* len;      // [U]
  float* out;          // [U][L][2*LH]
  float* hs;           // exchange [2 dir][L][L_NWG][NB][UNITS_WG]: each line written once by one workgroup
  float* sg;           // [2][U][L][4][LH] saved gates i,f,g,o
  float* sc;           // [2][U][L][LH]    saved cell state c_t
  unsigned* sync;      // per direction 16 words: [0] counter, [1] error
  int U, L;
  unsigned* status;    // host-mapped process status word (rnn_fail)
  unsigned spin;
};

__global__ __launch_bounds__(256) void lstm_fwd_kernel(LstmFwdArgs a) {
  __shared__ int ok_lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dir = blockIdx.x / L_NWG;
  const int u0 = (blockIdx.x % L_NWG) * UNITS_WG + wave * UNITS_WAVE;
  const float* whh = a.whh + (size_t)dir * 4 * LH * LH;
  const float* bhh = a.bhh + dir * 4 * LH;
  unsigned* sync = a.sync + dir * 16;
  // rows r = gate*4 + unit; k = 2*lane + e
  float w[16][2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r >> 2) * LH + u0 + (r & 3);
    w[r][0] = whh[(size_t)row * LH + 2 * lane];
    w[r][1] = whh[(size_t)row * LH + 2 * lane + 1];
  }
  const int my_unit = u0 + ((lane >> 2) & 3);
  const int my_b0 = 2 * (lane & 3);
  float bi = 0.f, bf = 0.f, bg = 0.f, bo = 0.f;
  if (lane < 16) { bi = bhh[my_unit]; bf = bhh[LH + my_unit]; bg = bhh[2 * LH + my_unit]; bo = bhh[3 * LH + my_unit]; }
  int mylen[2] = {0, 0};
  if (lane < 16) {
    if (my_b0 < a.U) mylen[0] = a.len[my_b0];
    if (my_b0 + 1 < a.U) mylen[1] = a.len[my_b0 + 1];
  }
  float c[2] = {0.f, 0.f};
  float* hs = a.hs + (size_t)dir * a.L * NB * LH;  // step-indexed image [L][L_NWG][NB][UNITS_WG]
  const int wgi = blockIdx.x % L_NWG;
  const int xk = ((2 * lane) >> 4) * NB * UNITS_WG + ((2 * lane) & 15);          // + b*16
  const int xw = wgi * NB * UNITS_WG + wave * UNITS_WAVE + ((lane >> 2) & 3);    // + b*16

  for (int s = 0; s < a.L; ++s) {
    const int t = dir == 0 ? s : a.L - 1 - s;
    const float* hprev = hs + (size_t)(s > 0 ? s - 1 : 0) * NB * LH;   // slot written in step s-1
    float* hnext = hs + (size_t)s * NB * LH;
    float hp[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      hp[b][0] = s > 0 ? hprev[xk + b * UNITS_WG] : 0.f;       // initial state is zero
      hp[b][1] = s > 0 ? hprev[xk + b * UNITS_WG + 1] : 0.f;
    }
    float acc[128];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[r * 8 + b] = fmaf(w[r][1], hp[b][1], w[r][0] * hp[b][0]);
    halve<128, 32>(acc, lane);
    halve<64, 16>(acc, lane);
    halve<32, 8>(acc, lane);
    halve<16, 4>(acc, lane);
    halve<8, 2>(acc, lane);
    halve<4, 1>(acc, lane);
    float f0 = __shfl(acc[0], lane + 16, 64), f1 = __shfl(acc[1], lane + 16, 64);
    float g0 = __shfl(acc[0], lane + 32, 64), g1 = __shfl(acc[1], lane + 32, 64);
    float o0 = __shfl(acc[0], lane + 48, 64), o1 = __shfl(acc[1], lane + 48, 64);
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int b = my_b0 + i;
        if (b < a.U) {
          const float hold = s > 0 ? hprev[xw + b * UNITS_WG] : 0.f;
          const bool active = t < mylen[i];
          float hnew = hold;
          float outv = 0.f;
          if (active) {
            const float* g = a.gi + (((size_t)b * a.L + t) * 2 + dir) * 4 * LH;
            float gi_ = sigmoidf_(g[my_unit] + (i ? acc[1] : acc[0]) + bi);
            float gf_ = sigmoidf_(g[LH + my_unit] + (i ? f1 : f0) + bf);
            float gg_ = tanhf(g[2 * LH + my_unit] + (i ? g1 : g0) + bg);
            float go_ = sigmoidf_(g[3 * LH + my_unit] + (i ? o1 : o0) + bo);
            c[i] = gf_ * c[i] + gi_ * gg_;
            hnew = go_ * tanhf(c[i]);
            outv = hnew;
            float* sgp = a.sg + ((((size_t)dir * a.U + b) * a.L + t) * 4) * LH + my_unit;
            sgp[0] = gi_; sgp[LH] = gf_; sgp[2 * LH] = gg_; sgp[3 * LH] = go_;
            a.sc[(((size_t)dir * a.U + b) * a.L + t) * LH + my_unit] = c[i];
          }
          hnext[xw + b * UNITS_WG] = hnew;
          a.out[((size_t)b * a.L + t) * 2 * LH + dir * LH + my_unit] = outv;
        }
      }
    }
    if (s + 1 < a.L) {
      if (!grid_barrier(sync, (unsigned)L_NWG * (unsigned)(s + 1), tid, &ok_lds, a.spin)) {
        rnn_fail(a.status, 4u);
        rnn_poison(a.out, (size_t)a.U * a.L, 2 * LH, dir * LH + wgi * UNITS_WG, UNITS_WG);
        return;
      }
    }
  }
}

struct LstmBwdArgs {
  const float* dout;   // [U][L][2*LH]
  const float* whh;    // [2][4*LH][LH]
  const int* len;      // [U]
  const float* sg;     // [2][U][L][4][LH]
  const float* sc;     // [2][U][L][LH]
  float* dg;           // [U][L][2][4*LH]  gradient of the gate pre-activations (= d gi = d gh)
  unsigned* sync;
  float* xg;           // exchange [2 dir][L][L_NWG][NB][4][UNITS_WG]: each line written once
  int U, L;
  unsigned* status;    // host-mapped process status word (rnn_fail)
  unsigned spin;
};

__global__ __launch_bounds__(256) void lstm_bwd_kernel(LstmBwdArgs a) {
  __shared__ int ok_lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dir = blockIdx.x / L_NWG;
  const int u0 = (blockIdx.x % L_NWG) * UNITS_WG + wave * UNITS_WAVE;
  const float* whh = a.whh + (size_t)dir * 4 * LH * LH;
  unsigned* sync = a.sync + dir * 16;
  // columns u0..u0+3 of W_hh over the 4*LH gate rows: k = 256*q + 4*lane + e
  float wt[4][8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 256 * q + 4 * lane + e;
#pragma unroll
      for (int u = 0; u < 4; ++u) wt[u][q * 4 + e] = whh[(size_t)row * LH + u0 + u];
    }
  const int my_unit = u0 + (lane >> 4);
  const int my_b = (lane >> 1) & 7;
  const bool worker = ((lane & 1) == 0) && (my_b < a.U);
  const int mylen = worker ? a.len[my_b] : 0;
  float carry_h = 0.f, carry_c = 0.f;
  constexpr int LXG_WG = NB * 4 * UNITS_WG;
  const int wgi = blockIdx.x % L_NWG;
  const int xgw = wgi * LXG_WG + my_b * 4 * UNITS_WG + wave * UNITS_WAVE + (lane >> 4);   // + gate*16

  for (int s = a.L - 1; s >= 0; --s) {
    const int t = dir == 0 ? s : a.L - 1 - s;
    float* xcur = a.xg + ((size_t)dir * a.L + s) * L_NWG * LXG_WG;
    float dh_direct = 0.f;
    if (worker) {
      float* dgp = a.dg + (((size_t)my_b * a.L + t) * 2 + dir) * 4 * LH + my_unit;
      if (t < mylen) {
        const size_t o = (((size_t)dir * a.U + my_b) * a.L + t);
        const float* sgp = a.sg + o * 4 * LH + my_unit;
        float gi_ = sgp[0], gf_ = sgp[LH], gg_ = sgp[2 * LH], go_ = sgp[3 * LH];
        float cn = a.sc[o * LH + my_unit];
        // previous cell state in processing order (0 at the first active step)
        const int tp = dir == 0 ? t - 1 : t + 1;
        float cp = (tp >= 0 && tp < mylen) ? a.sc[((((size_t)dir * a.U + my_b) * a.L + tp)) * LH + my_unit] : 0.f;
        float dh = a.dout[((size_t)my_b * a.L + t) * 2 * LH + dir * LH + my_unit] + carry_h;
        float tc = tanhf(cn);
        float do_pre = dh * tc * go_ * (1.0f - go_);
        float dc = dh * go_ * (1.0f - tc * tc) + carry_c;
        float di_pre = dc * gg_ * gi_ * (1.0f - gi_);
        float df_pre = dc * cp * gf_ * (1.0f - gf_);
        float dg_pre = dc * gi_ * (1.0f - gg_ * gg_);
        carry_c = dc * gf_;
        dgp[0] = di_pre; dgp[LH] = df_pre; dgp[2 * LH] = dg_pre; dgp[3 * LH] = do_pre;
        xcur[xgw] = di_pre; xcur[xgw + UNITS_WG] = df_pre; xcur[xgw + 2 * UNITS_WG] = dg_pre; xcur[xgw + 3 * UNITS_WG] = do_pre;
      } else {
        dgp[0] = 0.f; dgp[LH] = 0.f; dgp[2 * LH] = 0.f; dgp[3 * LH] = 0.f;
        xcur[xgw] = 0.f; xcur[xgw + UNITS_WG] = 0.f; xcur[xgw + 2 * UNITS_WG] = 0.f; xcur[xgw + 3 * UNITS_WG] = 0.f;
        dh_direct = carry_h;  // frozen state: gradient passes straight through
      }
    }
    if (!grid_barrier(sync, (unsigned)L_NWG * (unsigned)(a.L - s), tid, &ok_lds, a.spin)) {
      rnn_fail(a.status, 8u);
      for (int g = 0; g < 4; ++g)
        rnn_poison(a.dg, (size_t)a.U * a.L, 2 * 4 * LH, dir * 4 * LH + g * LH + wgi * UNITS_WG, UNITS_WG);
      return;
    }
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b < a.U) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int k = 256 * q + 4 * lane;          // gate row -> gate k/128, unit k%128
          const int gate = k >> 7, unit = k & 127;
          f32x4 g = *reinterpret_cast<const f32x4*>(xcur + (unit >> 4) * LXG_WG + b * 4 * UNITS_WG + gate * UNITS_WG + (unit & 15));
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u * 8 + b] = fmaf(wt[u][q * 4 + e], g[e], acc[u * 8 + b]);
        }
      }
    }
    halve<32, 32>(acc, lane);
    halve<16, 16>(acc, lane);
    halve<8, 8>(acc, lane);
    halve<4, 4>(acc, lane);
    halve<2, 2>(acc, lane);
    float sum = acc[0] + __shfl_xor(acc[0], 1, 64);
    carry_h = dh_direct + sum;
  }
}

}  // namespace

// per token and direction: forward image NB*LH floats (4 KB), backward image 4x that (16 KB)
extern "C" int64_t wsmg_lstm_workspace_bytes(int L) { return 256 + (int64_t)2 * L * L_NWG * NB * 4 * UNITS_WG * 4; }

extern "C" int wsmg_lstm_fwd(const float* gi, const float* w_hh, const float* b_hh, const int32_t* lengths, int U, int L,
                             int hidden, float* out, float* save_gates, float* save_c, void* state_ws,
                             wsmg_stream_t stream) {
  if (hidden != LH || U <= 0 || U > NB || L <= 0) return WSMG_EINVAL;
  hipStream_t s = wsmg_s(stream);
  if (((uintptr_t)state_ws & 127) != 0) return WSMG_EINVAL;
  hipError_t e = hipMemsetAsync(state_ws, 0, 256, s);   // barrier words
  if (e != hipSuccess) return (int)e;
  LstmFwdArgs a{gi, w_hh, b_hh, lengths, out, (float*)((char*)state_ws + 256), save_gates, save_c,
                (unsigned*)state_ws, U, L, rnn_status_dev(), g_spin};
  hipLaunchKernelGGL(lstm_fwd_kernel, dim3(2 * L_NWG), dim3(256), 0, s, a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_lstm_bwd(const float* dout, const float* w_hh, const int32_t* lengths, const float* save_gates,
                             const float* save_c, int U, int L, int hidden, float* dgates, void* state_ws,
                             wsmg_stream_t stream) {
  if (hidden != LH || U <= 0 || U > NB || L <= 0) return WSMG_EINVAL;
  hipStream_t s = wsmg_s(stream);
  if (((uintptr_t)state_ws & 127) != 0) return WSMG_EINVAL;
  hipError_t e = hipMemsetAsync(state_ws, 0, 256, s);
  if (e != hipSuccess) return (int)e;
  LstmBwdArgs a{dout, w_hh, lengths, save_gates, save_c, dgates, (unsigned*)state_ws, (float*)((char*)state_ws + 256), U, L, rnn_status_dev(), g_spin};
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3(2 * L_NWG), dim3(256), 0, s, a);
  WSMG_RETURN_LAUNCH();
}
