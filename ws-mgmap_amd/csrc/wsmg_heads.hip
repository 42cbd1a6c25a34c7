// The update path's heads, auxiliary-loss reduction and trainer loss as five launches instead of ≈70.
//
// After the second GRU the reference's update step (models/policy.py:91-103, 58-89; common/aux_losses.py:24-35;
// dagger_trainer.py:526-534) is a tail of tiny operators on [B, 512] features and [B] loss vectors: the action mean
// (Linear 512 -> A), the progress head (Linear 512 -> 1, tanh) with its squared error, the masked mean of the registered
// auxiliary losses, tanh / squared error / weight normalisation of the DAgger loss — ≈35 element-wise and reduction launches
// forward and as many backward, each 2-6 us of dependent latency between the two recurrences' forward and backward
// (0.31 ms of a 12.2 ms update at B = 512, measured with HIP events).  Here:
//
//   update_heads_fwd_kernel   pred = features Wm^T + bm,  prog = tanh(features Wp^T + bp),  prog_rows = (prog - progress)^2
//   aux_reduce_fwd_kernel     aux = sum_k alpha_k * sum_b [mask_b] loss_k[b] / sum_b [mask_b]        (aux_losses.py:24-35)
//   dagger_loss_fwd_kernel    loss = mean_n( sum_t w[t,n] |tanh(pred[t,n]) - wp[t,n]|^2 / sum_t w[t,n] ) + aux
//   dagger_loss_bwd_kernel    d pred, (d aux = d loss)
//   update_heads_bwd_kernel   d loss rows, d features, d Wm, d bm, d Wp, d bp — every sum in a fixed order (no atomics)
//
// float32 throughout; the masked rows are DROPPED (selected), not multiplied by zero, as the reference's masked_select does.
#include "wsmg_common.h"

namespace {

constexpr int MAXA = 4;     // action dimensions (the reference's waypoint head has 2)
constexpr int MAXL = 4;     // auxiliary loss vectors in one reduction (the reference registers 3)

struct HeadsFwd {
  const float* x;      // [B][K]
  const float* wm;     // [A][K]
  const float* bm;     // [A]
  const float* wp;     // [1][K]
  const float* bp;     // [1]
  const float* progress;   // [B] or null
  float* pred;         // [B][A]
  float* prog;         // [B]
  float* prog_rows;    // [B] or null
  int B, K, A;
};

// one wave per row: lane l holds 8 consecutive features per 512-feature pass
__global__ __launch_bounds__(256) void update_heads_fwd_kernel(HeadsFwd a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + wave;
  if (b >= a.B) return;
  float s[MAXA + 1];
#pragma unroll
  for (int o = 0; o <= MAXA; ++o) s[o] = 0.f;
  const float* __restrict__ xr = a.x + (size_t)b * a.K;
  for (int k = lane * 4; k < a.K; k += 256) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + k);
#pragma unroll
    for (int o = 0; o < MAXA; ++o)
      if (o < a.A) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(a.wm + (size_t)o * a.K + k);
        s[o] += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
      }
    const f32x4 pv = *reinterpret_cast<const f32x4*>(a.wp + k);
    s[MAXA] += xv[0] * pv[0] + xv[1] * pv[1] + xv[2] * pv[2] + xv[3] * pv[3];
  }
#pragma unroll
  for (int o = 0; o <= MAXA; ++o) s[o] = wave_sum(s[o]);
  if (lane == 0) {
#pragma unroll
    for (int o = 0; o < MAXA; ++o)
      if (o < a.A) a.pred[(size_t)b * a.A + o] = s[o] + a.bm[o];
    const float p = tanhf(s[MAXA] + a.bp[0]);
    a.prog[b] = p;
    if (a.prog_rows) {
      const float d = p - a.progress[b];
      a.prog_rows[b] = d * d;
    }
  }
}

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads; result valid in every thread
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

struct AuxReduce {
  const float* rows[MAXL];   // [B] each
  float alpha[MAXL];
  const unsigned char* mask; // [B] (bool)
  float* out;                // [2]: aux, number of selected rows
  int B, L;
};

__global__ __launch_bounds__(256) void aux_reduce_fwd_kernel(AuxReduce a) {
  __shared__ float red[4];
  float s[MAXL], n = 0.f;
#pragma unroll
  for (int k = 0; k < MAXL; ++k) s[k] = 0.f;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    const bool m = a.mask[b] != 0;
    n += m ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < MAXL; ++k)
      if (k < a.L) s[k] += m ? a.rows[k][b] : 0.f;
  }
  n = block_sum(n, red);
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < MAXL; ++k)
    if (k < a.L) tot += block_sum(s[k], red) * a.alpha[k];
  if (threadIdx.x == 0) {
    a.out[0] = tot / n;     // (an empty selection is NaN, as masked_select(...).mean() is)
    a.out[1] = n;
  }
}

struct DaggerLoss {
  const float* pred;      // [T*N][A]
  const float* waypoint;  // [T*N][ld_wp], the first A columns are used
  const float* weights;   // [T][N]
  const float* aux;       // [1] or null
  float* out;             // [2]: loss, action loss
  float* den;             // [N]: sum_t w[t][n] (kept for the backward)
  int T, N, A, ld_wp;
};

// One workgroup.  Phase 1, all threads: the weighted squared error and the weight of every row (t, n) of a chunk into LDS; phase 2,
// thread n: its episode's rows of the chunk added in step order (the order of the sums never depends on the thread count).
constexpr int DL_CHUNK = 2048;   // rows per chunk
__global__ __launch_bounds__(256) void dagger_loss_fwd_kernel(DaggerLoss a) {
  __shared__ float red[4];
  __shared__ float term[DL_CHUNK], wrow[DL_CHUNK];
  const int rows = a.T * a.N;
  // chunks hold whole steps: tpc steps of N rows (N <= DL_CHUNK is checked by the host)
  const int tpc = DL_CHUNK / a.N;
  float num = 0.f, den = 0.f;        // thread n < N: its episode
  for (int t0 = 0; t0 < a.T; t0 += tpc) {
    const int nt = a.T - t0 < tpc ? a.T - t0 : tpc;
    for (int i = threadIdx.x; i < nt * a.N; i += 256) {
      const size_t r = (size_t)t0 * a.N + i;
      float al = 0.f;
      for (int j = 0; j < a.A; ++j) {
        const float d = tanhf(a.pred[r * a.A + j]) - a.waypoint[r * a.ld_wp + j];
        al += d * d;
      }
      const float w = a.weights[r];
      term[i] = w * al;
      wrow[i] = w;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < a.N; n += 256)      // (N <= 256: one episode per thread)
      for (int t = 0; t < nt; ++t) { num += term[t * a.N + n]; den += wrow[t * a.N + n]; }
    __syncthreads();
  }
  (void)rows;
  float tot = 0.f;
  if (threadIdx.x < a.N) {
    a.den[threadIdx.x] = den;
    tot = num / den;
  }
  tot = block_sum(tot, red);
  if (threadIdx.x == 0) {
    const float action = tot / (float)a.N;
    a.out[1] = action;
    a.out[0] = action + (a.aux ? a.aux[0] : 0.f);
  }
}

__global__ __launch_bounds__(256) void dagger_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ waypoint,
                                                              const float* __restrict__ weights, const float* __restrict__ den,
                                                              const float* __restrict__ dloss, float* __restrict__ dpred, int T, int N,
                                                              int A, int ld_wp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)T * N * A) return;
  const int64_t r = i / A;
  const int j = (int)(i - r * A), n = (int)(r % N);
  const float lg = tanhf(pred[i]);
  const float g = dloss[0] / (float)N * (weights[r] / den[n]);
  dpred[i] = g * 2.f * (lg - waypoint[r * ld_wp + j]) * (1.f - lg * lg);
}

struct HeadsBwd {
  const float* x;        // [B][K]
  const float* wm;       // [A][K]
  const float* wp;       // [K]
  const float* prog;     // [B]
  const float* progress; // [B] or null (no progress loss)
  const float* dpred;    // [B][A] or null
  const float* dprog;    // [B] or null: a gradient on prog from outside the auxiliary loss
  const float* dprows;   // [B] or null: gradient of the progress loss rows
  float* dx;             // [B][K]
  float* dwm;            // [A][K]
  float* dbm;            // [A]
  float* dwp;            // [K]
  float* dbp;            // [1]
  int B, K, A;
};

// A workgroup owns 32 features (columns) for ALL rows: thread = (feature f, row group g of 8); row b belongs to group b % 8.
// d x[b][f] = sum_o g_o[b] w_o[f] is written directly; d w_o[f] = sum_b g_o[b] x[b][f] is accumulated per thread over its rows
// in row order and the 32 groups are added in group order — no atomics, the same bits every run.  g_o[b] (A + 1 numbers per row)
// is recomputed by every workgroup: it is a handful of flops.
__global__ __launch_bounds__(256) void update_heads_bwd_kernel(HeadsBwd a) {
  // (round 5, end: 8 feature columns x 32 row groups per workgroup instead of 32 x 8 — 64 workgroups of 16-row loops instead of 16 of
  //  64-row loops: the kernel sits alone between the recurrent core's forward and backward, 45 us of dependent loads)
  constexpr int FC = 8, RG = 32;
  __shared__ float part[RG][MAXA + 1][FC];
  const int f = threadIdx.x & (FC - 1), g = threadIdx.x / FC;
  const int col = blockIdx.x * FC + f;
  const bool live = col < a.K;
  float wv[MAXA + 1], acc[MAXA + 1], bacc[MAXA + 1];
#pragma unroll
  for (int o = 0; o <= MAXA; ++o) { wv[o] = 0.f; acc[o] = 0.f; bacc[o] = 0.f; }
  if (live) {
#pragma unroll
    for (int o = 0; o < MAXA; ++o)
      if (o < a.A) wv[o] = a.wm[(size_t)o * a.K + col];
    wv[MAXA] = a.wp[col];
  }
#pragma unroll 4
  for (int b = g; b < a.B; b += RG) {       // (independent rows: several in flight)
    float go[MAXA + 1];
#pragma unroll
    for (int o = 0; o < MAXA; ++o) go[o] = (o < a.A && a.dpred) ? a.dpred[(size_t)b * a.A + o] : 0.f;
    const float p = a.prog[b];
    float gp = a.dprog ? a.dprog[b] : 0.f;
    if (a.dprows && a.progress) gp += a.dprows[b] * 2.f * (p - a.progress[b]);
    go[MAXA] = gp * (1.f - p * p);
    if (live) {
      const float xv = a.x[(size_t)b * a.K + col];
      float d = 0.f;
#pragma unroll
      for (int o = 0; o <= MAXA; ++o) {
        d += go[o] * wv[o];
        acc[o] += go[o] * xv;
      }
      a.dx[(size_t)b * a.K + col] = d;
    }
#pragma unroll
    for (int o = 0; o <= MAXA; ++o) bacc[o] += go[o];
  }
#pragma unroll
  for (int o = 0; o <= MAXA; ++o) part[g][o][f] = acc[o];
  __syncthreads();
  if (g == 0 && live) {
#pragma unroll
    for (int o = 0; o <= MAXA; ++o) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < RG; ++q) s += part[q][o][f];
      if (o < MAXA) { if (o < a.A) a.dwm[(size_t)o * a.K + col] = s; }
      else a.dwp[col] = s;
    }
  }
  if (blockIdx.x == 0) {      // bias gradients: the groups' row sums, in group order (every feature lane holds the same numbers)
    __syncthreads();
#pragma unroll
    for (int o = 0; o <= MAXA; ++o) part[g][o][f] = bacc[o];
    __syncthreads();
    if (threadIdx.x <= MAXA) {
      const int o = threadIdx.x;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < RG; ++q) s += part[q][o][0];
      if (o < MAXA) { if (o < a.A) a.dbm[o] = s; }
      else a.dbp[0] = s;
    }
  }
}

// d rows_k[b] = [mask_b] daux * alpha_k / nsel  (the masked mean's gradient), for the L loss vectors of aux_reduce_fwd
__global__ __launch_bounds__(256) void aux_reduce_bwd_kernel(const unsigned char* __restrict__ mask, const float* __restrict__ nsel,
                                                             const float* __restrict__ daux, AuxReduce a, float* __restrict__ drows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.B * a.L) return;
  const int k = i / a.B, b = i - k * a.B;
  drows[i] = mask[b] ? daux[0] * a.alpha[k] / nsel[0] : 0.f;
}

}  // namespace

extern "C" int wsmg_update_heads_fwd(const float* x, const float* wm, const float* bm, const float* wp, const float* bp,
                                     const float* progress, int B, int K, int A, float* pred, float* prog, float* prog_rows,
                                     wsmg_stream_t stream) {
  if (!x || !wm || !bm || !wp || !bp || !pred || !prog || B <= 0 || K <= 0 || (K & 3) || A <= 0 || A > MAXA) return WSMG_EINVAL;
  if (prog_rows && !progress) return WSMG_EINVAL;
  HeadsFwd a{x, wm, bm, wp, bp, progress, pred, prog, prog_rows, B, K, A};
  hipLaunchKernelGGL(update_heads_fwd_kernel, dim3((unsigned)wsmg_cdiv(B, 4)), dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_aux_reduce_fwd(const float* const* rows, const float* alpha, int L, const unsigned char* mask, int B, float* out2,
                                   wsmg_stream_t stream) {
  if (!rows || !alpha || !mask || !out2 || L <= 0 || L > MAXL || B <= 0) return WSMG_EINVAL;
  AuxReduce a{};
  for (int k = 0; k < L; ++k) {
    if (!rows[k]) return WSMG_EINVAL;
    a.rows[k] = rows[k];
    a.alpha[k] = alpha[k];
  }
  a.mask = mask; a.out = out2; a.B = B; a.L = L;
  hipLaunchKernelGGL(aux_reduce_fwd_kernel, dim3(1), dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_aux_reduce_bwd(const float* alpha, int L, const unsigned char* mask, const float* nsel, const float* daux, int B,
                                   float* drows, wsmg_stream_t stream) {
  if (!alpha || !mask || !nsel || !daux || !drows || L <= 0 || L > MAXL || B <= 0) return WSMG_EINVAL;
  AuxReduce a{};
  for (int k = 0; k < L; ++k) a.alpha[k] = alpha[k];
  a.B = B; a.L = L;
  hipLaunchKernelGGL(aux_reduce_bwd_kernel, dim3((unsigned)wsmg_cdiv((int64_t)B * L, 256)), dim3(256), 0, wsmg_s(stream), mask, nsel, daux, a,
                     drows);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_update_heads_bwd(const float* x, const float* wm, const float* wp, const float* prog, const float* progress,
                                     const float* dpred, const float* dprog, const float* dprog_rows, int B, int K, int A, float* dx,
                                     float* dwm, float* dbm, float* dwp, float* dbp, wsmg_stream_t stream) {
  if (!x || !wm || !wp || !prog || !dx || !dwm || !dbm || !dwp || !dbp || B <= 0 || K <= 0 || A <= 0 || A > MAXA) return WSMG_EINVAL;
  if (dprog_rows && !progress) return WSMG_EINVAL;
  HeadsBwd a{x, wm, wp, prog, progress, dpred, dprog, dprog_rows, dx, dwm, dbm, dwp, dbp, B, K, A};
  hipLaunchKernelGGL(update_heads_bwd_kernel, dim3((unsigned)wsmg_cdiv(K, 8)), dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_dagger_loss_fwd(const float* pred, const float* waypoint, int ld_waypoint, const float* weights, const float* aux,
                                    int T, int N, int A, float* out2, float* den, wsmg_stream_t stream) {
  if (!pred || !waypoint || !weights || !out2 || !den || T <= 0 || N <= 0 || N > 256 || A <= 0 || ld_waypoint < A) return WSMG_EINVAL;
  DaggerLoss a{pred, waypoint, weights, aux, out2, den, T, N, A, ld_waypoint};
  hipLaunchKernelGGL(dagger_loss_fwd_kernel, dim3(1), dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_dagger_loss_bwd(const float* pred, const float* waypoint, int ld_waypoint, const float* weights, const float* den,
                                    const float* dloss, int T, int N, int A, float* dpred, wsmg_stream_t stream) {
  if (!pred || !waypoint || !weights || !den || !dloss || !dpred || T <= 0 || N <= 0 || A <= 0 || ld_waypoint < A) return WSMG_EINVAL;
  hipLaunchKernelGGL(dagger_loss_bwd_kernel, dim3((unsigned)wsmg_cdiv((int64_t)T * N * A, 256)), dim3(256), 0, wsmg_s(stream), pred, waypoint,
                     weights, den, dloss, dpred, T, N, A, ld_waypoint);
  WSMG_RETURN_LAUNCH();
}
