// Cross-modal single-query attention (instruction tokens, then map cells), forward and
// backward: MGMapNet._attn at mg_map_policy.py:173-178 and its two call sites :229-235.
//
//   logits[i] = sum_c q[c] k[i][c] - 1e8*mask[i];  attn = softmax(logits*scale);  out[c] = sum_i attn[i] v[i][c]
//
// One query per row makes both contractions GEMV-shaped (about 1 flop per byte of k / v), so
// the kernel is HBM-bound by construction: one workgroup per row streams k and v exactly once
// (token-major [I][256], one 1-KiB wave-wide float4 load per token), keeps the logits in LDS and
// reduces with wavefront shuffles.  The matrix cores are used where the path has a real
// contraction — the key projection (Conv1d k=1), which runs on the conv engine's MFMA GEMM.
#include "wsmg_common.h"

namespace {

constexpr int AC = 256;       // channels (hidden_size/2)
constexpr int AWAVES = 4;
constexpr int AMAX_I = 2560;  // logits kept in LDS (10 KiB); 2304 = the 48 x 48 encoded map of the E=196 geometry
constexpr int SAME_WAVES = 16;  // waves per row of the keys-are-values kernels (B = 512 rows alone give only 2 workgroups per CU)

__device__ inline float block_reduce_max(float v, float* sh, int wave, int lane) {
  v = wave_max(v);
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < AWAVES; ++w) r = fmaxf(r, sh[w]);
  __syncthreads();
  return r;
}
__device__ inline float block_reduce_sum(float v, float* sh, int wave, int lane) {
  v = wave_sum(v);
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < AWAVES; ++w) r += sh[w];
  __syncthreads();
  return r;
}

template <class T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, const T* __restrict__ k,
                                                       const T* __restrict__ v,
                                                       const uint8_t* __restrict__ mask, float scale, int I,
                                                       float* __restrict__ out, float* __restrict__ attn,
                                                       const int64_t* __restrict__ row_index) {
  __shared__ float lg[AMAX_I];
  __shared__ float red[AWAVES];
  __shared__ __attribute__((aligned(16))) float part[AWAVES][AC];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // row_index: keys / values / mask are shared between rows (one set per unique instruction); kr = the row's set
  const size_t kr = row_index ? (size_t)row_index[b] : (size_t)b;
  const f32x4 qv = reinterpret_cast<const f32x4*>(q + (size_t)b * AC)[lane];
  const T* kb = k + kr * I * AC + lane * 4;
  const T* vb = v + kr * I * AC + lane * 4;

  // phase 1: logits (one wave per token; 64 lanes x float4 = the 256 channels)
  // four tokens per trip: four independent 512-byte loads in flight per wave (one token per trip was bound by the
  // load -> shuffle-reduce latency chain: 1.7 TB/s on the 576-token map stage)
  for (int i0 = wave * 4; i0 < I; i0 += AWAVES * 4) {
    f32x4 kv[4];
    float d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) kv[u] = (i0 + u < I) ? ld4(kb + (size_t)(i0 + u) * AC) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = wave_sum(qv[0] * kv[u][0] + qv[1] * kv[u][1] + qv[2] * kv[u][2] + qv[3] * kv[u][3]);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u;
        if (i < I) {
          float v = d[u];
          if (mask && mask[kr * I + i]) v = v - 1e8f;
          lg[i] = v * scale;
        }
      }
    }
  }
  __syncthreads();
  // phase 2: softmax over the tokens
  float mx = -INFINITY;
  for (int i = tid; i < I; i += 256) mx = fmaxf(mx, lg[i]);
  mx = block_reduce_max(mx, red, wave, lane);
  float sm = 0.f;
  for (int i = tid; i < I; i += 256) {
    float e = expf(lg[i] - mx);
    lg[i] = e;
    sm += e;
  }
  sm = block_reduce_sum(sm, red, wave, lane);
  const float inv = 1.f / sm;
  for (int i = tid; i < I; i += 256) {
    float a = lg[i] * inv;
    lg[i] = a;
    attn[(size_t)b * I + i] = a;
  }
  __syncthreads();
  // phase 3: out = attn . v (each wave a token subset, combined through LDS)
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i0 = wave * 4; i0 < I; i0 += AWAVES * 4) {
    f32x4 vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = (i0 + u < I) ? ld4(vb + (size_t)(i0 + u) * AC) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a = (i0 + u < I) ? lg[i0 + u] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += a * vv[u][j];
    }
  }
  reinterpret_cast<f32x4*>(&part[wave][0])[lane] = acc;
  __syncthreads();
  float o = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
  out[(size_t)b * AC + tid] = o;
}

// backward: da = dout.v + dattn; s = sum attn*da; dl = attn*(da-s)*scale;
//           dq = sum_i dl[i] k[i]; dk[i] = dl[i] q; dv[i] = attn[i] dout
// dk == dv (same buffer) selects the keys-are-values form used when the key projection is folded into the
// query: one gradient tensor dk[i] = dl[i] q + attn[i] dout is written instead of two.
template <class T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ q, const T* __restrict__ k,
                                                       const T* __restrict__ v,
                                                       const float* __restrict__ attn,
                                                       const float* __restrict__ dout,
                                                       const float* __restrict__ dattn, float scale, int I,
                                                       float* __restrict__ dq, T* __restrict__ dk,
                                                       T* __restrict__ dv, const int64_t* __restrict__ row_index,
                                                       float* __restrict__ dlogits) {
  __shared__ float da[AMAX_I];
  __shared__ float red[AWAVES];
  __shared__ __attribute__((aligned(16))) float part[AWAVES][AC];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f32x4 qv = reinterpret_cast<const f32x4*>(q + (size_t)b * AC)[lane];
  const f32x4 gv = reinterpret_cast<const f32x4*>(dout + (size_t)b * AC)[lane];
  // shared-set form (row_index != NULL): k / v belong to the row's instruction; dk / dv are NOT written per row
  // (they are sums over the rows of a set: the caller forms them from dlogits, attn, q and dout with two small GEMMs)
  const size_t kr = row_index ? (size_t)row_index[b] : (size_t)b;
  const bool shared = row_index != nullptr;
  const T* kb = k + kr * I * AC + lane * 4;
  const T* vb = v + kr * I * AC + lane * 4;
  T* dkb = dk + (size_t)b * I * AC + lane * 4;
  T* dvb = dv + (size_t)b * I * AC + lane * 4;
  const float* ab = attn + (size_t)b * I;
  const bool same = (dk == dv) && !shared;

  // the incoming gradient of the attention weights goes to LDS with coalesced loads first (a per-token scalar load
  // by lane 0 inside the token loop sat on its critical path: 261 vs 149 us on the map stage)
  for (int i = tid; i < I; i += 256) da[i] = dattn ? dattn[(size_t)b * I + i] : 0.f;
  __syncthreads();
  for (int i0 = wave * 4; i0 < I; i0 += AWAVES * 4) {
    f32x4 vv[4];
    float d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = (i0 + u < I) ? ld4(vb + (size_t)(i0 + u) * AC) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = wave_sum(gv[0] * vv[u][0] + gv[1] * vv[u][1] + gv[2] * vv[u][2] + gv[3] * vv[u][3]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u;
      if (i >= I) break;
      if (!same && !shared) {
        const float a = ab[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a * gv[j];
        st4(dvb + (size_t)i * AC, o);
      }
      if (lane == 0) da[i] += d[u];
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < I; i += 256) s += ab[i] * da[i];
  s = block_reduce_sum(s, red, wave, lane);
  for (int i = tid; i < I; i += 256) {
    const float dl = ab[i] * (da[i] - s) * scale;
    da[i] = dl;
    if (dlogits) dlogits[(size_t)b * I + i] = dl;
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i0 = wave * 4; i0 < I; i0 += AWAVES * 4) {
    f32x4 kv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) kv[u] = (i0 + u < I) ? ld4(kb + (size_t)(i0 + u) * AC) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u;
      if (i >= I) break;
      const float dl = da[i];
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] += dl * kv[u][j];
        o[j] = dl * qv[j];
      }
      if (same) {
        const float a = ab[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += a * gv[j];
      }
      if (!shared) st4(dkb + (size_t)i * AC, o);
    }
  }
  reinterpret_cast<f32x4*>(&part[wave][0])[lane] = acc;
  __syncthreads();
  dq[(size_t)b * AC + tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
}


// ---------------------------------------------------------------------------------------------------
// Keys ARE values (k == v): the map stage after the key projection has been folded into the query
// (ops._AttnFolded; mg_map_policy.py:233-235 on the 576 map tokens).  The generic kernels above read the
// token tensor twice per pass (logits, then the weighted sum; gradient dots, then dq) — 295 KB per row,
// more than a CU's share of L2 — and pay a six-step shuffle reduction per token.  Here:
//   forward  = ONE pass, online softmax (running max / sum per wave, rescaled accumulator), raw logits kept
//              in LDS so the attention weights can still be written out;
//   backward = one READ pass + one WRITE pass.  With da_i = dattn_i + dout.x_i and s = sum_i a_i da_i:
//              dq = scale * (sum_i a_i da_i x_i - s * sum_i a_i x_i)   -> both sums accumulate in the read pass;
//              dx_i = a_i (da_i - s) scale * q + a_i * dout            -> needs no x at all (write-only pass).
// Eight tokens per trip and wave; their eight partial dot products are reduced together with a halving
// butterfly (v_permlane32/16_swap, DPP): 10 lane exchanges per 8 tokens instead of 48.
template <int C, int D, int NV>
__device__ __forceinline__ void halve8(float (&v)[NV], int lane) {
  if constexpr (D == 32 || D == 16) {
#pragma unroll
    for (int i = 0; i < C / 2; ++i) {
      const int lo = __float_as_int(v[i]), hi = __float_as_int(v[i + C / 2]);
      auto r = (D == 32) ? __builtin_amdgcn_permlane32_swap(lo, hi, false, false)
                         : __builtin_amdgcn_permlane16_swap(lo, hi, false, false);
      v[i] = __int_as_float(r[0]) + __int_as_float(r[1]);
    }
  } else {   // D == 8: DPP row rotate by 8
    const bool up = (lane & D) != 0;
#pragma unroll
    for (int i = 0; i < C / 2; ++i) {
      const float keep = up ? v[i + C / 2] : v[i];
      const float send = up ? v[i] : v[i + C / 2];
      const int x = __float_as_int(send);
      v[i] = keep + __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false));
    }
  }
}
// dot products of 8 token rows with one vector, all 8 results in every lane: d[u] = sum_c g[c] x_u[c]
__device__ __forceinline__ void dots8(const f32x4 (&xv)[8], const f32x4 g, int lane, float (&d)[8]) {
  float p[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) p[u] = g[0] * xv[u][0] + g[1] * xv[u][1] + g[2] * xv[u][2] + g[3] * xv[u][3];
  halve8<8, 32>(p, lane);
  halve8<4, 16>(p, lane);
  halve8<2, 8>(p, lane);
  float r = p[0];   // token ((lane>>5)&1)*4 + ((lane>>4)&1)*2 + ((lane>>3)&1), summed over lanes that differ in bits 3..5
  {
    int x = __float_as_int(r);
    r += __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    x = __float_as_int(r);
    r += __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    x = __float_as_int(r);
    r += __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));  // row_half_mirror: lane i <-> 7 - i
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) d[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 8 * u));
}

template <class T, int NW>
__global__ __launch_bounds__(NW * 64) void attn_same_fwd_kernel(const float* __restrict__ q, const T* __restrict__ x,
                                                            const uint8_t* __restrict__ mask, float scale, int I,
                                                            float* __restrict__ out, float* __restrict__ attn) {
  __shared__ float lg[AMAX_I];
  __shared__ float wm[NW], wsum[NW];
  __shared__ __attribute__((aligned(16))) float part[NW][AC];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f32x4 qv = reinterpret_cast<const f32x4*>(q + (size_t)b * AC)[lane];
  const T* xb = x + (size_t)b * I * AC + lane * 4;
  float m = -INFINITY, ssum = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // the next trip's eight token rows are requested before this trip is reduced (the loop is latency-bound otherwise:
  // 8 waves per CU, one 8-load batch in flight per wave)
  f32x4 xv[8], xn[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) xn[u] = (wave * 8 + u < I) ? ld4(xb + (size_t)(wave * 8 + u) * AC) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i0 = wave * 8; i0 < I; i0 += NW * 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = xn[u];
    const int i1 = i0 + NW * 8;
    if (i1 < I) {
#pragma unroll
      for (int u = 0; u < 8; ++u) xn[u] = (i1 + u < I) ? ld4(xb + (size_t)(i1 + u) * AC) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float d[8];
    dots8(xv, qv, lane, d);
    float mx = m;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float l = d[u];
      if (mask && i0 + u < I && mask[(size_t)b * I + i0 + u]) l = l - 1e8f;
      l = (i0 + u < I) ? l * scale : -INFINITY;
      d[u] = l;
      mx = fmaxf(mx, l);
    }
    if (lane < 8 && i0 + lane < I) {
      float mine = d[0];
#pragma unroll
      for (int u = 1; u < 8; ++u) mine = (lane == u) ? d[u] : mine;
      lg[i0 + lane] = mine;
    }
    const float corr = expf(m - mx);   // m = -inf on the first trip: exp(-inf) = 0 and acc, ssum are 0
    ssum *= corr;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] *= corr;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float e = expf(d[u] - mx);   // tokens past I: exp(-inf) = 0
      ssum += e;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += e * xv[u][j];
    }
    m = mx;
  }
  if (lane == 0) { wm[wave] = m; wsum[wave] = ssum; }
  reinterpret_cast<f32x4*>(&part[wave][0])[lane] = acc;
  __syncthreads();
  float M = wm[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) M = fmaxf(M, wm[w]);
  float S = 0.f, o = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float f = expf(wm[w] - M);   // a wave that saw no token has m = -inf, f = 0
    S += wsum[w] * f;
    o += part[w][tid & (AC - 1)] * f;
  }
  const float inv = 1.f / S;
  if (tid < AC) out[(size_t)b * AC + tid] = o * inv;
  for (int i = tid; i < I; i += NW * 64) attn[(size_t)b * I + i] = expf(lg[i] - M) * inv;
}

template <class T, int NW>
__global__ __launch_bounds__(NW * 64) void attn_same_bwd_kernel(const float* __restrict__ q, const T* __restrict__ x,
                                                            const float* __restrict__ attn,
                                                            const float* __restrict__ dout,
                                                            const float* __restrict__ dattn, float scale, int I,
                                                            float* __restrict__ dq, T* __restrict__ dx,
                                                            float* __restrict__ dlogits) {
  __shared__ float da[AMAX_I];
  __shared__ float wsum[NW];
  __shared__ __attribute__((aligned(16))) float pa[NW][AC], po[NW][AC];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f32x4 qv = reinterpret_cast<const f32x4*>(q + (size_t)b * AC)[lane];
  const f32x4 gv = reinterpret_cast<const f32x4*>(dout + (size_t)b * AC)[lane];
  const T* xb = x + (size_t)b * I * AC + lane * 4;
  T* dxb = dx + (size_t)b * I * AC + lane * 4;
  const float* ab = attn + (size_t)b * I;
  const float* dab = dattn ? dattn + (size_t)b * I : nullptr;
  // read pass
  float s = 0.f;
  f32x4 A = {0.f, 0.f, 0.f, 0.f}, O = {0.f, 0.f, 0.f, 0.f};
  f32x4 xv[8], xn[8];   // (next trip prefetched, as in the forward kernel)
#pragma unroll
  for (int u = 0; u < 8; ++u) xn[u] = (wave * 8 + u < I) ? ld4(xb + (size_t)(wave * 8 + u) * AC) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i0 = wave * 8; i0 < I; i0 += NW * 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = xn[u];
    const int i1 = i0 + NW * 8;
    if (i1 < I) {
#pragma unroll
      for (int u = 0; u < 8; ++u) xn[u] = (i1 + u < I) ? ld4(xb + (size_t)(i1 + u) * AC) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the per-token scalars of this trip: lane u < 8 fetches token i0 + u, then broadcast
    float a_l = 0.f, g_l = 0.f;
    if (lane < 8 && i0 + lane < I) { a_l = ab[i0 + lane]; g_l = dab ? dab[i0 + lane] : 0.f; }
    float d[8];
    dots8(xv, gv, lane, d);
    float mine = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l), u));
      const float dai = d[u] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_l), u));
      const float w = a * dai;   // a = 0 past I
      s += w;
#pragma unroll
      for (int j = 0; j < 4; ++j) { A[j] += w * xv[u][j]; O[j] += a * xv[u][j]; }
      mine = (lane == u) ? dai : mine;
    }
    if (lane < 8 && i0 + lane < I) da[i0 + lane] = mine;
  }
  if (lane == 0) wsum[wave] = s;
  reinterpret_cast<f32x4*>(&pa[wave][0])[lane] = A;
  reinterpret_cast<f32x4*>(&po[wave][0])[lane] = O;
  __syncthreads();
  s = 0.f;
  float sa = 0.f, so = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { s += wsum[w]; sa += pa[w][tid & (AC - 1)]; so += po[w][tid & (AC - 1)]; }
  if (tid < AC) dq[(size_t)b * AC + tid] = scale * (sa - s * so);
  // write pass: dx_i = dl_i * q + a_i * dout
  for (int i0 = wave * 8; i0 < I; i0 += NW * 8) {
    float a_l = 0.f, dl_l = 0.f;
    if (lane < 8 && i0 + lane < I) {
      a_l = ab[i0 + lane];
      dl_l = a_l * (da[i0 + lane] - s) * scale;
      if (dlogits) dlogits[(size_t)b * I + i0 + lane] = dl_l;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + u >= I) break;
      const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a_l), u));
      const float dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dl_l), u));
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = dl * qv[j] + a * gv[j];
      st4(dxb + (size_t)(i0 + u) * AC, o);
    }
  }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// float32 -> e4m3 (round to nearest even, saturating at +-448), 4 values per thread
__global__ __launch_bounds__(256) void quantize_e4m3_kernel(const float* __restrict__ x, int64_t n4, float inv_scale,
                                                            unsigned* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j] * inv_scale, -448.f), 448.f);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
  y[i] = (unsigned)w;
}
}  // namespace

extern "C" int wsmg_attn_fwd(const float* q, const float* k, const float* v, const uint8_t* mask, float scale, int B,
                             int I, int C, float* out, float* attn, wsmg_stream_t stream) {
  if (C != AC || B <= 0 || I <= 0 || I > AMAX_I) return WSMG_EINVAL;
  if (k == v) {   // keys are values: one-pass form
    hipLaunchKernelGGL((attn_same_fwd_kernel<float, SAME_WAVES>), dim3(B), dim3(SAME_WAVES * 64), 0, wsmg_s(stream), q, k, mask, scale, I, out, attn);
    WSMG_RETURN_LAUNCH();
  }
  hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(B), dim3(256), 0, wsmg_s(stream), q, k, v, mask, scale, I, out, attn, (const int64_t*)nullptr);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_attn_fwd_bf16(const float* q, const void* k, const void* v, const uint8_t* mask, float scale, int B,
                                  int I, int C, float* out, float* attn, wsmg_stream_t stream) {
  if (C != AC || B <= 0 || I <= 0 || I > AMAX_I) return WSMG_EINVAL;
  if (k == v) {
    hipLaunchKernelGGL((attn_same_fwd_kernel<bf16_t, SAME_WAVES>), dim3(B), dim3(SAME_WAVES * 64), 0, wsmg_s(stream), q, (const bf16_t*)k, mask, scale, I, out, attn);
    WSMG_RETURN_LAUNCH();
  }
  hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, wsmg_s(stream), q, (const bf16_t*)k,
                     (const bf16_t*)v, mask, scale, I, out, attn, (const int64_t*)nullptr);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_attn_bwd(const float* q, const float* k, const float* v, const float* attn, const float* dout,
                             const float* dattn, float scale, int B, int I, int C, float* dq, float* dk, float* dv,
                             wsmg_stream_t stream) {
  if (C != AC || B <= 0 || I <= 0 || I > AMAX_I) return WSMG_EINVAL;
  if (k == v && dk == dv) {   // keys are values: one read pass + one write pass
    hipLaunchKernelGGL((attn_same_bwd_kernel<float, SAME_WAVES>), dim3(B), dim3(SAME_WAVES * 64), 0, wsmg_s(stream), q, k, attn, dout, dattn, scale, I, dq, dk,
                       (float*)nullptr);
    WSMG_RETURN_LAUNCH();
  }
  hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(B), dim3(256), 0, wsmg_s(stream), q, k, v, attn, dout, dattn, scale,
                     I, dq, dk, dv, (const int64_t*)nullptr, (float*)nullptr);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_attn_bwd_bf16(const float* q, const void* k, const void* v, const float* attn, const float* dout,
                                  const float* dattn, float scale, int B, int I, int C, float* dq, void* dk, void* dv,
                                  wsmg_stream_t stream) {
  if (C != AC || B <= 0 || I <= 0 || I > AMAX_I) return WSMG_EINVAL;
  if (k == v && dk == dv) {
    hipLaunchKernelGGL((attn_same_bwd_kernel<bf16_t, SAME_WAVES>), dim3(B), dim3(SAME_WAVES * 64), 0, wsmg_s(stream), q, (const bf16_t*)k, attn, dout, dattn, scale, I,
                       dq, (bf16_t*)dk, (float*)nullptr);
    WSMG_RETURN_LAUNCH();
  }
  hipLaunchKernelGGL(attn_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, wsmg_s(stream), q, (const bf16_t*)k,
                     (const bf16_t*)v, attn, dout, dattn, scale, I, dq, (bf16_t*)dk, (bf16_t*)dv, (const int64_t*)nullptr, (float*)nullptr);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_quantize_e4m3(const float* x, int64_t n, float inv_scale, uint8_t* y, wsmg_stream_t stream) {
  if (n <= 0 || (n & 3) != 0) return WSMG_EINVAL;
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(quantize_e4m3_kernel, dim3((unsigned)wsmg_cdiv(n4, 256)), dim3(256), 0, wsmg_s(stream), x, n4, inv_scale,
                     reinterpret_cast<unsigned*>(y));
  WSMG_RETURN_LAUNCH();
}

// ---- shared-set form: B query rows attend over U << B key/value sets (one per unique instruction; row b uses set
// row_index[b]).  Replaces the reference's per-row copies of the instruction keys / values (mg_map_policy.py:229-232 on
// tensors that repeat every instruction T times) — nothing is gathered, and dK / dV of a set are formed by the caller.
extern "C" int wsmg_attn_shared_fwd(const float* q, const float* k_sets, const float* v_sets, const uint8_t* mask_sets,
                                    const int64_t* row_index, float scale, int B, int I, int C, float* out, float* attn,
                                    wsmg_stream_t stream) {
  if (C != AC || B <= 0 || I <= 0 || I > AMAX_I || !row_index) return WSMG_EINVAL;
  hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(B), dim3(256), 0, wsmg_s(stream), q, k_sets, v_sets, mask_sets, scale, I, out,
                     attn, row_index);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_attn_shared_fwd_bf16(const float* q, const void* k_sets, const void* v_sets, const uint8_t* mask_sets,
                                         const int64_t* row_index, float scale, int B, int I, int C, float* out, float* attn,
                                         wsmg_stream_t stream) {
  if (C != AC || B <= 0 || I <= 0 || I > AMAX_I || !row_index) return WSMG_EINVAL;
  hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, wsmg_s(stream), q, (const bf16_t*)k_sets,
                     (const bf16_t*)v_sets, mask_sets, scale, I, out, attn, row_index);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_attn_shared_bwd(const float* q, const float* k_sets, const float* v_sets, const float* attn,
                                    const float* dout, const float* dattn, const int64_t* row_index, float scale, int B, int I,
                                    int C, float* dq, float* dlogits, wsmg_stream_t stream) {
  if (C != AC || B <= 0 || I <= 0 || I > AMAX_I || !row_index || !dlogits) return WSMG_EINVAL;
  hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(B), dim3(256), 0, wsmg_s(stream), q, k_sets, v_sets, attn, dout, dattn, scale, I,
                     dq, (float*)nullptr, (float*)nullptr, row_index, dlogits);
  WSMG_RETURN_LAUNCH();
}
extern "C" int wsmg_attn_shared_bwd_bf16(const float* q, const void* k_sets, const void* v_sets, const float* attn,
                                         const float* dout, const float* dattn, const int64_t* row_index, float scale, int B,
                                         int I, int C, float* dq, float* dlogits, wsmg_stream_t stream) {
  if (C != AC || B <= 0 || I <= 0 || I > AMAX_I || !row_index || !dlogits) return WSMG_EINVAL;
  hipLaunchKernelGGL(attn_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, wsmg_s(stream), q, (const bf16_t*)k_sets,
                     (const bf16_t*)v_sets, attn, dout, dattn, scale, I, dq, (bf16_t*)nullptr, (bf16_t*)nullptr, row_index, dlogits);
  WSMG_RETURN_LAUNCH();
}
