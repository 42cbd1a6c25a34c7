// Rollout-size dense layers (one row per environment): the policy's nn.Linear calls of one rollout step
// (mg_map_policy.py:150-197 of the reference: rgb / depth / map projections, the attention queries, the second-state
// compression, the GRU input projections) and its heads (policy.py:34-56: progress, action mean, critic; the diagonal
// Gaussian's mode / sample and log-probability of common/distributions.py:21-29,58-71).
//
// At 1-16 rows a GEMM library call is three launches (bias copy, GEMM with beta = 1, activation) of ~5 us each plus the gap
// in front of the GEMM, and the heads are ~20 element-wise launches on two numbers per row: the rollout step is bound by its
// launch count (≈200 launches of ≈5 us), not by arithmetic.  Here a layer is ONE launch — one workgroup per output feature,
// the weight row read once with 16-byte loads and reused for up to 8 rows at a time — and the heads are one launch per step.
#include "wsmg_common.h"

namespace {

constexpr int ROWS = 8;   // rows sharing one pass over a weight row

__device__ __forceinline__ float block_sum_256(float v, float* red) {   // result valid in every thread
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// y[r][o] = act(sum_k x[r][k] * w[o][k] + b[o]), act: 0 none, 1 ReLU, 2 tanh.  pool > 1: x is [B][K][pool] and the layer reads
// its mean over the last axis (rgb_linear's AdaptiveAvgPool1d(1) + Flatten in front of the Linear).
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B, int K, int O,
                                                          int act, int pool) {
  __shared__ float red[4];
  const int o = blockIdx.x, tid = threadIdx.x;
  const float* __restrict__ wr = w + (size_t)o * K;
  for (int r0 = 0; r0 < B; r0 += ROWS) {
    const int nr = B - r0 < ROWS ? B - r0 : ROWS;
    float acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
    if (pool == 1 && (K & 3) == 0) {
      for (int k = tid * 4; k < K; k += 1024) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          if (r < nr) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)(r0 + r) * K + k);
            acc[r] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
          }
      }
    } else {
      const float inv = 1.f / (float)pool;
      for (int k = tid; k < K; k += 256) {
        const float wv = wr[k];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          if (r < nr) {
            const float* __restrict__ xp = x + ((size_t)(r0 + r) * K + k) * pool;
            float s = 0.f;
            for (int p = 0; p < pool; ++p) s += xp[p];
            acc[r] += wv * (s * inv);
          }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (r >= nr) break;      // (uniform)
      float v = block_sum_256(acc[r], red);
      if (tid == 0) {
        v += bias ? bias[o] : 0.f;
        if (act == 1) v = v > 0.f ? v : 0.f;
        if (act == 2) v = tanhf(v);
        y[(size_t)(r0 + r) * O + o] = v;
      }
    }
  }
}

struct HeadsArgs {
  const float* feat;    // [B][K]
  const float *w_prog, *b_prog;   // [K], [1]
  const float *w_mean, *b_mean;   // [A][K], [A]
  const float* logstd;            // [A]
  const float *w_crit, *b_crit;   // [K], [1]
  const float* noise;             // [B][A] standard normals, or null: the mode
  float *prog, *value, *action, *logp;   // [B][1], [B][1], [B][A], [B]
  int K, A;
};

// one workgroup per row; wave w computes the dot products of outputs w, w + 4, ... (progress, critic, the A means)
__global__ __launch_bounds__(256) void act_heads_kernel(HeadsArgs a) {
  __shared__ float out[2 + 16];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* __restrict__ f = a.feat + (size_t)b * a.K;
  for (int j = wave; j < 2 + a.A; j += 4) {
    const float* __restrict__ w = j == 0 ? a.w_prog : j == 1 ? a.w_crit : a.w_mean + (size_t)(j - 2) * a.K;
    float s = 0.f;
    for (int k = lane; k < a.K; k += 64) s += f[k] * w[k];
    s = wave_sum(s);
    if (lane == 0) out[j] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  a.prog[b] = tanhf(out[0] + a.b_prog[0]);
  a.value[b] = out[1] + a.b_crit[0];
  float lp = 0.f;
  for (int j = 0; j < a.A; ++j) {
    const float mean = out[2 + j] + a.b_mean[j];
    const float scale = expf(a.logstd[j]);
    // the draw of torch.distributions.Normal.sample(): noise * scale, then + loc (two roundings, no fused multiply-add)
    const float act = a.noise ? __fadd_rn(__fmul_rn(a.noise[(size_t)b * a.A + j], scale), mean) : mean;
    a.action[(size_t)b * a.A + j] = act;
    // Normal.log_prob: -((x - loc)^2) / (2 var) - log(scale) - log(sqrt(2 pi))
    const float d = __fsub_rn(act, mean);
    const float var = __fmul_rn(scale, scale);
    lp += __fsub_rn(__fsub_rn(-__fmul_rn(d, d) / __fmul_rn(2.f, var), logf(scale)), 0.91893853320467274178f);
  }
  a.logp[b] = lp;
}

// ---- a list of device-to-device copies in one launch (the inputs of a captured rollout step into its static tensors: 9-12
// copies of 4 bytes to 0.8 MB each, ≈8 us apiece as separate copy launches).  A workgroup owns 16 KB of one copy.
constexpr int COPY_MAX = 32;
constexpr int COPY_CHUNK = 16384;
struct CopyBatch {
  unsigned char* dst[COPY_MAX];
  const unsigned char* src[COPY_MAX];
  long long bytes[COPY_MAX];
  int first_block[COPY_MAX + 1];
  int count;
};
__global__ __launch_bounds__(256) void copy_multi_kernel(CopyBatch b) {
  int lo = 0, hi = b.count;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= b.first_block[mid]) lo = mid; else hi = mid;
  }
  unsigned char* __restrict__ d = b.dst[lo];
  const unsigned char* __restrict__ s = b.src[lo];
  const long long i0 = (long long)((int)blockIdx.x - b.first_block[lo]) * COPY_CHUNK;
  const long long i1 = i0 + COPY_CHUNK < b.bytes[lo] ? i0 + COPY_CHUNK : b.bytes[lo];
  typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
  if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
    const long long nv = i0 + ((i1 - i0) & ~15ll);
    for (long long i = i0 + 16 * (long long)threadIdx.x; i < nv; i += 16 * 256)
      *reinterpret_cast<u32x4c*>(d + i) = *reinterpret_cast<const u32x4c*>(s + i);
    for (long long i = nv + threadIdx.x; i < i1; i += 256) d[i] = s[i];
  } else {
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) d[i] = s[i];
  }
}

}  // namespace

extern "C" int wsmg_copy_multi(const WsmgCopyDesc* descs, int n, wsmg_stream_t stream) {
  if (n < 0 || (n > 0 && !descs)) return WSMG_EINVAL;
  for (int i = 0; i < n;) {
    CopyBatch b;
    b.count = 0;
    int blocks = 0;
    for (; i < n && b.count < COPY_MAX; ++i) {
      const WsmgCopyDesc& c = descs[i];
      if (c.bytes < 0 || (c.bytes > 0 && (!c.dst || !c.src))) return WSMG_EINVAL;
      if (c.bytes == 0) continue;
      const int k = b.count++;
      b.dst[k] = (unsigned char*)c.dst; b.src[k] = (const unsigned char*)c.src; b.bytes[k] = c.bytes;
      b.first_block[k] = blocks;
      const long long nb = (c.bytes + COPY_CHUNK - 1) / COPY_CHUNK;
      if (nb > (1ll << 30) - blocks) return WSMG_EINVAL;
      blocks += (int)nb;
    }
    if (!b.count) continue;
    b.first_block[b.count] = blocks;
    hipLaunchKernelGGL(copy_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, wsmg_s(stream), b);
  }
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_linear_rows(const float* x, const float* w, const float* bias, float* y, int B, int K, int O, int act, int pool,
                                wsmg_stream_t stream) {
  if (!x || !w || !y || B <= 0 || K <= 0 || O <= 0 || act < 0 || act > 2 || pool < 1) return WSMG_EINVAL;
  hipLaunchKernelGGL(linear_rows_kernel, dim3((unsigned)O), dim3(256), 0, wsmg_s(stream), x, w, bias, y, B, K, O, act, pool);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_act_heads(const float* feat, int B, int K, const float* w_prog, const float* b_prog, const float* w_mean,
                              const float* b_mean, const float* logstd, int A, const float* w_crit, const float* b_crit,
                              const float* noise, float* prog, float* value, float* action, float* logp, wsmg_stream_t stream) {
  if (!feat || !w_prog || !b_prog || !w_mean || !b_mean || !logstd || !w_crit || !b_crit || !prog || !value || !action || !logp ||
      B <= 0 || K <= 0 || A <= 0 || A > 16)
    return WSMG_EINVAL;
  HeadsArgs a{feat, w_prog, b_prog, w_mean, b_mean, logstd, w_crit, b_crit, noise, prog, value, action, logp, K, A};
  hipLaunchKernelGGL(act_heads_kernel, dim3((unsigned)B), dim3(256), 0, wsmg_s(stream), a);
  WSMG_RETURN_LAUNCH();
}

// ---- round 6: the bias gradients of the recurrent core's dense layers in ONE launch -------------------------------------------
// d bias = sum over rows of a [rows][cols] float32 matrix (mg_map_policy.py:118-123,126-132,147-152: the Linear / GRU biases; torch
// runs a reduction kernel + a memset per tensor: 14 launches for the seven of them in the core's leaf pass).  A workgroup = one
// 64-column block of one tensor: lane = column (coalesced 256-byte rows), the four waves take rows w, w + 4, ..., float32 sums in
// row order per wave, waves combined in wave order — bit-reproducible.
namespace {
struct ColsumBatch {
  WsmgColsumDesc d[16];
  int first_block[17];    // prefix sum of ceil(cols / 64)
  int n;
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumBatch b) {
  __shared__ float sh[4][64];
  int t = 0;
  while (t + 1 < b.n && (int)blockIdx.x >= b.first_block[t + 1]) ++t;
  const WsmgColsumDesc d = b.d[t];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = ((int)blockIdx.x - b.first_block[t]) * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (c < d.cols) {
    int r = wave;
    for (; r + 4 < d.rows; r += 8) { s0 += d.x[(size_t)r * d.cols + c]; s1 += d.x[(size_t)(r + 4) * d.cols + c]; }
    if (r < d.rows) s0 += d.x[(size_t)r * d.cols + c];
  }
  sh[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && c < d.cols) d.out[c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}
}  // namespace

extern "C" int wsmg_colsum_multi(const WsmgColsumDesc* descs, int n, wsmg_stream_t stream) {
  if (!descs || n <= 0 || n > 16) return WSMG_EINVAL;
  ColsumBatch b;
  b.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!descs[i].x || !descs[i].out || descs[i].rows <= 0 || descs[i].cols <= 0) return WSMG_EINVAL;
    b.d[i] = descs[i];
    b.first_block[i] = blocks;
    blocks += (descs[i].cols + 63) / 64;
  }
  b.first_block[n] = blocks;
  hipLaunchKernelGGL(colsum_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, wsmg_s(stream), b);
  WSMG_RETURN_LAUNCH();
}
