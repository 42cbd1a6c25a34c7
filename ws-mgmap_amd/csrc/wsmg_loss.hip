// The contrastive-monitor auxiliary loss of the update path (policy.py:72-82 of the reference) in one launch per direction:
//
//   t  = (hi - dis) / (hi - lo)                 dis = gt_path [B][H][W]; lo, hi = its batch-global min / max (device scalars)
//   a  = area-resize(t, S x S)                  = adaptive average pooling: bin i = [floor(i H / S), ceil((i + 1) H / S))
//   tg = softmax(a / tau) over the S*S bins
//   kl[b] = mean_j tg_j (log tg_j - log att_j)   = F.kl_div(log att, tg, reduction='none').mean(-1); xlogy: 0 where tg_j == 0
//
// As torch ops this is 13 launches of 4-16 us forward and 8 backward, every one waiting for the one before (the region between
// the second GRU's forward and its backward is ≈70 such launches).  One workgroup per row: the row's H*W distances are read once
// (40 KB), bins are thread-private, the softmax statistics go through LDS.  Only att receives a gradient:
//   d att_j = - g[b] tg_j / att_j / (S*S).
#include "wsmg_common.h"

namespace {

__global__ __launch_bounds__(256) void path_kl_fwd_kernel(const float* __restrict__ dis, const float* __restrict__ lo_p,
                                                          const float* __restrict__ hi_p, const float* __restrict__ att, int H, int W,
                                                          int S, float inv_tau, float* __restrict__ target, float* __restrict__ kl) {
  __shared__ float red[4];
  __shared__ float bc;
  const int b = blockIdx.x, tid = threadIdx.x, n = S * S;
  const float lo = *lo_p, hi = *hi_p, rng = hi - lo;
  const float* __restrict__ d = dis + (size_t)b * H * W;
  float* __restrict__ tg = target + (size_t)b * n;
  // pass 1: pooled values -> target (temporarily a / tau), row maximum
  float mx = -INFINITY;
  for (int j = tid; j < n; j += 256) {
    const int by = j / S, bx = j - by * S;
    const int y0 = (by * H) / S, y1 = ((by + 1) * H + S - 1) / S;
    const int x0 = (bx * W) / S, x1 = ((bx + 1) * W + S - 1) / S;
    float s = 0.f;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) s += (hi - d[y * W + x]) / rng;
    const float z = s / (float)((y1 - y0) * (x1 - x0)) * inv_tau;
    tg[j] = z;
    mx = fmaxf(mx, z);
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  mx = bc;
  float sum = 0.f;
  for (int j = tid; j < n; j += 256) {
    const float e = expf(tg[j] - mx);
    tg[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  if (tid == 0) bc = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  const float inv = 1.f / bc;
  const float* __restrict__ a = att + (size_t)b * n;
  float acc = 0.f;
  for (int j = tid; j < n; j += 256) {
    const float t = tg[j] * inv;
    tg[j] = t;
    acc += (t > 0.f ? t * logf(t) : 0.f) - t * logf(a[j]);
  }
  acc = wave_sum(acc);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) kl[b] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

__global__ __launch_bounds__(256) void path_kl_bwd_kernel(const float* __restrict__ gkl, const float* __restrict__ target,
                                                          const float* __restrict__ att, int64_t total, int n, float* __restrict__ datt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  datt[i] = -gkl[i / n] * target[i] / att[i] / (float)n;
}

}  // namespace

extern "C" int wsmg_path_kl_fwd(const float* dis, const float* lo, const float* hi, const float* att, int B, int H, int W, int S, float tau,
                                float* target, float* kl, wsmg_stream_t s) {
  if (B <= 0 || H <= 0 || W <= 0 || S <= 0 || S > H || S > W || !(tau > 0.f)) return WSMG_EINVAL;
  hipLaunchKernelGGL(path_kl_fwd_kernel, dim3((unsigned)B), dim3(256), 0, wsmg_s(s), dis, lo, hi, att, H, W, S, 1.f / tau, target, kl);
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_path_kl_bwd(const float* gkl, const float* target, const float* att, int B, int n, float* datt, wsmg_stream_t s) {
  if (B <= 0 || n <= 0) return WSMG_EINVAL;
  const int64_t total = (int64_t)B * n;
  hipLaunchKernelGGL(path_kl_bwd_kernel, dim3((unsigned)wsmg_cdiv(total, 256)), dim3(256), 0, wsmg_s(s), gkl, target, att, total, n, datt);
  WSMG_RETURN_LAUNCH();
}
