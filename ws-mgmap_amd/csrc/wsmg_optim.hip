// Adam step over a LIST of parameter tensors in a few launches (the update's optimizer step: `torch.optim.Adam` constructed at
// common_trainer.py:67-69 and stepped at dagger_trainer.py:540-541 in the reference).
//
// The policy has 102 live parameter tensors, 8.1 M floats: read p, g, m, v and write p, m, v = 227 MB, 45 us at HBM speed.
// The stock multi-tensor implementation is 15 launches and 0.25 ms of GPU time per step (0.5 ms when issued back to back).
// Here a launch carries a table of up to 48 tensors in its kernel arguments; a workgroup owns 4 096 consecutive elements
// of one tensor (16-byte accesses when all four pointers are 16-byte aligned) and finds its tensor by a search over the
// table's block prefix.  Arithmetic as torch.optim.Adam (amsgrad = False, maximize = False), in its order:
//     g' = g + wd p;  m += (1 - b1) (g' - m);  v = b2 v + (1 - b2) g' g';  p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
#include "wsmg_common.h"

namespace {

constexpr int ADAM_MAX = 48;      // tensors per launch (kernel-argument table)
constexpr int ADAM_CHUNK = 4096;  // elements per workgroup

struct AdamBatch {
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  int first_block[ADAM_MAX + 1];  // prefix of workgroups per tensor
  long long n[ADAM_MAX];
  int count;
  float lr_bc1, beta1c, beta2, beta2c, sqrt_bc2, eps, wd;
  const float* step_dev;   // or null: the step count lives on the device (HIP-graph replay: the arguments are frozen at capture) and
  float lr, beta1;         //          the bias corrections are computed from it here
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamBatch& b, float lr_bc1, float sqrt_bc2) {
  if (b.wd != 0.f) g = fmaf(b.wd, p, g);
  m = m + b.beta1c * (g - m);
  v = v * b.beta2 + b.beta2c * g * g;
  const float denom = sqrtf(v) / sqrt_bc2 + b.eps;
  p = p - lr_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamBatch b) {
  float lr_bc1 = b.lr_bc1, sqrt_bc2 = b.sqrt_bc2;
  if (b.step_dev) {   // 1 - beta^step in double, as the host path does
    const double st = (double)*b.step_dev;
    lr_bc1 = (float)((double)b.lr / (1.0 - pow((double)b.beta1, st)));
    sqrt_bc2 = (float)sqrt(1.0 - pow((double)b.beta2, st));
  }
  int lo = 0, hi = b.count;          // tensor t with first_block[t] <= blockIdx.x < first_block[t + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= b.first_block[mid]) lo = mid; else hi = mid;
  }
  float* __restrict__ p = b.p[lo];
  const float* __restrict__ g = b.g[lo];
  float* __restrict__ m = b.m[lo];
  float* __restrict__ v = b.v[lo];
  const long long n = b.n[lo];
  const long long i0 = (long long)((int)blockIdx.x - b.first_block[lo]) * ADAM_CHUNK;
  const long long i1 = i0 + ADAM_CHUNK < n ? i0 + ADAM_CHUNK : n;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  if (vec) {
    const long long nv = i0 + ((i1 - i0) & ~3ll);
    for (long long i = i0 + 4 * (long long)threadIdx.x; i < nv; i += 4 * 256) {
      f32x4 pp = ld4(p + i), mm = ld4(m + i), vv = ld4(v + i);
      const f32x4 gg = ld4(g + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pj = pp[j], mj = mm[j], vj = vv[j];
        adam1(pj, gg[j], mj, vj, b, lr_bc1, sqrt_bc2);
        pp[j] = pj; mm[j] = mj; vv[j] = vj;
      }
      st4(p + i, pp); st4(m + i, mm); st4(v + i, vv);
    }
    for (long long i = nv + threadIdx.x; i < i1; i += 256) adam1(p[i], g[i], m[i], v[i], b, lr_bc1, sqrt_bc2);
  } else {
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) adam1(p[i], g[i], m[i], v[i], b, lr_bc1, sqrt_bc2);
  }
}

}  // namespace

static int adam_launch(const WsmgAdamDesc* descs, int n, float lr, float beta1, float beta2, float eps, float weight_decay,
                       double bias_correction1, double bias_correction2, const float* step_dev, wsmg_stream_t s) {
  if (n < 0 || (n > 0 && !descs)) return WSMG_EINVAL;
  if (!step_dev && (!(bias_correction1 > 0.0) || !(bias_correction2 > 0.0))) return WSMG_EINVAL;
  for (int i = 0; i < n;) {
    AdamBatch b;
    b.count = 0;
    int blocks = 0;
    for (; i < n && b.count < ADAM_MAX; ++i) {
      const WsmgAdamDesc& d = descs[i];
      if (d.n < 0 || (d.n > 0 && (!d.param || !d.grad || !d.exp_avg || !d.exp_avg_sq))) return WSMG_EINVAL;
      if (d.n == 0) continue;
      const int k = b.count++;
      b.p[k] = d.param; b.g[k] = d.grad; b.m[k] = d.exp_avg; b.v[k] = d.exp_avg_sq; b.n[k] = d.n;
      b.first_block[k] = blocks;
      const long long nb = (d.n + ADAM_CHUNK - 1) / ADAM_CHUNK;
      if (nb > (1ll << 30) - blocks) return WSMG_EINVAL;
      blocks += (int)nb;
    }
    if (!b.count) continue;
    b.first_block[b.count] = blocks;
    b.lr_bc1 = step_dev ? 0.f : (float)((double)lr / bias_correction1);
    b.beta1c = 1.f - beta1;
    b.beta2 = beta2;
    b.beta2c = 1.f - beta2;
    b.sqrt_bc2 = step_dev ? 1.f : (float)sqrt(bias_correction2);
    b.eps = eps;
    b.wd = weight_decay;
    b.step_dev = step_dev;
    b.lr = lr;
    b.beta1 = beta1;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, wsmg_s(s), b);
  }
  WSMG_RETURN_LAUNCH();
}

extern "C" int wsmg_adam_step_multi(const WsmgAdamDesc* descs, int n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                    double bias_correction1, double bias_correction2, wsmg_stream_t s) {
  return adam_launch(descs, n, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2, nullptr, s);
}

// The same step with the step COUNT read from device memory (one float32, already incremented for this step): what a captured
// HIP graph replays — its kernel arguments are frozen, so the bias corrections cannot be passed by value.
extern "C" int wsmg_adam_step_multi_dev(const WsmgAdamDesc* descs, int n, float lr, float beta1, float beta2, float eps,
                                        float weight_decay, const float* step_dev, wsmg_stream_t s) {
  if (!step_dev) return WSMG_EINVAL;
  return adam_launch(descs, n, lr, beta1, beta2, eps, weight_decay, 0.0, 0.0, step_dev, s);
}
