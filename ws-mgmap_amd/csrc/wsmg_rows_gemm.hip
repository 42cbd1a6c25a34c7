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
// Tile = 32 rows x 32 columns per workgroup on v_mfma_f32_32x32x2f32 (bit-for-bit an fmaf chain in k order per wave), the four
// waves split K (each a contiguous quarter) and meet in LDS in a fixed order: deterministic.  Operands go global -> registers with
// 16-byte loads along K (lane (r, h) takes k0 + 4 h .. + 3 of row r: the four values feed four consecutive MFMAs, A and W with the
// same k assignment), 32 k values in flight per wave.  At M = 128 a layer is 32-192 workgroups of 0.26-1 MFLOP each: ~3-5 us.
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
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__global__ __launch_bounds__(256) void rows_gemm_f32_kernel(RowsGemmArgs a) {
  __shared__ float red[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int kq = a.K >> 2;                          // this wave's share of the reduction (host: K % 32 == 0)
  const int kbeg = wave * kq, kend = kbeg + kq;
  const int row = m0 + r < a.M ? m0 + r : a.M - 1;  // rows past M are computed on the last row and never stored
  const int s1 = a.ka[0], s2 = a.ka[0] + a.ka[1];
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = 0.f;

  auto a_ptr = [&](int k) -> const float* {          // &A[row][k], k a multiple of 4 inside one segment (host: segments % 8 == 0)
    if (k < s1) return a.a[0] + (size_t)row * a.lda[0] + k;
    if (k < s2) return a.a[1] + (size_t)row * a.lda[1] + (k - s1);
    return a.a[2] + (size_t)row * a.lda[2] + (k - s2);
  };
  constexpr int U = 4;                               // 8-deep chunks in flight
  for (int k0 = kbeg; k0 < kend; k0 += 8 * U) {
    f32x4 av[U], wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 8 * u + 4 * h;
      const bool ok = k0 + 8 * u < kend;
      av[u] = ok ? ldg4(a_ptr(k)) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (!a.nn) {
        wv[u] = ok ? ldg4(a.w + (size_t)(n0 + r) * a.ldw + k) : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[u][j] = ok ? a.w[(size_t)(k + j) * a.ldw + n0 + r] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][j], wv[u][j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int g = 0; g < 16; ++g) red[wave][g][lane] = acc[g];
  __syncthreads();
  // column segment of this tile (host: segment widths % 32 == 0, so a tile lies in one segment)
  int seg = 0, cn = n0;
  if (cn >= a.nc[0]) { cn -= a.nc[0]; seg = 1; if (cn >= a.nc[1]) { cn -= a.nc[1]; seg = 2; } }
  float* const cb = a.c[seg];
  const float* const ib = a.cin ? a.cin_seg[seg] : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + 256 * i, g = e >> 6, l = e & 63;
    const int orow = m0 + (g & 3) + 8 * (g >> 2) + 4 * (l >> 5), col = l & 31;
    float v = ((red[0][g][l] + red[1][g][l]) + red[2][g][l]) + red[3][g][l];
    if (orow >= a.M) continue;
    if (a.bias) v += a.bias[n0 + col];
    if (ib) v += ib[(size_t)orow * a.ldcin[seg] + cn + col];
    if (a.relu) v = v > 0.f ? v : 0.f;
    if (a.mask) v = a.mask[(size_t)orow * a.ldmask + n0 + col] > 0.f ? v : 0.f;
    cb[(size_t)orow * a.ldc[seg] + cn + col] = v;
  }
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
                                  int M, wsmg_stream_t stream) {
  if (!a0 || !w || !c0 || M <= 0 || ka0 <= 0 || nc0 <= 0) return WSMG_EINVAL;
  if ((ka1 > 0 && !a1) || (ka2 > 0 && !a2) || (nc1 > 0 && !c1) || (nc2 > 0 && !c2) || ka1 < 0 || ka2 < 0 || nc1 < 0 || nc2 < 0) return WSMG_EINVAL;
  if (ka2 > 0 && ka1 <= 0) return WSMG_EINVAL;
  if (nc2 > 0 && nc1 <= 0) return WSMG_EINVAL;
  const int K = ka0 + ka1 + ka2, N = nc0 + nc1 + nc2;
  if (K % 32 || ka0 % 8 || ka1 % 8 || ka2 % 8 || nc0 % 32 || nc1 % 32 || nc2 % 32) return WSMG_EINVAL;
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
  hipLaunchKernelGGL(rows_gemm_f32_kernel, dim3((unsigned)(N / 32), (unsigned)wsmg_cdiv(M, 32)), dim3(256), 0, wsmg_s(stream), g);
  WSMG_RETURN_LAUNCH();
}

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
