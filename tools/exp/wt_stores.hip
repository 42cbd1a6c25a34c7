// experiment (round 5): does the ~6 us gap after every large kernel (L2 write-back at the kernel boundary, 8 non-coherent XCD L2s)
// go away when the kernel's output stores do not leave dirty lines in L2?  20 dependent streaming launches, four store flavours.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    f32x4 v = in[i];
    v[0] += 1.0f;
    if (MODE == 0) out[i] = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, out + i);
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(out + i), "v"(v) : "memory");
    else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(out + i), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(out + i), "v"(v) : "memory");
  }
}
template <int MODE>
float run(f32x4* a, f32x4* b, size_t n, int reps, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(copy_kernel<MODE>, dim3(grid), dim3(256), 0, 0, a, b, n);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(copy_kernel<MODE>, dim3(grid), dim3(256), 0, 0, a, b, n);
    f32x4* t = a; a = b; b = t;
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
int main() {
  for (size_t mb : {1, 4, 16, 64, 256}) {
    size_t n = mb * (1 << 20) / 16;
    f32x4 *a, *b;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
    int grid = (int)((n + 255) / 256); if (grid > 8192) grid = 8192;
    printf("%4zu MB per launch, 40 dependent launches, us per launch: plain %.2f  nt %.2f  sc1 %.2f  sc0sc1 %.2f  sc0sc1nt %.2f   (2 x %zu MB at 8 TB/s = %.1f us)\n", mb,
           run<0>(a, b, n, 40, grid), run<1>(a, b, n, 40, grid), run<2>(a, b, n, 40, grid), run<3>(a, b, n, 40, grid), run<4>(a, b, n, 40, grid), mb, 2.0 * mb * 1.048576 / 8.0);
    hipFree(a); hipFree(b);
  }
  return 0;
}
