// probe: buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950 — destination addressing and out-of-range behaviour
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* g, float* out, int nvalid_bytes) {
  __shared__ __attribute__((aligned(16))) float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, nvalid_bytes, 0x00020000);
  // wave w writes to lds + w*1024 bytes; lane l fetches 16 bytes at a PERMUTED source offset ((l ^ 3) * 16); odd lanes of wave 1 out of range
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int voff = wave * 1024 + ((lane ^ 3) * 16);
  if (wave == 1 && (lane & 1)) voff = 0x7fff0000;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
int main() {
  std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *g, *o; hipMalloc(&g, 4096); hipMalloc(&o, 4096); hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, g, o, 4096);
  std::vector<float> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  printf("wave0: lds[0..7] = "); for (int i = 0; i < 8; ++i) printf("%g ", r[i]); printf(" (lane 0 wanted src floats 12..15)\n");
  printf("wave0: lds[12..19] = "); for (int i = 12; i < 20; ++i) printf("%g ", r[i]); printf("\n");
  printf("wave1: lds[256..263] = "); for (int i = 256; i < 264; ++i) printf("%g ", r[i]); printf(" (lane 0 in range: 268..271, lane 1 out of range)\n");
  printf("untouched lds[600] = %g\n", r[600]);
  return 0;
}
