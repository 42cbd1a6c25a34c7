// Probe: what does ds_read_b64_tr_b16 deliver to each lane?  tile[row][col] = row*100+col (int16), 64-col rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short t[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += 64) t[i] = (short)((i / 64) * 100 + (i % 64));
  __syncthreads();
  int lane = threadIdx.x;
  int g = lane / 16, i = lane % 16, q = i / 4, p = i % 4;
  // group g: rows 4g..4g+3 (block row q), cols 4p..4p+3 of a 16-col block starting at col 16*g
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(&t[(4 * g + q) * 64 + 16 * g + 4 * p]));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
