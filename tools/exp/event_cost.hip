// experiment (round 5): what an event record / a cross-stream wait between two dependent kernels costs on the GPU's side.
// 200 back-to-back launches of a ~20 us kernel on one stream: plain; + hipEventRecord (no timing) after each; + record with timing;
// + record on this stream and hipStreamWaitEvent on a second, idle stream after each (what a policy's fork/join does).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void work(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { f32x4 v = in[i]; v[0] += 1.f; out[i] = v; }
}
int main() {
  const size_t n = (size_t)64 << 20 >> 4;   // 64 MB in, 64 MB out: ~20 us
  f32x4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 0, n * 16);
  hipStream_t s, s2; hipStreamCreate(&s); hipStreamCreate(&s2);
  const int R = 200;
  hipEvent_t e0, e1, ev[R], evt[R];
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < R; ++i) { hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); hipEventCreate(&evt[i]); }
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s);
      for (int i = 0; i < R; ++i) {
        hipLaunchKernelGGL(work, dim3(4096), dim3(256), 0, s, a, b, n);
        if (mode == 1) hipEventRecord(ev[i], s);
        if (mode == 2) hipEventRecord(evt[i], s);
        if (mode == 3) { hipEventRecord(ev[i], s); hipStreamWaitEvent(s2, ev[i], 0); }
        if (mode == 4) { hipEventRecord(ev[i], s); hipStreamWaitEvent(s2, ev[i], 0); hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s2, a, b + n - 64, (size_t)64); hipEventRecord(evt[i], s2); hipStreamWaitEvent(s, evt[i], 0); }
      }
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) printf("mode %d: %.2f us per launch  (%s)\n", mode, ms * 1e3 / R,
        mode == 0 ? "plain" : mode == 1 ? "+ event record, no timing" : mode == 2 ? "+ event record with timing" : mode == 3 ? "+ record, second stream waits" : "+ fork to a second stream (tiny kernel there) and join back");
    }
  }
  return 0;
}
