// Probe (kernel experiment, not product): what do device-scope 64-bit integer atomics cost as the LAST step of a per-channel reduction?
// 1024 workgroups x 256 threads (the grid of y5_chan_reduce_kernel); every workgroup ends with 2*C values -- either stored as a partial row
// (today: a second kernel sums the rows) or added to 2*C global accumulators with atomicAdd(unsigned long long).  Times per launch, C = 64 / 128 / 256 / 512.
// build: hipcc --offload-arch=gfx950 -O2 -o atomic_sums atomic_sums.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void tail(float* partial, unsigned long long* acc, const float* src, int C2, int spin) {
  // a little streaming work in front so that workgroups do not all arrive at once (roughly what the real pass does: reads of a 26 MB tensor)
  float s = 0.f;
  for (int i = 0; i < spin; ++i) s += src[((size_t)blockIdx.x * spin + i) * 256 + threadIdx.x];
  for (int o = threadIdx.x; o < C2; o += 256) {
    const float v = s + (float)o;
    if (MODE == 0) partial[(size_t)blockIdx.x * C2 + o] = v;
    else atomicAdd(&acc[o], (unsigned long long)(long long)(v * 1048576.0f));
  }
}

int main() {
  const int NB = 1024, SPIN = 24;
  float *partial, *src;
  unsigned long long* acc;
  hipMalloc(&partial, (size_t)NB * 1024 * 4);
  hipMalloc(&acc, 1024 * 8);
  hipMalloc(&src, (size_t)NB * SPIN * 256 * 4);
  hipMemset(src, 0, (size_t)NB * SPIN * 256 * 4);
  hipMemset(acc, 0, 1024 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int C2 : {128, 256, 512, 1024}) {
    float ms[2];
    for (int mode = 0; mode < 2; ++mode) {
      for (int w = 0; w < 3; ++w) {
        if (mode == 0) hipLaunchKernelGGL(tail<0>, dim3(NB), dim3(256), 0, 0, partial, acc, src, C2, SPIN);
        else hipLaunchKernelGGL(tail<1>, dim3(NB), dim3(256), 0, 0, partial, acc, src, C2, SPIN);
      }
      hipEventRecord(e0, 0);
      for (int it = 0; it < 100; ++it) {
        if (mode == 0) hipLaunchKernelGGL(tail<0>, dim3(NB), dim3(256), 0, 0, partial, acc, src, C2, SPIN);
        else hipLaunchKernelGGL(tail<1>, dim3(NB), dim3(256), 0, 0, partial, acc, src, C2, SPIN);
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
    printf("2C = %4d: partial rows %.2f us per launch, 64-bit atomics %.2f us per launch (100 back-to-back launches)\n", C2, ms[0] * 10.f, ms[1] * 10.f);
  }
  return 0;
}
