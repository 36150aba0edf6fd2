// Kernel-experiment aid (GPU): what the matrix cores of THIS box sustain.  Every wave issues back-to-back independent
// v_mfma_f32_32x32x16_f16 (NACC accumulators, operands in registers, no memory traffic); reports TFLOP/s for 1, 2 and 4 waves per
// SIMD, the s_memtime tick rate against s_memrealtime (100 MHz) and the ticks one MFMA occupies a SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak scripts/ubench/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* ticks, int iters) {
  float16_t acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}

template <int NACC>
void run(int blocks_per_cu, int threads, int ncu) {
  const int iters = 20000;
  const int grid = blocks_per_cu * ncu;
  float* out; unsigned long long* ticks;
  hipMalloc(&out, sizeof(float) * grid * threads);
  hipMalloc(&ticks, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, ticks, 100);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(threads), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
    const double nm = (double)grid * (threads / 64) * iters * NACC;
    const double mf_per_simd = (double)iters * NACC * (threads / 64) * blocks_per_cu / 4.0;  // MFMAs one SIMD executed
    printf("NACC %d  %d waves/SIMD: %.3f ms  %.1f TFLOP/s | s_memtime %.3f ticks/ns, %.1f ticks per MFMA per SIMD, %.2f ns per MFMA per SIMD\n", NACC,
           blocks_per_cu * threads / 256, ms, nm * 32768.0 / ms / 1e9, (double)h[0] / (h[1] * 10.0), (double)h[0] / mf_per_simd, h[1] * 10.0 / mf_per_simd);
  }
  hipFree(out); hipFree(ticks);
}

int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  printf("CUs %d\n", ncu);
  run<4>(1, 256, ncu);
  run<8>(1, 256, ncu);
  run<8>(2, 256, ncu);
  run<4>(4, 256, ncu);
  run<8>(1, 64, ncu);   // one wave per CU: the single-wave issue rate
  return 0;
}
