// Probe (kernel experiment, not product): issue rate of the transcendental VALU instructions on gfx950 relative to v_fma_f32.
// One workgroup of 256 threads (one wave per SIMD) per CU, 8 independent chains per lane, N iterations; cycles from s_memtime (constant 100 MHz on this part?
// -> wall clock from hipEvents is what is printed: ns per instruction per wave).
// build: hipcc --offload-arch=gfx950 -O2 -o trans_rate trans_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int n) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 0.5f + 0.001f * (threadIdx.x + i);
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
      else if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.5f) * 0.5f;        // v_mul + v_exp + v_mul
      else if (OP == 2) a[i] = __builtin_amdgcn_rcpf(a[i] + 1.0f);                // v_add + v_rcp
      else if (OP == 3) { const float t = a[i]; a[i] = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.442695f)) + 0.5f; }  // SiLU: mul exp add rcp mul (+add)
      else if (OP == 4) a[i] = __builtin_amdgcn_exp2f(a[i]);                      // bare v_exp (value drifts, irrelevant)
      else if (OP == 5) a[i] = __builtin_amdgcn_rcpf(a[i]);                       // bare v_rcp
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
float run(float* out, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, n);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256), 0, 0, out, n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 256 * 4);
  const int n = 20000;
  const char* names[] = {"v_fma_f32", "mul + v_exp_f32 + mul", "add + v_rcp_f32", "SiLU (mul exp add rcp mul add)", "bare v_exp_f32", "bare v_rcp_f32"};
  float ms[6] = {run<0>(out, n), run<1>(out, n), run<2>(out, n), run<3>(out, n), run<4>(out, n), run<5>(out, n)};
  for (int i = 0; i < 6; ++i)
    printf("%-32s %8.3f ms for %d x 8 per lane (one wave per SIMD) = %.2f ns per statement per wave = %.2fx the fma statement\n", names[i], ms[i], n, ms[i] * 1e6 / (n * 8.0),
           ms[i] / ms[0]);
  return 0;
}
