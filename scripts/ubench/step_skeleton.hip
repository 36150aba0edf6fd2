// Kernel-experiment aid (GPU): the inner step of an LDS-fed MFMA loop in isolation -- R ds_read_b128 fragment loads, then M
// v_mfma_f32_32x32x16_f16, optionally one s_barrier per step and LDS-DMA traffic beside it.  Prints shader cycles per step for
// 4 and 8 waves per CU.  Build: hipcc --offload-arch=gfx950 -O3 -w -o step_skeleton scripts/ubench/step_skeleton.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

// TM activation fragments + TN filter fragments per k-step, 2 k-steps per step: R = 2 (TM + TN), M = 2 TM TN
// MODE 0: conflict-free rows (row = lane & 31, XOR swizzle), 1: halo pattern with row wraps (TW = 20), 2: all lanes the same address (broadcast)
template <int TM, int TN, int MODE, bool BAR, bool DMA, int NT>
__global__ __launch_bounds__(NT) void step_kernel(const char* g, float* out, unsigned long long* ticks, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gq = lane >> 5, frow = lane & 31;
  for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 127);
  __syncthreads();
  int hp0[TM];
  for (int i = 0; i < TM; ++i) {
    const int m = ((wave & 1) * TM + i) * 32 + frow;
    if (MODE == 1) { const int r = m / 20, c = m - r * 20; hp0[i] = r * 22 + c; }
    else if (MODE == 2) hp0[i] = i * 32;
    else hp0[i] = m;
  }
  int w_rd[TN];
  for (int j = 0; j < TN; ++j) w_rd[j] = 64 * 1024 + (((wave >> 1) * TN + j) * 32 + frow) * 64 + ((gq ^ ((frow >> 2) & 3)) << 4);
  float16_t acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, 1 << 20, 0x00020000);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    if (BAR) __builtin_amdgcn_s_barrier();
    if (DMA) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 100 * 1024 + wave * 2048 + (s & 1) * 1024), 16, (lane * 16 + (s & 63) * 1024), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 100 * 1024 + wave * 2048 + (s & 1) * 1024), 16, (lane * 16 + (s & 63) * 1024 + 65536), 0, 0, 0);
    }
    int hoff = (s % 9) / 3 * 22 + (s % 3);
    asm volatile("" : "+s"(hoff));
    int a0[TM];
    for (int i = 0; i < TM; ++i) { const int hp = hp0[i] + hoff; a0[i] = (hp << 6) | ((gq ^ ((hp >> 2) & 3)) << 4); }
    half8_t af[2][TM], wf[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const half8_t*>(smem + (w_rd[j] ^ (ks * 32)));
#pragma unroll
      for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const half8_t*>(smem + (a0[i] ^ (ks * 32)));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70 | 2);  // vmcnt(2)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int TM, int TN, int MODE, bool BAR, bool DMA, int NT>
__global__ __launch_bounds__(NT) void pipe_kernel(const char* g, float* out, unsigned long long* ticks, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gq = lane >> 5, frow = lane & 31;
  for (int i = threadIdx.x; i < 96 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 127);
  __syncthreads();
  int hp0[TM];
  for (int i = 0; i < TM; ++i) {
    const int m = ((wave & 1) * TM + i) * 32 + frow;
    if (MODE == 1) { const int r = m / 20, c = m - r * 20; hp0[i] = r * 22 + c; }
    else if (MODE == 2) hp0[i] = i * 32;
    else hp0[i] = m;
  }
  int w_rd[TN];
  for (int j = 0; j < TN; ++j) w_rd[j] = 64 * 1024 + (((wave >> 1) * TN + j) * 32 + frow) * 64 + ((gq ^ ((frow >> 2) & 3)) << 4);
  float16_t acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, 1 << 20, 0x00020000);
  half8_t af[2][TM], wf[2][TN];
  for (int ks = 0; ks < 2; ++ks) { for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const half8_t*>(smem + (w_rd[j] ^ (ks * 32))); for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const half8_t*>(smem + ((hp0[i] << 6) ^ (ks * 32))); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    if (BAR) __builtin_amdgcn_s_barrier();
    if (DMA) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 100 * 1024 + wave * 2048 + (s & 1) * 1024), 16, (lane * 16 + (s & 63) * 1024), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 100 * 1024 + wave * 2048 + (s & 1) * 1024), 16, (lane * 16 + (s & 63) * 1024 + 65536), 0, 0, 0);
    }
    int hoff = ((s + 1) % 9) / 3 * 22 + ((s + 1) % 3);
    asm volatile("" : "+s"(hoff));
    int a0[TM];
    for (int i = 0; i < TM; ++i) { const int hp = hp0[i] + hoff; a0[i] = (hp << 6) | ((gq ^ ((hp >> 2) & 3)) << 4); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[ks][j] = *reinterpret_cast<const half8_t*>(smem + (w_rd[j] ^ (ks * 32)));
#pragma unroll
      for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const half8_t*>(smem + (a0[i] ^ (ks * 32)));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70 | 2);  // vmcnt(2)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int TM, int TN, int MODE, bool BAR, bool DMA, int threads, bool PIPE = false>
void run(const char* name, const char* g, float* out, unsigned long long* ticks) {
  const int steps = 4000, grid = 256;
  auto k = PIPE ? pipe_kernel<TM, TN, MODE, BAR, DMA, threads> : step_kernel<TM, TN, MODE, BAR, DMA, threads>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 128 * 1024, 0, g, out, ticks, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 128 * 1024, 0, g, out, ticks, steps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  const int M = 2 * TM * TN, R = 2 * (TM + TN);
  const double per_simd = (double)M * (threads / 256);
  printf("%-44s waves %d  R %2d M %2d: %7.0f cycles/step  (MFMA floor %4.0f, LDS floor %4.0f)  %.0f TFLOP/s\n", name, threads / 64, R, M, (double)h / steps, per_simd * 32,
         (double)R * (threads / 64) * 4, (double)grid * (threads / 64) * steps * M * 32768.0 / ms / 1e9);
}

int main() {
  char* g; float* out; unsigned long long* ticks;
  hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20); hipMalloc(&out, 4 * 256 * 512); hipMalloc(&ticks, 16);
  run<5, 2, 0, false, false, 256>("5x2 conflict-free", g, out, ticks);
  run<5, 2, 1, false, false, 256>("5x2 halo pattern", g, out, ticks);
  run<5, 2, 2, false, false, 256>("5x2 broadcast", g, out, ticks);
  run<5, 2, 0, true, false, 256>("5x2 conflict-free + barrier", g, out, ticks);
  run<5, 2, 1, true, true, 256>("5x2 halo + barrier + DMA", g, out, ticks);
  run<5, 1, 0, false, false, 512>("5x1 conflict-free", g, out, ticks);
  run<5, 1, 1, true, true, 512>("5x1 halo + barrier + DMA", g, out, ticks);
  run<7, 1, 1, true, true, 512>("7x1 halo + barrier + DMA", g, out, ticks);
  run<4, 4, 0, false, false, 256>("4x4 conflict-free", g, out, ticks);
  run<4, 4, 0, true, true, 256>("4x4 conflict-free + barrier + DMA", g, out, ticks);
  run<2, 2, 0, false, false, 256>("2x2 conflict-free", g, out, ticks);
  run<2, 2, 0, false, false, 512>("2x2 conflict-free", g, out, ticks);
  run<2, 2, 0, true, true, 512>("2x2 conflict-free + barrier + DMA", g, out, ticks);
  run<4, 2, 0, true, true, 512>("4x2 conflict-free + barrier + DMA", g, out, ticks);
  run<4, 2, 1, true, true, 512>("4x2 halo + barrier + DMA", g, out, ticks);
  run<4, 1, 1, true, true, 512>("4x1 halo + barrier + DMA", g, out, ticks);
  run<2, 4, 0, true, true, 512>("2x4 conflict-free + barrier + DMA", g, out, ticks);
  printf("--- software-pipelined: next step's fragments read between the MFMA halves of the current step\n");
  run<5, 2, 1, true, true, 256, true>("5x2 halo + barrier + DMA", g, out, ticks);
  run<5, 2, 0, true, true, 256, true>("5x2 conflict-free + barrier + DMA", g, out, ticks);
  run<7, 2, 1, true, true, 256, true>("7x2 halo + barrier + DMA", g, out, ticks);
  run<4, 2, 1, true, true, 256, true>("4x2 halo + barrier + DMA", g, out, ticks);
  run<5, 1, 1, true, true, 512, true>("5x1 halo + barrier + DMA", g, out, ticks);
  run<7, 1, 1, true, true, 512, true>("7x1 halo + barrier + DMA", g, out, ticks);
  run<4, 2, 1, true, true, 512, true>("4x2 halo + barrier + DMA", g, out, ticks);
  run<4, 4, 0, true, true, 256, true>("4x4 conflict-free + barrier + DMA", g, out, ticks);
  run<2, 2, 0, true, true, 256, true>("2x2 conflict-free + barrier + DMA", g, out, ticks);
  run<2, 2, 0, true, true, 512, true>("2x2 conflict-free + barrier + DMA", g, out, ticks);
  return 0;
}
