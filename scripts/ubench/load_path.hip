// Kernel-experiment aid (GPU): L2-resident operand streaming into LDS, per CU --
//   (a) LDS-DMA: buffer_load_dwordx4 ... lds (16 B/lane, no registers),
//   (b) register path: global/buffer_load_dwordx4 -> VGPR -> ds_write_b128,
//   (c) loads only (to registers, no LDS write) as the ceiling of the vector-memory path.
// Every workgroup (256 or 512 threads, one per CU) sweeps the same 2 MiB (L2 / Infinity-Cache resident) region; prints bytes/clk/CU and TB/s.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o load_path scripts/ubench/load_path.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

template <int MODE, int UNROLL, int NT>
__global__ __launch_bounds__(NT) void load_kernel(const char* g, unsigned* out, unsigned long long* ticks, int iters, unsigned span) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, (int)span, 0x00020000);
  unsigned off = (blockIdx.x * 4096 + threadIdx.x * 16) % span;
  uint4_t accv = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (wave * UNROLL + u) * 1024), 16, (int)off, 0, 0, 0);
        off += NT * 16; if (off >= span) off -= span;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70 | (UNROLL < 15 ? UNROLL : 15));  // keep UNROLL in flight
    } else {
      uint4_t v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        off += NT * 16; if (off >= span) off -= span;
      }
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) *reinterpret_cast<uint4_t*>(smem + ((wave * UNROLL + u) * 64 + lane) * 16) = v[u];
      } else {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) accv ^= v[u];
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = accv[0] ^ accv[1] ^ accv[2] ^ accv[3];
  s ^= reinterpret_cast<unsigned*>(smem)[threadIdx.x];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE, int UNROLL, int NT>
void run(const char* name, const char* g, unsigned* out, unsigned long long* ticks, unsigned span) {
  const int iters = 2000, grid = 256;
  auto k = load_kernel<MODE, UNROLL, NT>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), 128 * 1024, 0, g, out, ticks, 50, span);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), 128 * 1024, 0, g, out, ticks, iters, span);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  const double bytes_cu = (double)iters * UNROLL * NT * 16;
  printf("%-40s waves %d unroll %2d span %5u KiB: %6.1f B/clk/CU  %6.2f TB/s  (%.0f clk per 1 KiB wave-instruction per CU)\n", name, NT / 64, UNROLL, span >> 10, bytes_cu / (double)h, bytes_cu * grid / ms / 1e9,
         (double)h / (iters * UNROLL * (NT / 64)));
}

int main() {
  char* g; unsigned* out; unsigned long long* ticks;
  const unsigned big = 64u << 20;
  hipMalloc(&g, big); hipMemset(g, 1, big); hipMalloc(&out, 4 * 256 * 512); hipMalloc(&ticks, 16);
  for (unsigned span : {2u << 20, 64u << 20}) {
    run<0, 4, 256>("LDS-DMA", g, out, ticks, span);
    run<0, 8, 256>("LDS-DMA", g, out, ticks, span);
    run<0, 8, 512>("LDS-DMA", g, out, ticks, span);
    run<0, 4, 1024>("LDS-DMA", g, out, ticks, span);
    run<1, 4, 256>("load -> VGPR -> ds_write_b128", g, out, ticks, span);
    run<1, 8, 256>("load -> VGPR -> ds_write_b128", g, out, ticks, span);
    run<1, 8, 512>("load -> VGPR -> ds_write_b128", g, out, ticks, span);
    run<1, 4, 1024>("load -> VGPR -> ds_write_b128", g, out, ticks, span);
    run<2, 8, 256>("load -> VGPR only", g, out, ticks, span);
    run<2, 8, 512>("load -> VGPR only", g, out, ticks, span);
    run<2, 8, 1024>("load -> VGPR only", g, out, ticks, span);
  }
  return 0;
}
