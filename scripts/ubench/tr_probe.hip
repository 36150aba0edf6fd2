// Semantics probe for gfx950's LDS transpose read (ds_read_b64_tr_b16): prints what every lane receives for two address patterns, so the
// emulator's model of the instruction (tests/hipemu/include/hip/hip_runtime.h emu_ds_read_tr16_b64) is pinned against the hardware.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/tr_probe.hip -o scripts/ubench/tr_probe && scripts/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4_t __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(short* out, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* s = (short*)smem;
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, grp = l >> 4;
  // mode 0: lane-linear 8-byte addresses; mode 1: rows of 256 B, lane i of a group -> row i/4 (+ 8 * (grp >> 1)), columns 4 (i%4) .. +3 (+ 16 * (grp & 1))
  const int off = mode == 0 ? l * 8 : ((i >> 2) + 8 * (grp >> 1)) * 256 + (4 * (i & 3) + 16 * (grp & 1)) * 2;
  s4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)(smem + off));
  *(s4_t*)(out + l * 4) = v;
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, 512);
  int bad = 0;
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, grp = l >> 4;
      printf("mode %d lane %2d:", mode, l);
      for (int j = 0; j < 4; ++j) {
        // model: element j of lane i = the 16-bit element (i % 4) of what lane (4 j + i / 4) of the same 16-lane group addressed
        const int src = (l & ~15) + 4 * j + (i >> 2), si = src & 15, sg = src >> 4;
        const int soff = mode == 0 ? src * 8 : ((si >> 2) + 8 * (sg >> 1)) * 256 + (4 * (si & 3) + 16 * (sg & 1)) * 2;
        const int expect = soff / 2 + (i & 3);
        printf(" %4d%s", h[l * 4 + j], h[l * 4 + j] == expect ? "" : "!");
        bad += h[l * 4 + j] != expect;
      }
      printf("\n");
    }
  }
  printf("model mismatches: %d\n", bad);
  return bad != 0;
}
