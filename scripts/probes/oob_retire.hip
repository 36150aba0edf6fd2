// Probe (kernel experiment, not product).  Two questions about "dummy" LDS-DMA loads (all lanes out of range, used to keep counted vmcnt waits uniform):
// (1) does the compiler keep D identical dummies as D instructions?  (BARRIER = false: no -- they are merged into one, so `s_waitcnt vmcnt(D)` stops
//     waiting for anything; BARRIER = true: a compiler barrier behind each keeps them);
// (2) with D real dummy instructions: does an out-of-range LDS-DMA (buffer_load ... lds with an offset beyond num_records) retire -- i.e. decrement
// vmcnt -- ahead of OLDER in-range LDS-DMA loads that miss in every cache?  Each workgroup (one wave) issues R real 16-byte-per-lane loads from a cold
// 1 GB buffer, then D out-of-range ones into a dummy region, waits for vmcnt <= D, and checks whether the real data is in LDS.
// build: hipcc --offload-arch=gfx950 -O2 -o oob_retire oob_retire.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int rsrc_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int D, bool OOB, bool MIXED, bool BARRIER>
__global__ void probe(const unsigned* __restrict__ src, unsigned bytes, unsigned stride, unsigned* stale, unsigned* hot, unsigned hot_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2 + 64 * 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 * 2; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)bytes, 0x00020000);
  auto hs = __builtin_amdgcn_make_buffer_rsrc(hot, 0, (int)hot_bytes, 0x00020000);
  const unsigned off = (blockIdx.x * 64u + lane) * stride;   // every lane its own cache line of a cold buffer
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, (int)off, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + 256), 16, (int)(off + 64), 0, 0, 0);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (OOB) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + 512), 16, (int)0x80000000u, 0, 0, 0);
    else if (MIXED) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds + 512), 16, (lane & 1) ? (int)0x80000000u : (int)(lane * 16), 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(hs, LDS_PTR(lds + 512), 16, (int)(lane * 16), 0, 0, 0);   // in range, L2 / TCP hit
    if (BARRIER) asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
  unsigned bad = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) bad += lds[lane * 4 + (k & 3) + (k >> 2) * 256] == 0xdeadbeefu;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (bad) atomicAdd(stale, bad);
}

template <int D, bool OOB, bool MIXED, bool BARRIER>
void run(const char* name, unsigned* src, size_t bytes, unsigned* stale, unsigned* hot) {
  unsigned total = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(stale, 0, 4);
    // evict: touch another big buffer region by memset of the second half, then read the first half cold
    hipMemset(reinterpret_cast<char*>(src) + bytes / 2, rep + 1, bytes / 2);
    hipDeviceSynchronize();
    const unsigned stride = 4096 + 256 * rep;
    hipLaunchKernelGGL((probe<D, OOB, MIXED, BARRIER>), dim3(2048), dim3(64), 0, 0, src, (unsigned)(bytes / 2), stride, stale, hot, 4096u);
    hipDeviceSynchronize();
    unsigned h = 0;
    hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost);
    total += h;
  }
  printf("%-44s %s D=%d stale words seen: %u of %u\n", name, BARRIER ? "[kept apart]" : "[mergeable] ", D, total, 4u * 2048u * 64u * 8u);
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  unsigned *src, *stale, *hot;
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&stale, 4) != hipSuccess || hipMalloc(&hot, 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(src, 1, bytes);
  hipMemset(hot, 2, 4096);
  run<1, true, false, false>("real x2 then OUT-OF-RANGE dummies", src, bytes, stale, hot);
  run<4, true, false, false>("real x2 then OUT-OF-RANGE dummies", src, bytes, stale, hot);
  run<1, true, false, true>("real x2 then OUT-OF-RANGE dummies", src, bytes, stale, hot);
  run<2, true, false, true>("real x2 then OUT-OF-RANGE dummies", src, bytes, stale, hot);
  run<4, true, false, true>("real x2 then OUT-OF-RANGE dummies", src, bytes, stale, hot);
  run<8, true, false, true>("real x2 then OUT-OF-RANGE dummies", src, bytes, stale, hot);
  run<4, false, true, true>("real x2 then half-out-of-range loads", src, bytes, stale, hot);
  run<4, false, false, true>("real x2 (miss) then in-range cache hits", src, bytes, stale, hot);
  run<8, false, false, true>("real x2 (miss) then in-range cache hits", src, bytes, stale, hot);
  return 0;
}
