// TEST INFRASTRUCTURE ONLY.  A tiny single-threaded HIP *emulator* (fibers via ucontext) that lets the unmodified
// kernel sources under yolov5_amd/csrc be compiled for the HOST and executed lane by lane in the GPU-less build
// container, so indexing / tiling / barrier logic is checked against the oracle before GPU minutes are spent.
// It shadows <hip/hip_runtime.h> ONLY when tests/hipemu/build.sh compiles liby5emu.so; the product library is
// always built by hipcc against the real ROCm headers and never sees this file.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct int2 { int x, y; };

namespace emu {
struct Lane;
extern Lane* cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
dim3& lane_tid();
int lane_id();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& fn);
void barrier_block();
// wave collective: every live lane of the wave deposits `bytes` of payload; returns after all arrived.
// out[l] points at lane l's payload (nullptr if lane l did not participate).
void wave_exchange(const void* payload, int bytes, const void* out[64]);
// LDS-DMA model (Y5_EMU_ASYNC=1): a `buffer_load ... lds` lands only when a covering s_waitcnt vmcnt(N) of the issuing lane retires it (or at kernel
// end) -- the LATEST moment the hardware allows, so a wait that is one piece too loose reads poison instead of passing by luck.  Default (0): lands
// at issue, the EARLIEST moment (write-after-read hazards of a ring show up in this mode).  Tests of the counted-vmcnt kernels run both.
void dma_issue(void* dst, const void* src, int size);   // src == nullptr: zero fill
void vm_op_note();                                   // Y5_EMU_VM_OP (y5_common.h): a global store takes a slot of the same in-order queue
void dma_wait(int keep);                                // retire all but the `keep` youngest pieces of the calling lane
}  // namespace emu

#define threadIdx (emu::lane_tid())
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

// ---- runtime API subset used by csrc/*.hip ---------------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum { hipSuccess = 0, hipErrorNotSupported = 801 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : 2; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 3; return hipSuccess; }  // tiny "GPU": 3 CUs
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(1); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
constexpr unsigned hipStreamNonBlocking = 1;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [&]() { kern(__VA_ARGS__); })

// ---- device builtins ---------------------------------------------------------------------------------
#define __syncthreads() emu::barrier_block()
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_readlane(v, l) __shfl((v), (l))
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
#define __builtin_amdgcn_s_waitcnt(x) emu::dma_wait((int)(((x) & 15) | ((((x) >> 14) & 3) << 4)))
#define __builtin_amdgcn_s_barrier() emu::barrier_block()
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_sched_barrier(a) ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
#define Y5_EMU 1
inline void emu_wave_barrier() { const void* o[64]; char c = 0; emu::wave_exchange(&c, 1, o); }
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
inline float __expf(float x) { return expf(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int atomicAdd(int* p, int v) { int o = *p; *p += v; return o; }
inline int atomicExch(int* p, int v) { int o = *p; *p = v; return o; }
inline float atomicAdd(float* p, float v) { float o = *p; *p += v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

struct __amdgpu_buffer_rsrc_t { const char* base; unsigned num_records; };
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, int, int n, int) { return {(const char*)p, (unsigned)n}; }
inline void emu_bglds(__amdgpu_buffer_rsrc_t r, void* l, int size, unsigned voff) {
  char* d = (char*)l + (size_t)emu::lane_id() * size;
  emu::dma_issue(d, (unsigned long long)voff + size > r.num_records ? nullptr : r.base + voff, size);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, l, size, voff, soff, ioff, aux) emu_bglds((r), (void*)(uintptr_t)(l), (size), (unsigned)(voff))
inline void emu_glds(const void* g, void* l, int size) { emu::dma_issue((char*)l + (size_t)emu::lane_id() * size, g, size); }
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_glds((const void*)(uintptr_t)(g), (void*)(uintptr_t)(l), (size))

inline unsigned long long __ballot(int pred) {
  const void* o[64];
  unsigned char v = pred ? 1 : 0;
  emu::wave_exchange(&v, 1, o);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) if (o[l] && *(const unsigned char*)o[l]) m |= 1ull << l;
  return m;
}
template <typename T> inline T __shfl(T v, int src) {
  const void* o[64];
  emu::wave_exchange(&v, sizeof(T), o);
  T r = v;
  if (o[src & 63]) memcpy(&r, o[src & 63], sizeof(T));
  return r;
}

// v_permlane32_swap_b32 vdst, vsrc: vdst[lanes 32..63] <-> vsrc[lanes 0..31]; returns {new vdst, new vsrc} (pinned on the hardware by
// scripts/ubench/permlane_probe.hip)
typedef unsigned emu_uint2 __attribute__((ext_vector_type(2)));
inline emu_uint2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned vsrc, bool, bool) {
  struct P { unsigned d, s; } pl{vdst, vsrc};
  const void* o[64];
  emu::wave_exchange(&pl, sizeof(P), o);
  const int l = emu::lane_id();
  emu_uint2 r;
  if (l < 32) { r[0] = vdst; r[1] = o[l + 32] ? ((const P*)o[l + 32])->d : vsrc; }
  else { r[0] = o[l - 32] ? ((const P*)o[l - 32])->s : vdst; r[1] = vsrc; }
  return r;
}

// ds_read_b64_tr_b16 (gfx950 LDS transpose read): within a 16-lane group, lane i receives element i % 4 of the 8 bytes addressed by lanes
// 4 j + i / 4, j = 0..3 (pinned on the hardware by scripts/ubench/tr_probe.hip, profiles/r02/r02_ubench_tr_probe.log)
typedef short emu_s4 __attribute__((__vector_size__(4 * sizeof(short))));
inline emu_s4 emu_ds_read_tr16_b64(const void* addr) {
  struct P { const void* a; } pl{addr};
  const void* o[64];
  emu::wave_exchange(&pl, sizeof(P), o);
  const int l = emu::lane_id(), i = l & 15;
  emu_s4 r = {0, 0, 0, 0};
  for (int j = 0; j < 4; ++j) {
    const int src = (l & ~15) + 4 * j + (i >> 2);
    if (o[src]) {
      short v[4];
      memcpy(v, ((const P*)o[src])->a, 8);
      r[j] = v[i & 3];
    }
  }
  return r;
}

#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(l) emu_ds_read_tr16_b64((const void*)(uintptr_t)(l))

typedef _Float16 emu_half8 __attribute__((ext_vector_type(8)));
typedef float emu_float16 __attribute__((ext_vector_type(16)));
// D = A(32x16) * B(16x32) + C ; lane l holds A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31];
// D: col j = l&31, row i = (r&3) + 8*(r>>2) + 4*(l>>5)   (cdna_hip_programming.md section 3)
inline emu_float16 emu_mfma_32x32x16_f16(emu_half8 a, emu_half8 b, emu_float16 c) {
  struct P { emu_half8 a, b; } pl{a, b};
  const void* o[64];
  emu::wave_exchange(&pl, sizeof(P), o);
  const int l = emu::lane_id();
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
    float s = c[r];
    for (int k = 0; k < 16; ++k) {
      const P* pa = (const P*)o[i + 32 * (k >> 3)];
      const P* pb = (const P*)o[j + 32 * (k >> 3)];
      s += (float)pa->a[k & 7] * (float)pb->b[k & 7];
    }
    c[r] = s;
  }
  return c;
}
inline emu_float16 emu_mfma_32x32x2_f32(float a, float b, emu_float16 c) {
  struct P { float a, b; } pl{a, b};
  const void* o[64];
  emu::wave_exchange(&pl, sizeof(P), o);
  const int l = emu::lane_id();
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
    float s = c[r];
    for (int k = 0; k < 2; ++k) s = fmaf(((const P*)o[i + 32 * k])->a, ((const P*)o[j + 32 * k])->b, s);
    c[r] = s;
  }
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu_mfma_32x32x16_f16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_32x32x2_f32((a), (b), (c))
