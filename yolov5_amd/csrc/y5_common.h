// Shared device-side helpers for the yolov5_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <type_traits>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

#define Y5_WAVE 64

// address-space casts for the LDS-DMA builtin (global -> LDS, 16 B per lane, lane-linear destination)
#define Y5_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define Y5_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ void y5_glds16(const void* gsrc, void* lds_wave_base) {
  // LDS destination = wave-uniform base + lane*16 (hardware rule); global source is per lane.
  __builtin_amdgcn_global_load_lds(Y5_GLB_PTR(gsrc), Y5_LDS_PTR(lds_wave_base), 16, 0, 0);
}

// Buffer-addressed LDS-DMA (buffer_load_dwordx4 ... offen lds): 32-bit byte offsets against a scalar resource
// descriptor; an offset beyond num_records is NOT fetched and lands as zeros -- padding taps, tile tails and the
// "dummy" loads that keep vmcnt bookkeeping uniform cost no address arithmetic and no memory traffic.
typedef __amdgpu_buffer_rsrc_t y5_rsrc_t;
#define Y5_OOB 0xFFFFFFFFu
__device__ __forceinline__ y5_rsrc_t y5_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void y5_bglds16(y5_rsrc_t r, unsigned voff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, Y5_LDS_PTR(lds_wave_base), 16, (int)voff, 0, 0, 0);
}

// A load that moves nothing (every lane out of range -> zeros into a 1 KB dummy region): waves that have no piece of a stage issue it so that the
// counted vmcnt waits see the same number of loads in every wave.  The compiler barrier behind it is REQUIRED: two identical dummies in a row are
// otherwise merged into one (dead-store elimination on the intrinsic -- measured on conv_headk.h's ring tail, profiles/r05/r05_dummy_dma_merge.log),
// which silently makes every counted wait behind them too lenient.
__device__ __forceinline__ void y5_bglds16_dummy(y5_rsrc_t r, void* dummy_lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, Y5_LDS_PTR(dummy_lds), 16, (int)Y5_OOB, 0, 0, 0);
#if !defined(Y5_EMU) && !defined(Y5_DUMMY_MERGEABLE)   // (Y5_DUMMY_MERGEABLE: the A/B build of scripts/build_dummy_ab.sh only)
  asm volatile("" ::: "memory");
#endif
}

// 16-byte buffer store / load with an explicit cache policy (aux: 16 = sc1, write-through / bypass of the non-coherent levels): the
// transport of data that another workgroup -- possibly on another XCD -- reads within the same launch (stream-K slabs)
__device__ __forceinline__ void y5_buffer_store16(uint4_t v, y5_rsrc_t r, int voff, int aux) {
#ifdef Y5_EMU
  if ((unsigned long long)(unsigned)voff + 16 <= r.num_records) memcpy(const_cast<char*>(r.base) + (unsigned)voff, &v, 16);   // (out of range: dropped, as the hardware does)
#else
  if (aux == 16) __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 16);
  else if (aux == 2) __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 2);   // nt: streaming data (kernel experiments)
  else __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
#endif
}
__device__ __forceinline__ uint4_t y5_buffer_load16(y5_rsrc_t r, int voff, int aux) {
#ifdef Y5_EMU
  uint4_t v = {0u, 0u, 0u, 0u};
  if ((unsigned long long)(unsigned)voff + 16 <= r.num_records) memcpy(&v, r.base + (unsigned)voff, 16);   // (out of range: zeros)
  return v;
#else
  return aux == 16 ? __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
#endif
}

__device__ __forceinline__ float y5_sigmoid(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float y5_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// two fp32 -> one packed fp16 pair (round to nearest even, as the scalar conversions of the LDS-transposed epilogues)
__device__ __forceinline__ uint32_t y5_pack_h2(float a, float b) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  h2_t v;
  v[0] = (half_t)a;
  v[1] = (half_t)b;
  return __builtin_bit_cast(uint32_t, v);
}

// MFMA 32x32 D layout -> pixel-major 16-byte vectors WITHOUT an LDS round trip.  After `mfma(W, A)` lane (pixel = lane & 31, g = lane >> 5) owns the
// sixteen values of channels 8 q + 4 g + e (q, e = 0..3) of its pixel.  Packed to fp16 pairs h[q][0..1] and exchanged with v_permlane32_swap
// (a[lanes 32..63] <-> b[lanes 0..31], pinned on the hardware by scripts/ubench/permlane_probe.hip), lane (pixel, g) ends up with the EIGHT
// consecutive channels 16 ks + 8 g .. + 7 for ks = 0, 1: a 16-byte store per lane (two lanes = 32 contiguous bytes of the pixel), and at the same
// time exactly the B-operand fragment of a following MFMA over these channels (conv_front.h).
__device__ __forceinline__ void y5_swap_to_pixel_vectors(const uint32_t (&h)[4][2], uint4_t (&out)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const auto r0 = __builtin_amdgcn_permlane32_swap(h[2 * ks][0], h[2 * ks + 1][0], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(h[2 * ks][1], h[2 * ks + 1][1], false, false);
    out[ks][0] = r0[0]; out[ks][1] = r1[0]; out[ks][2] = r0[1]; out[ks][3] = r1[1];
  }
}

// XCD-aware bijective remap of a 1-D block id: blocks that land on the same XCD (bid % 8) get a
// contiguous range of logical tile ids so that neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int y5_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

constexpr int y5_waitcnt_vm(int n) {  // s_waitcnt immediate: vmcnt(n), expcnt/lgkmcnt untouched (gfx9 encoding)
  return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8);
}
template <int N> __device__ __forceinline__ void y5_wait_vm() {
  __builtin_amdgcn_s_waitcnt(y5_waitcnt_vm(N < 63 ? N : 63));
  asm volatile("" ::: "memory");
}

// drain the vector-memory queue with an instruction the compiler can neither drop nor merge (the guide's split-K recipe: hipcc's
// scoreboard removes a builtin wait it believes redundant, e.g. behind a fence)
#ifdef Y5_EMU
#define Y5_DRAIN_VM() emu::dma_wait(0)
#else
#define Y5_DRAIN_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// Kernels whose counted waits also count their global STORES and register loads (conv_front.h, conv_bneck.h, conv_pw.h, conv_k3.h, conv_stem.h: every
// vector-memory operation retires in issue order on ONE counter) mark each such instruction for the host emulator's worst-case landing model
// (tests/hipemu, Y5_EMU_ASYNC=1), which can intercept the LDS-DMA builtin but not a plain load or store.  `wave_issues` restates when the WAVE issues the instruction (some lane's predicate holds): that, not the lane's own
// predicate, is what the hardware counts.  Compiles to nothing on the GPU.
#ifdef Y5_EMU
#define Y5_EMU_VM_OP(wave_issues) do { if (wave_issues) emu::vm_op_note(); } while (0)
#else
#define Y5_EMU_VM_OP(wave_issues) ((void)0)
#endif

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void y5_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    y5_static_for<B + 1, E>(f);
  }
}

// ---- BatchNorm batch statistics in the epilogue of the convolution that produces z (models/common.py:82-88 train mode: act(bn(conv(x)))) ----------------
// The separate statistics pass (bn_kernels.h y5_chan_reduce_kernel<0>) re-reads every z the convolution has just written; the streaming kernels of the
// high-resolution levels (conv_pw.h, conv_k3.h) are HBM-bound with idle vector pipes, and after their LDS-transposed epilogue every lane holds eight
// consecutive channels of one pixel -- the SAME eight channels for every store pass and every tile it ever processes.  So a lane keeps sum z and sum z^2
// of its channel octet in registers over the whole launch (16 FMAs per 16-byte store, on the fp16-ROUNDED values: what the statistics pass would read),
// and at kernel end the lanes sharing an octet, then the waves, are added in a fixed order: one row [2][C] of partials per workgroup -- deterministic, and
// exactly the input format of y5_bn_finish_kernel.
struct Y5StatAcc {
  float s0[8], s1[8];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  }
  __device__ __forceinline__ void add(uint4_t raw) {
    const half8_t h = __builtin_bit_cast(half8_t, raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float z = (float)h[e]; s0[e] += z; s1[e] += z * z; }
  }
};
// SPR lanes (lane % SPR = octet index) x 64 / SPR rows; G groups of SPR octets (channels g * SPR * 8 + ...).  wave_slot: >= 2 * NCH floats of LDS owned by
// this wave (NCH = G * SPR * 8), slot_stride: bytes between the slots of consecutive waves.  Must be reached by EVERY wave of the workgroup.
template <int SPR, int G, int NWV>
__device__ __forceinline__ void y5_stat_flush(Y5StatAcc (&acc)[G], int lane, int tid, char* wave_slot, const char* slot0, int slot_stride, float* partial_row, int C) {
  constexpr int NCH = G * SPR * 8;
  float* mine = reinterpret_cast<float*>(wave_slot);
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[g].s0[e], b = acc[g].s1[e];
#pragma unroll
      for (int m = SPR; m < 64; m <<= 1) { a += __shfl(a, lane ^ m); b += __shfl(b, lane ^ m); }   // (xor butterfly: every lane ends with the same sum)
      if (lane < SPR) { mine[(g * SPR + lane) * 8 + e] = a; mine[NCH + (g * SPR + lane) * 8 + e] = b; }
    }
  }
  __syncthreads();
  for (int o = tid; o < 2 * NCH; o += NWV * 64) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) s += reinterpret_cast<const float*>(slot0 + w * slot_stride)[o];
    const int which = o / NCH, c = o - which * NCH;
    if (c < C) partial_row[which * C + c] = s;
  }
}
