// Batched non_max_suppression for gfx950: candidate filter (wave-aggregated compaction), per-image bitonic
// sort of 64-bit (confidence, index) keys in LDS, greedy IoU suppression with the kept boxes held in LDS.
//
// Follows utils/general.py:658-767 + torchvision.ops.nms (general.py:750); all arithmetic is IEEE fp32 without
// FMA contraction (this translation unit is built with -ffp-contract=off) so that the SELECTION ORDER is
// bit-exact with the CPU oracle:
//   conf_j = obj * cls_j ; keep obj > conf_thres and conf > conf_thres            (general.py:679,719,731)
//   box    = (cx - w/2, cy - h/2, cx + w/2, cy + h/2)                             (general.py:722)
//   order  = conf descending, ties: lower candidate index first                   (general.py:745, stable)
//   boxes += cls * max_wh unless agnostic                                         (general.py:748-749)
//   greedy: keep i unless a kept j has inter/(area_i + area_j - inter) > iou_thres (torchvision nms)
//   first max_det kept rows, in order                                             (general.py:751)
#pragma once
#include "y5_common.h"

#define Y5_NMS_SORT_LDS_KEYS 8192   // keys sorted entirely in LDS (64 KiB)
#define Y5_NMS_MAX_DET_CAP 4096

struct Y5NmsParams {
  const void* pred;
  const void* obj_hint;  // optional (bs, n) plane of pred's dtype: (approximately) pred[..., 4]; rows it excludes with a margin are never read
  int bs, n, no, nc, nm;
  float conf_thres, iou_thres, max_wh;
  int max_det, max_nms, flags;
  const int* classes;
  int nclasses;
  float* out;
  int* out_count;
  // workspace
  int* count;                 // [bs]
  int* rcount;                // [bs]    hint path: rows the objectness plane could not exclude
  int* rows;                  // [bs][n] hint path: their indices (any order)
  unsigned long long* keys;   // [bs][cap_pad]
  unsigned char* best_cls;    // [bs][n] (best-class mode)
  float* gbox;                // [bs][gcap / 64][12][64] (chunk-major, field planes of 64 candidates): bx1,by1,bx2,by2,area, x1,y1,x2,y2,conf,cls,
                              // row index (as int bits); gcap is a multiple of 64
  long long cap, cap_pad, gcap;
  // pruning of over-long candidate lists (K1c): only the max_nms best keys are ever looked at (general.py:735 `x[...argsort...][:max_nms]`)
  int* hist;                  // [bs][Y5_NMS_BINS] histogram of the keys' confidence bins
  int* thr;                   // [bs] lowest bin that is kept (-1: the list is short enough / pruning failed, sort everything), [bs..2bs): count2
  unsigned long long* keys2;  // [bs][cap2] the keys of the kept bins
  long long cap2;
  int cls_lists;              // greedy kernel: per-(class, wave) linked lists of the kept boxes live in LDS (host: they fit and NMS is per class)
};

template <typename T> __device__ __forceinline__ float y5_ldf(const T* p, long long i) { return (float)p[i]; }

// One prediction row per lane -> candidate key(s): the test of general.py:679,719-731 on the row's own values (shared by the plain filter and
// the objectness-plane path, so both produce the same candidates by construction).  EVERY lane of the wave calls it (valid = 0 for lanes
// without a row): candidates are appended by wave-aggregated compaction -- one ballot, ONE atomicAdd per wave, slots by popcount prefix --
// because 1200 same-address returning atomics per image (one per candidate lane) serialise at ~30 ns each: 40 us of a 55 us filter.
// Candidate order in `keys` is irrelevant (the keys are sorted next).
template <typename T>
__device__ __forceinline__ void y5_nms_eval_row(const Y5NmsParams& p, int b, int r, const T* row, bool valid) {
  const int lane = threadIdx.x & 63;
  const float obj = valid ? (float)row[4] : 0.f;
  const bool live = valid && obj > p.conf_thres;
  unsigned long long* keys = p.keys + (long long)b * p.cap_pad;
  auto class_ok = [&](int j) {
    if (!p.classes) return true;
    bool ok = false;
    for (int c = 0; c < p.nclasses; ++c) ok |= (p.classes[c] == j);
    return ok;
  };
  auto append = [&](bool hit, unsigned long long key) {  // wave-uniform call
    const unsigned long long m = __ballot(hit);
    if (m == 0ull) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(p.count + b, __popcll(m));
    base = __shfl(base, 0);
    if (hit) keys[base + __popcll(m & ((1ull << lane) - 1ull))] = key;
  };
  if (p.flags & 1) {  // multi_label: every (row, class) with obj*cls > thres (general.py:726-728)
    if (__ballot(live) == 0ull) return;  // wave-uniform
    // ONE reservation per wave for all classes: count the lane's hits, exclusive wave scan, one atomicAdd of the wave total, then every lane writes its
    // keys behind its prefix.  (A ballot + atomicAdd per CLASS was 31.5 k same-address returning atomics per image with val.py's thresholds --
    // ~650 ns each under contention: 20.5 of the 26.8 ms a 64-image batch took, profiles/r03/r03_nms_val_breakdown.txt.)
    int cnt = 0;
    if (live)
      for (int j = 0; j < p.nc; ++j) cnt += ((float)row[5 + j] * obj > p.conf_thres && class_ok(j)) ? 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl(incl, lane >= d ? lane - d : lane);
      if (lane >= d) incl += up;
    }
    const int total = __shfl(incl, 63);
    if (total == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(p.count + b, total);
    base = __shfl(base, 0);
    long long slot = (long long)base + incl - cnt;
    if (live)
      for (int j = 0; j < p.nc; ++j) {
        const float conf = (float)row[5 + j] * obj;
        if (conf > p.conf_thres && class_ok(j)) {
          const unsigned idx = (unsigned)r * (unsigned)p.nc + (unsigned)j;
          keys[slot++] = ((unsigned long long)__float_as_uint(conf) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
        }
      }
  } else {  // best class only: first maximal class (general.py:730)
    bool hit = false;
    float best = 0.f;
    int bj = 0;
    if (live) {
      best = (float)row[5] * obj;
      for (int j = 1; j < p.nc; ++j) {
        const float conf = (float)row[5 + j] * obj;
        if (conf > best) { best = conf; bj = j; }
      }
      hit = best > p.conf_thres && class_ok(bj);
      if (hit) p.best_cls[(long long)b * p.n + r] = (unsigned char)bj;
    }
    append(hit, ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)r));
  }
}

// ---- K1: filter + compaction -----------------------------------------------------------------------
// STAGED: the workgroup's blockDim.x consecutive prediction rows (one contiguous block of memory) are first copied into LDS
// with full-width coalesced loads -- a thread reading "its" 170-byte row straight from HBM touches every 64-byte sector of
// the tensor at a few bytes per request; the candidate logic below is identical in both variants.
template <typename T, bool STAGED>
__global__ void y5_nms_filter_kernel(const Y5NmsParams p) {
  const int b = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const T* row;
  if constexpr (STAGED) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tile = reinterpret_cast<T*>(smem);
    const long long r0 = (long long)blockIdx.x * blockDim.x;
    const long long rows = p.n - r0 < (long long)blockDim.x ? p.n - r0 : (long long)blockDim.x;
    const T* src = static_cast<const T*>(p.pred) + ((long long)b * p.n + r0) * p.no;
    const int total = (int)(rows * p.no);
    constexpr int V = 16 / (int)sizeof(T);
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const int nv = total / V;
      for (int i = threadIdx.x; i < nv; i += blockDim.x)
        reinterpret_cast<uint4_t*>(tile)[i] = reinterpret_cast<const uint4_t*>(src)[i];
      done = nv * V;
    }
    for (int i = done + threadIdx.x; i < total; i += blockDim.x) tile[i] = src[i];
    __syncthreads();
    row = tile + (long long)threadIdx.x * p.no;
  } else {
    row = static_cast<const T*>(p.pred) + ((long long)b * p.n + (r < p.n ? r : 0)) * p.no;
  }
  y5_nms_eval_row<T>(p, b, r, row, r < p.n);
}

// ---- K1 through the objectness plane (y5_nms_batched_hint) ----------------------------------------------------------------------------
// K1a: one thread per row reads 2-4 bytes of the plane written beside z by the Detect decode (y5_detect_decode_hint /
// y5_detect_head_fwd_hint) instead of the row's 85 values.  The plane is a HINT: a row is dropped only if its plane value is below the
// threshold by more than 2^-8 relative (four fp16 ulps: the plane may have been rounded through another instruction sequence than z);
// the others are appended to the image's row list by wave-aggregated compaction (ballot, one atomicAdd per wave, prefix by popcount).
template <typename T>
__global__ void y5_nms_hint_scan_kernel(const Y5NmsParams p) {
  const int b = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  bool pass = false;
  if (r < p.n) {
    const float hv = (float)static_cast<const T*>(p.obj_hint)[(long long)b * p.n + r];
    pass = hv * 1.00390625f > p.conf_thres;  // NaN: false, like a NaN objectness in general.py:679
  }
  const unsigned long long m = __ballot(pass);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(p.rcount + b, __popcll(m));
  base = __shfl(base, 0);
  if (pass) p.rows[(long long)b * p.n + base + __popcll(m & ((1ull << lane) - 1ull))] = r;
}

// K1b: the listed rows are copied into LDS by whole waves (a row is 2-3 consecutive cache lines: lane l fetches elements l, l + 64, ...; all
// loads of a workgroup's 256 rows are independent and in flight together), then one thread per row runs the plain filter's row test
// (y5_nms_eval_row) on the LDS copy -- the list is dense, so every lane works.  Reading the row from global memory inside the test instead
// (85 dependent-latency loads per lane) measured 48 us for 77 k rows, a wave per row with a shuffle reduction 525 us; this form is bound by
// one memory round trip per 256 rows.
template <typename T>
__global__ __launch_bounds__(256) void y5_nms_hint_rows_kernel(const Y5NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);  // [256][no]
  const int b = blockIdx.y;
  const int nrows = p.rcount[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T* __restrict__ pred = static_cast<const T*>(p.pred) + (long long)b * p.n * p.no;
  for (int base = blockIdx.x * 256; base < nrows; base += gridDim.x * 256) {
    // wave w owns rows [64 w, 64 w + 64) of this chunk of the list; lane l keeps row index l in a register (the copy loop below must not
    // read it back from LDS: with the index in LDS every iteration waited for the previous iteration's LDS stores, 64 serial memory round
    // trips per wave)
    const int mine = base + wave * 64 + lane;
    const bool valid = mine < nrows;
    const int my_r = valid ? p.rows[(long long)b * p.n + mine] : 0;
    T* wt = tile + (size_t)wave * 64 * p.no;
    constexpr int RB = 32;  // rows whose loads are in flight together (then their LDS stores): two memory round trips per 64 rows
    for (int k0 = 0; k0 < 64; k0 += RB) {
      T v0[RB], v1[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int rk = __shfl(my_r, k0 + u);
        const T* src = pred + (long long)rk * p.no;
        v0[u] = lane < p.no ? src[lane] : (T)0;
        v1[u] = lane + 64 < p.no ? src[lane + 64] : (T)0;
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        if (lane < p.no) wt[(k0 + u) * p.no + lane] = v0[u];
        if (lane + 64 < p.no) wt[(k0 + u) * p.no + lane + 64] = v1[u];
      }
      for (int o = lane + 128; o < p.no; o += 64)  // rows wider than 128 values (Segment: 117 fits; multi-hundred-class heads)
#pragma unroll
        for (int u = 0; u < RB; ++u) wt[(k0 + u) * p.no + o] = pred[(long long)__shfl(my_r, k0 + u) * p.no + o];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // a wave evaluates only rows it copied itself: no workgroup barrier
    y5_nms_eval_row<T>(p, b, my_r, wt + (size_t)lane * p.no, valid);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- K1c: prune an over-long candidate list to the keys that can be among the max_nms best ------------------------------------------------
// With val.py's thresholds (conf 0.001, multi_label) an image can produce n * nc = 2 M candidates, of which only the max_nms = 30 000 most
// confident survive general.py:735.  Sorting them all was a one-workgroup bitonic network over global memory: 1.5 ms per image.  Instead:
// histogram of the confidence's top 16 float bits (128 bins per octave, LDS-private per workgroup), the lowest bin T whose tail holds >= max_nms
// keys, and a wave-aggregated compaction of the keys in bins >= T -- exactly a superset of the max_nms best, normally a few hundred keys more --
// which the sort below then orders.  A list that does not fit the compaction buffer (a million ties) keeps the old path.
#define Y5_NMS_BINS 2048
#define Y5_NMS_BIN0 0x3800   // top 16 bits of 2^-15: everything below shares bin 0
__device__ __forceinline__ int y5_nms_bin(unsigned long long key) {
  const int b = (int)(key >> 48) - Y5_NMS_BIN0;
  return b < 0 ? 0 : (b > Y5_NMS_BINS - 1 ? Y5_NMS_BINS - 1 : b);
}
__global__ __launch_bounds__(256) void y5_nms_hist_kernel(const Y5NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* h = reinterpret_cast<int*>(smem);   // [Y5_NMS_BINS]
  const int b = blockIdx.y;
  long long n = p.count[b];
  if (n > p.cap) n = p.cap;
  if (n <= p.max_nms) return;
  for (int i = threadIdx.x; i < Y5_NMS_BINS; i += 256) h[i] = 0;
  __syncthreads();
  const unsigned long long* keys = p.keys + (long long)b * p.cap_pad;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) atomicAdd(&h[y5_nms_bin(keys[i])], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < Y5_NMS_BINS; i += 256)
    if (h[i]) atomicAdd(&p.hist[(long long)b * Y5_NMS_BINS + i], h[i]);
}
__global__ __launch_bounds__(256) void y5_nms_select_kernel(const Y5NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* part = reinterpret_cast<int*>(smem);   // [256]
  const int b = blockIdx.x, tid = threadIdx.x;
  long long n = p.count[b];
  if (n > p.cap) n = p.cap;
  if (tid == 0) { p.thr[b] = -1; p.thr[p.bs + b] = 0; }
  if (n <= p.max_nms) return;
  const int* h = p.hist + (long long)b * Y5_NMS_BINS;
  constexpr int PER = Y5_NMS_BINS / 256;
  int s = 0;
  for (int q = 0; q < PER; ++q) s += h[tid * PER + q];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int acc = 0, t = 255;
    for (; t >= 0; --t) {               // the 8-bin group in which the tail reaches max_nms, then the bin inside it
      if (acc + part[t] >= p.max_nms) break;
      acc += part[t];
    }
    int T = 0;
    if (t >= 0) {
      T = t * PER + PER - 1;
      for (; T > t * PER; --T) {
        if (acc + h[T] >= p.max_nms) break;
        acc += h[T];
      }
    }
    p.thr[b] = T;
  }
}
__global__ __launch_bounds__(256) void y5_nms_compact_kernel(const Y5NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_cnt = reinterpret_cast<int*>(smem);   // [0] takes of this workgroup, [1] its base in keys2, [2] running offset
  const int b = blockIdx.y;
  const int T = p.thr[b];
  if (T < 0) return;
  long long n = p.count[b];
  if (n > p.cap) n = p.cap;
  const unsigned long long* keys = p.keys + (long long)b * p.cap_pad;
  unsigned long long* dst = p.keys2 + (long long)b * p.cap2;
  const int lane = threadIdx.x & 63;
  // a contiguous slice per workgroup, ONE global reservation for it (a wave-aggregated atomicAdd per 64 keys was 17 k same-address atomics per image)
  const long long per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const long long i_begin = per * blockIdx.x, i_end = i_begin + per < n ? i_begin + per : n;
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int mine = 0;
  for (long long i = i_begin + threadIdx.x; i < i_end; i += 256) mine += y5_nms_bin(keys[i]) >= T ? 1 : 0;
  if (mine) atomicAdd(&s_cnt[0], mine);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt[0] > 0) s_cnt[1] = atomicAdd(&p.thr[p.bs + b], s_cnt[0]);
  __syncthreads();
  if (s_cnt[0] == 0) return;
  const long long base = s_cnt[1];
  for (long long i0 = i_begin; i0 < i_end; i0 += 256) {                      // (whole waves iterate together: the ballot needs every lane)
    const long long i = i0 + threadIdx.x;
    const unsigned long long k = i < i_end ? keys[i] : 0ull;
    const bool take = i < i_end && y5_nms_bin(k) >= T;
    const unsigned long long m = __ballot(take);
    if (m == 0ull) continue;
    int off = 0;
    if (lane == 0) off = atomicAdd(&s_cnt[2], __popcll(m));                  // LDS counter: order inside the slice is irrelevant (the keys are sorted next)
    off = __shfl(off, 0);
    const long long slot = base + off + __popcll(m & ((1ull << lane) - 1ull));
    if (take && slot < p.cap2) dst[slot] = k;
  }
}
// the pruned list replaces the image's candidate list (keys, count) -- unless it overflowed the buffer, in which case everything is sorted as before
__global__ __launch_bounds__(256) void y5_nms_adopt_kernel(const Y5NmsParams p) {
  const int b = blockIdx.y;
  if (p.thr[b] < 0) return;
  const long long n2 = p.thr[p.bs + b];
  if (n2 > p.cap2) return;
  unsigned long long* keys = p.keys + (long long)b * p.cap_pad;
  const unsigned long long* src = p.keys2 + (long long)b * p.cap2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) keys[i] = src[i];
}
__global__ void y5_nms_adopt_count_kernel(const Y5NmsParams p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= p.bs || p.thr[b] < 0) return;
  const int n2 = p.thr[p.bs + b];
  if (n2 <= p.cap2) p.count[b] = n2;
}

// ---- K2: per-image bitonic sort, descending ----------------------------------------------------------
__device__ __forceinline__ void y5_cmpswap_desc(unsigned long long& a, unsigned long long& b2, bool desc) {
  if ((a < b2) == desc) { const unsigned long long t = a; a = b2; b2 = t; }
}

// Large lists (more keys than one LDS window): the merge sizes k <= WIN of the bitonic network only touch aligned windows of WIN keys, so they run as
// (windows x images) independent workgroups, each sorting its window in LDS in the direction the network wants there; the one-workgroup-per-image
// kernel below then starts at k = 2 WIN -- for the 32 768 keys a pruned val.py list has, 2 of the 15 merge sizes are left to the serial kernel.
__global__ __launch_bounds__(1024) void y5_nms_sort_windows_kernel(const Y5NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* sk = reinterpret_cast<unsigned long long*>(smem);
  const int b = blockIdx.y;
  const int tid = threadIdx.x, nt = blockDim.x;
  unsigned long long* keys = p.keys + (long long)b * p.cap_pad;
  long long n = p.count[b];
  if (n > p.cap) n = p.cap;
  long long np2 = 64;
  while (np2 < n) np2 <<= 1;
  constexpr int WIN = Y5_NMS_SORT_LDS_KEYS;
  const long long w0 = (long long)blockIdx.x * WIN;
  if (np2 <= WIN || w0 >= np2) return;
  for (int i = tid; i < WIN; i += nt) sk[i] = w0 + i < n ? keys[w0 + i] : 0ull;
  __syncthreads();
  for (int k = 2; k <= WIN; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < WIN / 2; t += nt) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const bool desc = ((w0 + lo) & k) == 0;
        const unsigned long long a = sk[lo], c = sk[hi];
        if ((a < c) == desc) { sk[lo] = c; sk[hi] = a; }
      }
      const int next_j = j > 1 ? j >> 1 : k;
      if (j > 64 || next_j > 64) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  for (int i = tid; i < WIN; i += nt) keys[w0 + i] = sk[i];   // (padding zeros included: the serial kernel continues on np2 keys)
}

__global__ void y5_nms_sort_kernel(const Y5NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* sk = reinterpret_cast<unsigned long long*>(smem);
  const int b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  unsigned long long* keys = p.keys + (long long)b * p.cap_pad;
  long long n = p.count[b];
  if (n > p.cap) n = p.cap;
  if (n <= 1) return;
  long long np2 = 64;
  while (np2 < n) np2 <<= 1;
  if (np2 <= Y5_NMS_SORT_LDS_KEYS) {
    const int N = (int)np2;
    for (int i = tid; i < N; i += nt) sk[i] = i < n ? keys[i] : 0ull;
    __syncthreads();
    // Compare distances j <= 64 stay inside the 128 consecutive keys a wave's 64 pairs cover (pair t -> lo = 2 (t & ~(j-1)) | (t & (j-1)): for the 64
    // consecutive t of a wave and j <= 64 that is keys [128 w, 128 w + 128), also for the later rounds t += blockDim of a long list), so those steps
    // need a wave barrier only; a workgroup barrier separates them from the steps with j >= 128, whose pairs cross waves: 10 instead of 66 of the
    // 16-wave barriers for 2048 keys.
    for (int k = 2; k <= N; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < N / 2; t += nt) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const bool desc = (lo & k) == 0;
          unsigned long long a = sk[lo], c = sk[hi];
          if ((a < c) == desc) { sk[lo] = c; sk[hi] = a; }
        }
        const int next_j = j > 1 ? j >> 1 : k;  // distance of the step that follows (k: first step of the next merge size)
        if (j > 64 || next_j > 64) {
          __syncthreads();
        } else {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
    for (int i = tid; i < n; i += nt) keys[i] = sk[i];
  } else {
    // large candidate sets (val-style thresholds): bitonic network over global memory; strides that fit a
    // Y5_NMS_SORT_LDS_KEYS window are finished inside LDS
    // (the windows are sorted and zero-padded already: y5_nms_sort_windows_kernel)
    const long long half = np2 >> 1;
    constexpr int WIN = Y5_NMS_SORT_LDS_KEYS;
    for (long long k = 2 * WIN; k <= np2; k <<= 1) {
      long long j = k >> 1;
      for (; j >= WIN; j >>= 1) {
        for (long long t = tid; t < half; t += nt) {
          const long long lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const long long hi = lo | j;
          const bool desc = (lo & k) == 0;
          const unsigned long long a = keys[lo], c = keys[hi];
          if ((a < c) == desc) { keys[lo] = c; keys[hi] = a; }
        }
        __syncthreads();
      }
      // remaining strides j < WIN: independent inside aligned windows of WIN keys
      for (long long w0 = 0; w0 < np2; w0 += WIN) {
        for (int i = tid; i < WIN; i += nt) sk[i] = keys[w0 + i];
        __syncthreads();
        for (long long jj = j; jj > 0; jj >>= 1) {
          for (int t = tid; t < WIN / 2; t += nt) {
            const int lo = (int)(((t & ~(jj - 1)) << 1) | (t & (jj - 1)));
            const int hi = lo | (int)jj;
            const bool desc = ((w0 + lo) & k) == 0;
            const unsigned long long a = sk[lo], c = sk[hi];
            if ((a < c) == desc) { sk[lo] = c; sk[hi] = a; }
          }
          __syncthreads();
        }
        for (int i = tid; i < WIN; i += nt) keys[w0 + i] = sk[i];
        __syncthreads();
      }
    }
  }
}

// ---- K2b: gather -- sorted keys -> compact per-candidate records (fully parallel over the GPU) ---------------
// Takes the dependent key -> row -> box load chain off the serial greedy walk: record ci of image b =
// {class-offset box, area, output box, conf, cls, row}, 12 floats, stored as field planes per chunk of 64 candidates -- one chunk is
// 3 KiB of contiguous memory that K3 pulls into its LDS ring with three LDS-DMA instructions.
#define Y5_NMS_REC 12
template <typename T>
__global__ void y5_nms_gather_kernel(const Y5NmsParams p) {
  const int b = blockIdx.y;
  const long long ci = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = p.count[b];
  if (n > p.cap) n = p.cap;
  if (n > p.max_nms) n = p.max_nms;
  if (ci >= n) return;
  const unsigned long long key = p.keys[(long long)b * p.cap_pad + ci];
  const float conf = __uint_as_float((unsigned)(key >> 32));
  const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
  long long rowi;
  int cls;
  if (p.flags & 1) { rowi = idx / (unsigned)p.nc; cls = (int)(idx - (unsigned)rowi * (unsigned)p.nc); }
  else { rowi = idx; cls = p.best_cls[(long long)b * p.n + rowi]; }
  const T* row = static_cast<const T*>(p.pred) + ((long long)b * p.n + rowi) * p.no;
  const float cx = (float)row[0], cy = (float)row[1], w = (float)row[2], h = (float)row[3];
  const float hw = w / 2.0f, hh = h / 2.0f;
  const float x1 = cx - hw, y1 = cy - hh, x2 = cx + hw, y2 = cy + hh;   // general.py:722 xywh2xyxy
  const float clsf = (float)cls;
  const float c = clsf * ((p.flags & 2) ? 0.0f : p.max_wh);             // general.py:748
  const float bx1 = x1 + c, by1 = y1 + c, bx2 = x2 + c, by2 = y2 + c;
  float* r = p.gbox + ((long long)b * p.gcap + (ci & ~63LL)) * Y5_NMS_REC + (ci & 63);  // chunk base + lane; field f at r[f * 64]
  r[0] = bx1; r[64] = by1; r[128] = bx2; r[192] = by2; r[256] = (bx2 - bx1) * (by2 - by1);
  r[320] = x1; r[384] = y1; r[448] = x2; r[512] = y2; r[576] = conf; r[640] = clsf; r[704] = __uint_as_float((unsigned)rowi);
}

// ---- K3: greedy suppression, kept boxes in LDS -------------------------------------------------------
// One workgroup of 16 waves per image walks the sorted candidates in chunks of 64.  The chunk records come through a ring of
// Y5_NMS_RING LDS slots filled by LDS-DMA: wave 1 requests chunk c + RING - 1 at the top of iteration c and waits with a COUNTED vmcnt
// until chunk c + 1 has landed -- RING - 2 chunks stay in flight, so the walk never waits for a memory round trip (one dependent
// load per chunk cost 2-4 us x 24 chunks of the 112 us this kernel took with a one-deep register prefetch):
//   (A) every wave tests the 64 candidates (lane = candidate) against a 1/16 slice of the kept list -> suppression ballots (per class: the
//       slice is the lane's own (class, wave) list of kept boxes, the only ones that can suppress it);
//   (M) every wave also forms 4 rows of the chunk's 64x64 "i suppresses j > i" bit matrix and leaves them as per-lane nibbles (columns);
//   (B) wave 0 resolves the chunk by iterating K <- alive & ~{j : col_j & K != 0} to its fixed point (the greedy answer), then appends the
//       survivors to the kept list and writes their output rows in parallel.
// Same decisions as torchvision.ops.nms: candidate i is kept iff no EARLIER KEPT candidate has IoU > thr with it.
#define Y5_NMS_GREEDY_WAVES 16
#define Y5_NMS_RING 8
#ifndef Y5_NMS_ABL   // kernel-experiment builds only: 1 = no kept-list test, 2 = no intra-chunk matrix, 4 = no serial resolution (keeps nothing)
#define Y5_NMS_ABL 0
#endif
template <typename T>
__global__ __launch_bounds__(1024)
void y5_nms_greedy_kernel(const Y5NmsParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KS = 6;                                                           // floats per kept box
  float* kept = reinterpret_cast<float*>(smem);                                   // [max_det][6]: x1,y1,x2,y2 (class offset), area, class
  float* cand = kept + (((size_t)p.max_det * KS + 3) & ~(size_t)3);              // [RING][12][64] (record fields x candidates), 16-byte aligned
  unsigned char* colpart = reinterpret_cast<unsigned char*>(cand + Y5_NMS_RING * Y5_NMS_REC * 64);  // [64 lanes][16 waves]: matrix nibbles
  unsigned long long* supmask = reinterpret_cast<unsigned long long*>(colpart + 64 * Y5_NMS_GREEDY_WAVES);  // [16]
  int* s_nkept = reinterpret_cast<int*>(supmask + Y5_NMS_GREEDY_WAVES);
  // cls_lists: kept box k hangs in list (its class, k % 16); wave w walks, per lane, the list (the lane's class, w).  A kept box only ever
  // suppresses candidates of its own class, so these are exactly the tests that can come out true -- with 80 classes a handful per lane
  // instead of the whole kept list behind a ballot each (which was 39 of the kernel's 91 us).  Any order: the tests are OR-ed.
  int* lhead = s_nkept + 4;                                // [nc][16] list heads (-1 = empty)
  int* lnext = lhead + p.nc * Y5_NMS_GREEDY_WAVES;         // [max_det]

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long n = p.count[b];
  if (n > p.cap) n = p.cap;
  if (n > p.max_nms) n = p.max_nms;
  const float* gb = p.gbox + (long long)b * p.gcap * Y5_NMS_REC;
  const T* pred = static_cast<const T*>(p.pred) + (long long)b * p.n * p.no;
  const int mi = 5 + p.nc;
  const int ow = 6 + p.nm;
  float* out = p.out + (long long)b * p.max_det * ow;

  if (tid == 0) *s_nkept = 0;
  if (p.cls_lists)
    for (int i = tid; i < p.nc * Y5_NMS_GREEDY_WAVES; i += 64 * Y5_NMS_GREEDY_WAVES) lhead[i] = -1;
  const long long last_chunk = p.gcap / 64 - 1;  // requests beyond the candidate list re-read the last chunk: the per-iteration LDS-DMA
                                                  // count stays uniform (counted vmcnt), the data is never looked at
  auto request = [&](long long chunk, int slot) {  // 3 KiB = three 1 KiB LDS-DMA instructions of wave 1
    const float* src = gb + (chunk < last_chunk ? chunk : last_chunk) * (Y5_NMS_REC * 64);
    float* dst = cand + slot * (Y5_NMS_REC * 64);
#pragma unroll
    for (int q = 0; q < 3; ++q) y5_glds16(src + q * 256 + lane * 4, dst + q * 256);
  };
  if (wave == 1) {
    for (int c = 0; c < Y5_NMS_RING - 1; ++c) request(c, c);
    y5_wait_vm<3 * (Y5_NMS_RING - 2)>();  // chunk 0 has landed
  }
  __syncthreads();
  int nkept = 0, cb = 0;
  long long chunk = 0;

  for (long long base = 0; base < n && nkept < p.max_det; base += 64, ++chunk, cb = cb + 1 == Y5_NMS_RING ? 0 : cb + 1) {
    const float* cd = cand + cb * (Y5_NMS_REC * 64);
    // the slot chunk - 1 occupied is free (its readers passed the barrier that ended the previous iteration): refill it
    if (wave == 1) request(chunk + Y5_NMS_RING - 1, cb == 0 ? Y5_NMS_RING - 1 : cb - 1);
    const bool valid = base + lane < n;
    const float bx1 = cd[lane], by1 = cd[64 + lane], bx2 = cd[128 + lane], by2 = cd[192 + lane], area = cd[256 + lane];
    // (A) against the kept list
    // Boxes of different classes never overlap (general.py:748 shifts them by cls * max_wh), so a kept box can only suppress candidates
    // of its own class: one compare + ballot per kept box decides whether the 35-instruction IoU test runs at all for this wave's 64
    // candidates.  The kernel is VALU-throughput-bound on ONE CU per image (1200 candidates x the growing kept list = 2e5 lane-tests);
    // with 80 classes most (kept box, chunk) pairs share no class.  Agnostic mode (all classes collide) tests everything.
    bool sup = false;
    const int nk_eff = (Y5_NMS_ABL & 1) ? 0 : nkept;
    const float my_cls = cd[10 * 64 + lane];
    const bool by_class = !(p.flags & 2);
    if (p.cls_lists) {
      if (valid && nk_eff > 0) {
        for (int k = lhead[(int)my_cls * Y5_NMS_GREEDY_WAVES + wave]; k >= 0 && !sup; k = lnext[k]) {
          const float kx1 = kept[k * KS + 0], ky1 = kept[k * KS + 1], kx2 = kept[k * KS + 2], ky2 = kept[k * KS + 3];
          const float ka = kept[k * KS + 4];
          const float iw = fmaxf(0.0f, fminf(kx2, bx2) - fmaxf(kx1, bx1));
          const float ih = fmaxf(0.0f, fminf(ky2, by2) - fmaxf(ky1, by1));
          const float inter = iw * ih;
          sup = (inter / (ka + area - inter)) > p.iou_thres;
        }
      }
    } else
    for (int k = wave; k < nk_eff; k += Y5_NMS_GREEDY_WAVES) {
      const float kc = kept[k * KS + 5];
      if (by_class && __ballot(valid && my_cls == kc) == 0ull) continue;
      const float kx1 = kept[k * KS + 0], ky1 = kept[k * KS + 1], kx2 = kept[k * KS + 2], ky2 = kept[k * KS + 3];
      const float ka = kept[k * KS + 4];
      const float iw = fmaxf(0.0f, fminf(kx2, bx2) - fmaxf(kx1, bx1));
      const float ih = fmaxf(0.0f, fminf(ky2, by2) - fmaxf(ky1, by1));
      const float inter = iw * ih;
      sup |= (inter / (ka + area - inter)) > p.iou_thres;
    }
    const unsigned long long m = __ballot(sup);
    if (lane == 0) supmask[wave] = m;
    // (M) rows 4*wave .. 4*wave+3 of the intra-chunk matrix (row i = the later candidates that candidate i suppresses).  The resolution below
    // wants COLUMNS (lane j: the earlier candidates that suppress j), so every wave leaves, per lane, the nibble of its four rows' bits for that lane
    static_assert(64 / Y5_NMS_GREEDY_WAVES == 4, "a nibble of matrix rows per wave");
    unsigned nib = 0u;
#pragma unroll
    for (int q = 0; q < ((Y5_NMS_ABL & 2) ? 0 : 4); ++q) {
      const int i = wave * 4 + q;
      // (per class: candidate i can only suppress later candidates of its own class -- no such lane, no IoU arithmetic for this row)
      if (by_class && __ballot(valid && lane > i && my_cls == cd[10 * 64 + i]) == 0ull) continue;
      const float kx1 = cd[i], ky1 = cd[64 + i], kx2 = cd[128 + i], ky2 = cd[192 + i], ka = cd[256 + i];
      const float iw = fmaxf(0.0f, fminf(kx2, bx2) - fmaxf(kx1, bx1));
      const float ih = fmaxf(0.0f, fminf(ky2, by2) - fmaxf(ky1, by1));
      const float inter = iw * ih;
      if (lane > i && (inter / (ka + area - inter)) > p.iou_thres) nib |= 1u << q;
    }
    colpart[lane * Y5_NMS_GREEDY_WAVES + wave] = (unsigned char)nib;
    if (wave == 1) y5_wait_vm<3 * (Y5_NMS_RING - 2)>();  // chunk + 1 is in LDS for the next iteration
    __syncthreads();
    // (B) resolution on wave 0.  Greedy rule: j is kept iff it is alive and no KEPT earlier candidate suppresses it.  Iterating
    //   K <- alive & ~{ j : col_j & K != 0 }   from K = alive
    // fixes candidate j after at most (its depth in the suppression chains + 1) rounds (j depends on lower indices only), so the first repeated K is the
    // greedy answer -- typically 3-5 rounds of a dozen instructions, where the one-kept-box-at-a-time scan (readlane of the row, 17 boxes per chunk)
    // was a third of the kernel.  max_det: decisions never depend on LATER candidates, so cutting K to its lowest `room` bits equals stopping the scan there.
    if (wave == 0) {
      unsigned long long dead = 0ull;
      for (int w2 = 0; w2 < Y5_NMS_GREEDY_WAVES; ++w2) dead |= supmask[w2];
      const unsigned long long alive = __ballot(valid) & ~dead;
      const uint4_t pb = *reinterpret_cast<const uint4_t*>(colpart + lane * Y5_NMS_GREEDY_WAVES);
      unsigned long long col = 0ull;
#pragma unroll
      for (int w2 = 0; w2 < Y5_NMS_GREEDY_WAVES; ++w2) col |= (unsigned long long)((pb[w2 >> 2] >> (8 * (w2 & 3))) & 0xFu) << (4 * w2);
      unsigned long long keepm = 0ull;
      if (!(Y5_NMS_ABL & 4)) {
        keepm = alive;
        for (int it = 0; it < 66; ++it) {
          const unsigned long long kn = alive & ~__ballot((col & keepm) != 0ull);
          if (kn == keepm) break;
          keepm = kn;
        }
        const int room = p.max_det - nkept;
        while (__popcll(keepm) > room) keepm &= ~(1ull << (63 - __builtin_clzll(keepm)));
      }
      const int cnt = nkept + __popcll(keepm);
      if ((keepm >> lane) & 1ull) {
        const int slot = nkept + __popcll(keepm & ((1ull << lane) - 1ull));
        kept[slot * KS + 0] = bx1; kept[slot * KS + 1] = by1; kept[slot * KS + 2] = bx2; kept[slot * KS + 3] = by2;
        kept[slot * KS + 4] = area; kept[slot * KS + 5] = cd[10 * 64 + lane];
        if (p.cls_lists) lnext[slot] = atomicExch(&lhead[(int)cd[10 * 64 + lane] * Y5_NMS_GREEDY_WAVES + (slot & (Y5_NMS_GREEDY_WAVES - 1))], slot);
        float* o = out + (long long)slot * ow;
        o[0] = cd[5 * 64 + lane]; o[1] = cd[6 * 64 + lane]; o[2] = cd[7 * 64 + lane]; o[3] = cd[8 * 64 + lane];
        o[4] = cd[9 * 64 + lane]; o[5] = cd[10 * 64 + lane];
        if (p.nm > 0) {
          const long long rowi = (long long)__float_as_uint(cd[11 * 64 + lane]);
          const T* row = pred + rowi * p.no;
          const float obj = (float)row[4];
          for (int q = 0; q < p.nm; ++q) o[6 + q] = (float)row[mi + q] * obj;  // general.py:719 scales masks too
        }
      }
      if (lane == 0) *s_nkept = cnt;
    }
    __syncthreads();
    nkept = *s_nkept;
  }
  if (tid == 0) p.out_count[b] = nkept;
}
