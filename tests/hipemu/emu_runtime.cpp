// TEST INFRASTRUCTURE ONLY: fiber scheduler of the HIP emulator (see include/hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <vector>

alignas(256) char smem[160 * 1024];  // the kernels' `extern __shared__ char smem[]`

namespace emu {
enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Dma { void* dst; int size; unsigned char data[16]; };
struct Lane {
  ucontext_t ctx;
  dim3 tid;
  int linear, state;
  unsigned seq;
  std::vector<Dma> dma;   // pieces in flight (Y5_EMU_ASYNC=1), oldest first
};
struct Slot { unsigned char data[64][64]; unsigned tag[64]; };
static std::vector<Lane> lanes;
static std::vector<Slot> slots;  // [wave][8]
static ucontext_t sched;
static const std::function<void()>* g_fn;
static char* stacks = nullptr;
static size_t stacks_cap = 0;
Lane* cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;
static constexpr size_t STACK = 256 * 1024;

dim3& lane_tid() { return cur->tid; }
int lane_id() { return cur->linear & 63; }

static void yield(int st) {
  Lane* me = cur;
  me->state = st;
  swapcontext(&me->ctx, &sched);
}
void barrier_block() { yield(WAIT_BLOCK); }

void wave_exchange(const void* payload, int bytes, const void* out[64]) {
  Lane* me = cur;
  const int wave = me->linear >> 6, l = me->linear & 63;
  if (bytes > 64) { fprintf(stderr, "emu: payload too large\n"); abort(); }
  const unsigned seq = ++me->seq;
  Slot& s = slots[(size_t)wave * 8 + (seq & 7)];
  memcpy(s.data[l], payload, bytes);
  s.tag[l] = seq;
  yield(WAIT_WAVE);
  for (int k = 0; k < 64; ++k) out[k] = s.tag[k] == seq ? s.data[k] : nullptr;
}

static int g_async = -1;
void dma_issue(void* dst, const void* src, int size) {
  if (g_async < 0) { const char* e = getenv("Y5_EMU_ASYNC"); g_async = e && e[0] == '1'; }
  if (!g_async || size > 16) {
    if (src) memcpy(dst, src, size); else memset(dst, 0, size);
    return;
  }
  Dma d{dst, size, {}};
  if (src) memcpy(d.data, src, size);
  cur->dma.push_back(d);
}
void vm_op_note() {   // a global store of the calling lane's wave: occupies a slot of the in-order queue, moves nothing
  if (g_async < 0) { const char* e = getenv("Y5_EMU_ASYNC"); g_async = e && e[0] == '1'; }
  if (g_async) cur->dma.push_back(Dma{nullptr, 0, {}});
}
void dma_wait(int keep) {
  std::vector<Dma>& q = cur->dma;
  const int n = (int)q.size() - (keep < 0 ? 0 : keep);
  if (n <= 0) return;
  for (int i = 0; i < n; ++i)
    if (q[i].size) memcpy(q[i].dst, q[i].data, q[i].size);
  q.erase(q.begin(), q.begin() + n);
}

static void trampoline() {
  (*g_fn)();
  dma_wait(0);
  cur->state = DONE;
  swapcontext(&cur->ctx, &sched);
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& fn) {
  const int T = (int)(block.x * block.y * block.z);
  if (shmem > sizeof(smem)) { fprintf(stderr, "emu: dynamic LDS %zu exceeds 160 KiB\n", shmem); abort(); }
  if ((size_t)T * STACK > stacks_cap) {
    free(stacks);
    stacks_cap = (size_t)T * STACK;
    stacks = (char*)aligned_alloc(4096, stacks_cap);
  }
  g_fn = &fn;
  g_blockDim = block;
  g_gridDim = grid;
  const int nwaves = (T + 63) / 64;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bxr = 0; bxr < grid.x; ++bxr) {
        // blocks run one after another, HIGHEST x first: hardware promises no dispatch order, and the one inter-workgroup protocol of
        // the library (stream-K, conv_igemm.h) has every workgroup depend on higher-numbered ones only
        const unsigned bx = grid.x - 1 - bxr;
        g_blockIdx = dim3(bx, by, bz);
        memset(smem, 0xCD, shmem);  // poison: uninitialised LDS reads show up as garbage, as on hardware
        lanes.assign(T, Lane{});
        slots.assign((size_t)nwaves * 8, Slot{});
        for (int t = 0; t < T; ++t) {
          Lane& L = lanes[t];
          L.linear = t;
          L.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          L.state = RUN;
          getcontext(&L.ctx);
          L.ctx.uc_stack.ss_sp = stacks + (size_t)t * STACK;
          L.ctx.uc_stack.ss_size = STACK;
          L.ctx.uc_link = nullptr;
          makecontext(&L.ctx, trampoline, 0);
        }
        int ndone = 0;
        while (ndone < T) {
          bool ran = false;
          for (int t = 0; t < T; ++t) {
            if (lanes[t].state != RUN) continue;
            cur = &lanes[t];
            swapcontext(&sched, &lanes[t].ctx);
            ran = true;
            if (lanes[t].state == DONE) ++ndone;
          }
          bool released = false;
          for (int w = 0; w < nwaves; ++w) {  // wave collectives
            bool all = true, any = false;
            for (int t = w * 64; t < T && t < (w + 1) * 64; ++t) {
              if (lanes[t].state == DONE) continue;
              any = true;
              all &= lanes[t].state == WAIT_WAVE;
            }
            if (any && all) {
              for (int t = w * 64; t < T && t < (w + 1) * 64; ++t) if (lanes[t].state == WAIT_WAVE) lanes[t].state = RUN;
              released = true;
            }
          }
          bool all = true, any = false;
          for (int t = 0; t < T; ++t) {
            if (lanes[t].state == DONE) continue;
            any = true;
            all &= lanes[t].state == WAIT_BLOCK;
          }
          if (any && all) {
            for (int t = 0; t < T; ++t) if (lanes[t].state == WAIT_BLOCK) lanes[t].state = RUN;
            released = true;
          }
          if (!ran && !released && ndone < T) { fprintf(stderr, "emu: deadlock (divergent barrier?)\n"); abort(); }
        }
      }
  cur = nullptr;
}
}  // namespace emu
