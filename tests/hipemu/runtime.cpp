// TEST INFRASTRUCTURE (see include/hip/hip_runtime.h): the fibre scheduler behind the host build of the kernel sources.
// One OS thread.  hipemu::launch walks the grid workgroup by workgroup; inside a workgroup every thread is a fibre with
// its own stack, run round-robin; a fibre gives up the CPU only at a workgroup barrier or a wave rendezvous.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <sys/mman.h>

#include <vector>

hipemu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

extern "C" void hipemu_switch(void** save_sp, void* to_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_switch, .-hipemu_switch
)");

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#endif
#endif

#ifdef HIPEMU_TSAN
// race-detector build: every thread of a workgroup is a ThreadSanitizer fibre; switching says nothing about ordering
// (no_sync), barriers / rendezvous / launch boundaries are the only happens-before edges.  This file is compiled WITHOUT
// instrumentation (the scheduler's own variables are not the subject).
#include <sanitizer/tsan_interface.h>
#define HIPEMU_TSAN_RELEASE(a) __tsan_release((void*)(a))
#define HIPEMU_TSAN_ACQUIRE(a) __tsan_acquire((void*)(a))
#else
#define HIPEMU_TSAN_RELEASE(a) ((void)0)
#define HIPEMU_TSAN_ACQUIRE(a) ((void)0)
#endif

namespace hipemu {
namespace {
#ifdef HIPEMU_ASAN
constexpr size_t STACK = 1024 * 1024;  // instrumented frames are several times larger
#else
constexpr size_t STACK = 192 * 1024;
#endif

struct Wave {
  int alive = 0, arrived = 0;
  unsigned gen = 0;
  uint64_t mask = 0;
  alignas(64) unsigned char buf[2][64][64];
  alignas(64) unsigned char res[2][64][16];
};
struct Fibre {
  void* fake = nullptr;  // AddressSanitizer's fake-stack handle while the fibre is switched out
  void* sp = nullptr;
  bool done = false;
  int wave = 0, lane = 0, par = 0;
  int waiting = 0;  // diagnostics: 1 = at the workgroup barrier, 2 = at a wave rendezvous
  unsigned long long nsync = 0;  // ... and how many barriers + rendezvous this thread has entered
  hipemu_idx tid;
};

// every change of stack goes through here: AddressSanitizer has to be told which stack the code runs on
void* main_fake = nullptr;
const void* main_bottom = nullptr;
size_t main_size = 0;
unsigned char* stack_of(int t);
void* main_tsan = nullptr;
std::vector<void*> tsan_fibres;  // one per thread slot, kept across workgroups and launches
int tsan_target = -1;            // slot the next switch goes to (-1: the scheduler's own context)
char launch_token[2];            // happens-before between consecutive launches (a kernel boundary orders everything)
char host_token, done_token;     // host code before the launch -> every thread; every thread -> host code after it
char wg_token[2];                // workgroup k -> workgroup k + 1: they run one after another here and share the static LDS arrays
unsigned long long wg_no = 0;
unsigned long long launch_no = 0;

void switch_stacks(void** save_sp, void** save_fake, void* to_sp, const void* to_bottom, size_t to_size) {
#ifdef HIPEMU_TSAN
  __tsan_switch_to_fiber(tsan_target < 0 ? main_tsan : tsan_fibres[(size_t)tsan_target], __tsan_switch_to_fiber_no_sync);
#endif
#ifdef HIPEMU_ASAN
  __sanitizer_start_switch_fiber(save_fake, to_bottom, to_size);
#endif
  hipemu_switch(save_sp, to_sp);
#ifdef HIPEMU_ASAN
  __sanitizer_finish_switch_fiber(save_fake ? *save_fake : nullptr, nullptr, nullptr);  // back on the saved stack
#endif
}

std::vector<Fibre> fibres;
std::vector<Wave> waves;
unsigned char* stacks = nullptr;
size_t stacks_n = 0;
void* main_sp = nullptr;
int cur = -1;
int blk_alive = 0, blk_arrived = 0;
unsigned blk_gen = 0;
const std::function<void()>* body = nullptr;

int nfib = 0;
unsigned long long stalled = 0;  // yields since the last barrier release / rendezvous completion / thread exit

// hand the CPU to the next thread of the workgroup that has not finished (round-robin), directly
void yield() {
  if (++stalled > 4ull * (unsigned long long)nfib + 8) {
    fprintf(stderr, "hipemu: deadlock in workgroup (%u,%u,%u): the remaining threads wait at a barrier / wave rendezvous "
                    "the others never reach (divergent collective?)\n", blockIdx.x, blockIdx.y, blockIdx.z);
    for (size_t w = 0; w < waves.size(); ++w) {
      int at_barrier = 0, at_wave = 0, done = 0;
      for (int t = (int)w * 64; t < std::min(nfib, (int)w * 64 + 64); ++t) {
        done += fibres[t].done;
        at_barrier += !fibres[t].done && fibres[t].waiting == 1;
        at_wave += !fibres[t].done && fibres[t].waiting == 2;
      }
      fprintf(stderr, "  wave %zu: %d lanes at the workgroup barrier, %d at a wave rendezvous, %d finished\n", w, at_barrier,
              at_wave, done);
      if (at_wave && at_wave < 64) {
        fprintf(stderr, "    at the rendezvous (thread: synchronisations entered):");
        for (int t = (int)w * 64; t < std::min(nfib, (int)w * 64 + 64); ++t)
          if (!fibres[t].done && fibres[t].waiting == 2) fprintf(stderr, " %d:%llu", t, fibres[t].nsync);
        fprintf(stderr, "\n    elsewhere:");
        for (int t = (int)w * 64; t < std::min(nfib, (int)w * 64 + 64); ++t)
          if (!fibres[t].done && fibres[t].waiting != 2) fprintf(stderr, " %d:%llu", t, fibres[t].nsync);
        fprintf(stderr, "\n");
      }
    }
    abort();
  }
  int nxt = cur;
  do { nxt = nxt + 1 == nfib ? 0 : nxt + 1; } while (fibres[nxt].done);
  if (nxt == cur) return;
  const int prev = cur;
  cur = nxt;
  threadIdx = fibres[nxt].tid;
  tsan_target = nxt;
  switch_stacks(&fibres[prev].sp, &fibres[prev].fake, fibres[nxt].sp, stack_of(nxt), STACK);
}

void release_wave(Wave& w) { w.arrived = 0; ++w.gen; stalled = 0; }
void release_block() { blk_arrived = 0; ++blk_gen; stalled = 0; }

void fibre_exit() {
  Fibre& f = fibres[cur];
  Wave& w = waves[f.wave];
  f.done = true;
  stalled = 0;
  --w.alive;
  w.mask &= ~(1ull << f.lane);
  if (w.alive > 0 && w.arrived == w.alive) release_wave(w);
  --blk_alive;
  if (blk_alive > 0 && blk_arrived == blk_alive) release_block();
  HIPEMU_TSAN_RELEASE(&launch_token[launch_no & 1]);
  HIPEMU_TSAN_RELEASE(&done_token);
  HIPEMU_TSAN_RELEASE(&wg_token[wg_no & 1]);
  if (blk_alive == 0) {
    tsan_target = -1;
    switch_stacks(&f.sp, nullptr, main_sp, main_bottom, main_size);  // nullptr: this stack is not coming back
  } else {
    int nxt = cur;
    do { nxt = nxt + 1 == nfib ? 0 : nxt + 1; } while (fibres[nxt].done);
    cur = nxt;
    threadIdx = fibres[nxt].tid;
    tsan_target = nxt;
    switch_stacks(&f.sp, nullptr, fibres[nxt].sp, stack_of(nxt), STACK);
  }
  abort();  // never resumed
}
extern "C" void hipemu_entry() {
#ifdef HIPEMU_ASAN
  {
    const void* from_bottom = nullptr;
    size_t from_size = 0;
    __sanitizer_finish_switch_fiber(nullptr, &from_bottom, &from_size);  // first time on this stack
    if (!main_bottom && cur == 0) { main_bottom = from_bottom; main_size = from_size; }
  }
#endif
  HIPEMU_TSAN_ACQUIRE(&launch_token[(launch_no + 1) & 1]);  // everything the previous launch did
  HIPEMU_TSAN_ACQUIRE(&host_token);
  HIPEMU_TSAN_ACQUIRE(&wg_token[(wg_no + 1) & 1]);
  (*body)();
  fibre_exit();
}
}  // namespace

namespace {
unsigned char* stack_of(int t) { return stacks + (size_t)t * STACK; }
}  // namespace
int lane() { return fibres[cur].lane; }
int wave_alive() { return waves[fibres[cur].wave].alive; }
uint64_t alive_mask() { return waves[fibres[cur].wave].mask; }

void sync_threads() {
  const unsigned g = blk_gen;
  HIPEMU_TSAN_RELEASE(&blk_gen);
  fibres[cur].waiting = 1;
  ++fibres[cur].nsync;
  if (++blk_arrived == blk_alive) release_block();
  else while (blk_gen == g) yield();
  fibres[cur].waiting = 0;
  HIPEMU_TSAN_ACQUIRE(&blk_gen);
}
void wave_sync() {
  Wave& w = waves[fibres[cur].wave];
  const unsigned g = w.gen;
  HIPEMU_TSAN_RELEASE(&w.gen);
  fibres[cur].waiting = 2;
  ++fibres[cur].nsync;
  if (++w.arrived == w.alive) release_wave(w);
  else while (w.gen == g) yield();
  fibres[cur].waiting = 0;
  HIPEMU_TSAN_ACQUIRE(&w.gen);
}
const unsigned char (*post(const void* payload, int n))[64] {
  Fibre& f = fibres[cur];
  Wave& w = waves[f.wave];
  const int p = f.par;
  f.par ^= 1;
  memcpy(w.buf[p][f.lane], payload, (size_t)n);
  wave_sync();
  return w.buf[p];
}

// rendezvous with ONE evaluation: the lane that completes the wave runs `fn(slots, results)` for all 64 lanes
const unsigned char (*post_once(const void* payload, int n, void (*fn)(const unsigned char (*)[64], unsigned char (*)[16])))[16] {
  Fibre& f = fibres[cur];
  Wave& w = waves[f.wave];
  const int p = f.par;
  f.par ^= 1;
  memcpy(w.buf[p][f.lane], payload, (size_t)n);
  const unsigned g = w.gen;
  ++f.nsync;
  HIPEMU_TSAN_RELEASE(&w.gen);
  if (++w.arrived == w.alive) {
    HIPEMU_TSAN_ACQUIRE(&w.gen);  // the other lanes' operands
    fn(w.buf[p], w.res[p]);
    HIPEMU_TSAN_RELEASE(&w.gen);  // ... and the results
    release_wave(w);
  } else {
    fibres[cur].waiting = 2;
    while (w.gen == g) yield();
    fibres[cur].waiting = 0;
  }
  HIPEMU_TSAN_ACQUIRE(&w.gen);
  return w.res[p];
}

// dynamic LDS: every `extern __shared__` symbol of the sources is one 160 KB array (generated dynlds.cpp registers them);
// under AddressSanitizer the part beyond what the launch asked for is poisoned, so a kernel that walks past its
// dynamic LDS size is reported like any other out-of-bounds access
namespace {
struct DynLds { unsigned char* base; size_t bytes; };
DynLds dyn_lds[32];
int n_dyn_lds = 0;
}  // namespace
void register_dynamic_lds(void* base, size_t bytes) {
  if (n_dyn_lds < 32) dyn_lds[n_dyn_lds++] = {(unsigned char*)base, bytes};
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& fn) {
#ifdef HIPEMU_ASAN
  for (int i = 0; i < n_dyn_lds; ++i) {
    const size_t keep = std::min(dyn_lds[i].bytes, (shmem + 7) & ~(size_t)7);
    ASAN_UNPOISON_MEMORY_REGION(dyn_lds[i].base, keep);
    ASAN_POISON_MEMORY_REGION(dyn_lds[i].base + keep, dyn_lds[i].bytes - keep);
  }
#else
  (void)shmem;
#endif
  if (cur >= 0) { fprintf(stderr, "hipemu: nested launch\n"); abort(); }
  const int n = (int)(block.x * block.y * block.z);
  if (n <= 0 || n > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", n); abort(); }
  if ((size_t)n > stacks_n) {
    if (stacks) munmap(stacks, stacks_n * STACK);
    stacks = (unsigned char*)mmap(nullptr, (size_t)n * STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == MAP_FAILED) { perror("hipemu mmap"); abort(); }
    stacks_n = (size_t)n;
  }
  fibres.assign((size_t)n, Fibre());
  const int nw = (n + 63) / 64;
  waves.resize((size_t)nw);
  blockDim = block;
  gridDim = grid;
  body = &fn;
  ++launch_no;
  HIPEMU_TSAN_RELEASE(&host_token);
#ifdef HIPEMU_TSAN
  main_tsan = __tsan_get_current_fiber();
  while (tsan_fibres.size() < (size_t)n) tsan_fibres.push_back(__tsan_create_fiber(0));
#endif
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        ++wg_no;
        for (int w = 0; w < nw; ++w) {
          waves[w].alive = std::min(64, n - 64 * w);
          waves[w].arrived = 0;
          waves[w].mask = waves[w].alive == 64 ? ~0ull : ((1ull << waves[w].alive) - 1);
        }
        blk_alive = n;
        blk_arrived = 0;
        for (int t = 0; t < n; ++t) {
          Fibre& f = fibres[t];
          f.done = false;
          f.par = 0;
          f.waiting = 0;
          f.nsync = 0;
          f.wave = t / 64;
          f.lane = t % 64;
          f.tid = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
          void** sp = (void**)(stacks + (size_t)(t + 1) * STACK - 64);
          for (int i = 0; i < 6; ++i) sp[i] = nullptr;
          sp[6] = (void*)&hipemu_entry;
          sp[7] = nullptr;
          f.sp = sp;
        }
        nfib = n;
        stalled = 0;
        cur = 0;
        threadIdx = fibres[0].tid;
        main_bottom = nullptr;
        tsan_target = 0;
        switch_stacks(&main_sp, &main_fake, fibres[0].sp, stack_of(0), STACK);  // back when the last thread has finished
        if (blk_alive != 0) { fprintf(stderr, "hipemu: scheduler returned with %d live threads\n", blk_alive); abort(); }
      }
  HIPEMU_TSAN_ACQUIRE(&done_token);
  cur = -1;
  body = nullptr;
}
}  // namespace hipemu
