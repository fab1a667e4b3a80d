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

namespace hipemu {
namespace {
constexpr size_t STACK = 192 * 1024;

struct Wave {
  int alive = 0, arrived = 0;
  unsigned gen = 0;
  uint64_t mask = 0;
  alignas(64) unsigned char buf[2][64][64];
  alignas(64) unsigned char res[2][64][16];
};
struct Fibre {
  void* sp = nullptr;
  bool done = false;
  int wave = 0, lane = 0, par = 0;
  hipemu_idx tid;
};

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
    abort();
  }
  int nxt = cur;
  do { nxt = nxt + 1 == nfib ? 0 : nxt + 1; } while (fibres[nxt].done);
  if (nxt == cur) return;
  const int prev = cur;
  cur = nxt;
  threadIdx = fibres[nxt].tid;
  hipemu_switch(&fibres[prev].sp, fibres[nxt].sp);
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
  if (blk_alive == 0) {
    hipemu_switch(&f.sp, main_sp);
  } else {
    int nxt = cur;
    do { nxt = nxt + 1 == nfib ? 0 : nxt + 1; } while (fibres[nxt].done);
    cur = nxt;
    threadIdx = fibres[nxt].tid;
    hipemu_switch(&f.sp, fibres[nxt].sp);
  }
  abort();  // never resumed
}
extern "C" void hipemu_entry() {
  (*body)();
  fibre_exit();
}
}  // namespace

int lane() { return fibres[cur].lane; }
int wave_alive() { return waves[fibres[cur].wave].alive; }
uint64_t alive_mask() { return waves[fibres[cur].wave].mask; }

void sync_threads() {
  const unsigned g = blk_gen;
  if (++blk_arrived == blk_alive) { release_block(); return; }
  while (blk_gen == g) yield();
}
void wave_sync() {
  Wave& w = waves[fibres[cur].wave];
  const unsigned g = w.gen;
  if (++w.arrived == w.alive) { release_wave(w); return; }
  while (w.gen == g) yield();
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
  if (++w.arrived == w.alive) {
    fn(w.buf[p], w.res[p]);
    release_wave(w);
  } else {
    while (w.gen == g) yield();
  }
  return w.res[p];
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& fn) {
  (void)shmem;
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
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
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
        hipemu_switch(&main_sp, fibres[0].sp);  // comes back when the last thread of the workgroup has finished
        if (blk_alive != 0) { fprintf(stderr, "hipemu: scheduler returned with %d live threads\n", blk_alive); abort(); }
      }
  cur = -1;
  body = nullptr;
}
}  // namespace hipemu
