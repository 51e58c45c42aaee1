// TEST-ONLY runtime of the HIP shim (see hip/hip_runtime.h).  Every HIP thread of a workgroup is a user-level fiber with its own stack;
// the fibers of one workgroup run on ONE OS thread and hand the processor to each other only inside __syncthreads() and the wave
// exchanges (shuffles, DPP, MFMA), so a barrier costs a register swap instead of a futex round trip (the first version of this file
// mapped HIP threads to OS threads: 20 minutes of system time per run of the emulator tests).  Independent workgroups of a launch are
// spread over a small pool of OS threads - blockIdx, the LDS arrays and the wave exchange slots are per OS thread - which is also what
// makes an unsynchronised read-modify-write of global memory by two workgroups a real race here, as on the GPU.
//
// Scheduling is deterministic inside a workgroup: fibers run in thread order on even workgroups and in reverse order on odd ones, each
// until its next barrier, so a missing barrier between an LDS write and a read by another thread reads stale data on one of the two.
// Nothing about wave lock-step is assumed: a lane never observes another lane's progress except through a barrier.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#define HIPEMU_ASAN 1
#else
#define HIPEMU_ASAN 0
#endif

#if !defined(__x86_64__)
#error "tests/hipemu switches fibers with a few lines of x86-64 assembly; port hipemu_swap to run the emulator tests elsewhere"
#endif

thread_local dim3 threadIdx, blockIdx;
dim3 blockDim, gridDim;
alignas(16) thread_local unsigned char cg_dyn_lds[160 * 1024];

// hipemu_swap(&save, load): push the callee-saved registers, store the stack pointer in *save, continue on `load`.
extern "C" void hipemu_swap(void** save, void* load);
asm(R"(
.text
.globl hipemu_swap
.type hipemu_swap,@function
hipemu_swap:
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
.size hipemu_swap,.-hipemu_swap
)");

namespace hipemu {
namespace {
constexpr int kMaxThreads = 1024;
constexpr size_t kStack = 256 * 1024, kGuard = 4096;

struct Bar { int arrived = 0; unsigned long gen = 0; };

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;          // kStack bytes, the lowest page is a guard
  bool done = true;
  void* asan_fake = nullptr;
};

// One per OS thread of the pool.
struct Worker {
  std::vector<Fiber> fibers;
  int n = 0, cur = 0, dir = 1, live = 0;
  void* main_sp = nullptr;
  void* main_fake = nullptr; const void* main_bottom = nullptr; size_t main_size = 0;
  unsigned long progress = 0;     // arrivals at barriers, releases and exits: a round of the workgroup without any is a deadlock
  Bar block_bar, wave_bar[kMaxThreads / 64];
  int wave_count[kMaxThreads / 64];
  alignas(16) unsigned char slots[kMaxThreads / 64][64][16];
  const std::function<void()>* body = nullptr;
};
thread_local Worker* W = nullptr;

void fiber_entry();

void prepare(Worker& w, int id) {
  Fiber& f = w.fibers[id];
  if (!f.stack) {
    void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { perror("hipemu: mmap"); abort(); }
    mprotect(p, kGuard, PROT_NONE);
    f.stack = (char*)p;
  }
  // top is 16-byte aligned; after hipemu_swap's six pops and its ret the entry function sees the stack of a freshly called function
  void** top = (void**)(f.stack + kStack);
  top[-1] = nullptr;                         // return address of fiber_entry (never used)
  top[-2] = (void*)&fiber_entry;
  for (int i = 3; i <= 8; ++i) top[-i] = nullptr;
  f.sp = (void*)(top - 8);
  f.done = false;
  f.asan_fake = nullptr;
}

// Continue on fiber `to` (or on the OS thread's own stack when to < 0); `dying`: the current fiber will not run again.
void switch_to(Worker& w, int from, int to, bool dying) {
  void** save = from < 0 ? &w.main_sp : &w.fibers[from].sp;
  void* load = to < 0 ? w.main_sp : w.fibers[to].sp;
  w.cur = to;
  if (to >= 0) threadIdx = dim3((unsigned)to, 0, 0);
#if HIPEMU_ASAN
  void** fake = dying ? nullptr : (from < 0 ? &w.main_fake : &w.fibers[from].asan_fake);
  if (to < 0) __sanitizer_start_switch_fiber(fake, w.main_bottom, w.main_size);
  else __sanitizer_start_switch_fiber(fake, w.fibers[to].stack + kGuard, kStack - kGuard);
#else
  (void)dying;
#endif
  hipemu_swap(save, load);
#if HIPEMU_ASAN
  // running again (as `from`)
  __sanitizer_finish_switch_fiber(from < 0 ? w.main_fake : w.fibers[from].asan_fake, nullptr, nullptr);
#endif
}

int next_live(Worker& w, int from) {
  int i = from;
  for (int k = 0; k < w.n; ++k) {
    i += w.dir;
    if (i >= w.n) i = 0; else if (i < 0) i = w.n - 1;
    if (!w.fibers[i].done) return i;
  }
  return -1;
}

void fiber_entry() {
  Worker& w = *W;
  const int me = w.cur;
#if HIPEMU_ASAN
  {
    const void* bottom = nullptr; size_t size = 0;
    __sanitizer_finish_switch_fiber(nullptr, &bottom, &size);
    if (!w.main_bottom) { w.main_bottom = bottom; w.main_size = size; }      // the first fiber of a worker is entered from its own stack
  }
#endif
  (*w.body)();
  w.fibers[me].done = true;
  --w.live; ++w.progress;
  switch_to(w, me, w.live ? next_live(w, me) : -1, true);
  abort();        // a finished fiber is never resumed
}

void yield_from_barrier(Worker& w, const char* what) {
  const int me = w.cur, nx = next_live(w, me);
  if (nx < 0 || nx == me) {
    fprintf(stderr, "hipemu: thread %d of workgroup (%u,%u,%u) waits at a %s barrier the other threads of its group never reach\n", me,
            blockIdx.x, blockIdx.y, blockIdx.z, what);
    abort();
  }
  const unsigned long p = w.progress;
  switch_to(w, me, nx, false);
  if (w.progress == p) {
    fprintf(stderr, "hipemu: deadlock in workgroup (%u,%u,%u): thread %d waits at a %s barrier and no thread of the group can move\n",
            blockIdx.x, blockIdx.y, blockIdx.z, me, what);
    abort();
  }
}

void wait(Worker& w, Bar& b, int count, const char* what) {
  ++w.progress;
  if (++b.arrived == count) { b.arrived = 0; ++b.gen; return; }
  const unsigned long g = b.gen;
  while (b.gen == g) yield_from_barrier(w, what);
}

void run_block(Worker& w, dim3 bidx, int n, bool reverse, const std::function<void()>& body) {
  blockIdx = bidx;
  if ((int)w.fibers.size() < n) w.fibers.resize(n);
  w.n = n; w.live = n; w.dir = reverse ? -1 : 1; w.body = &body;
  w.block_bar = Bar();
  for (int v = 0; v * 64 < n; ++v) { w.wave_bar[v] = Bar(); w.wave_count[v] = std::min(64, n - v * 64); }
  for (int i = 0; i < n; ++i) prepare(w, i);
  switch_to(w, -1, reverse ? n - 1 : 0, false);
}

// ---- the pool: workers pull workgroup indices of the current launch from a shared counter ----
struct Pool {
  std::mutex m, launch_mutex;
  std::condition_variable cv_start, cv_done;
  std::vector<std::thread> threads;
  unsigned long job = 0;
  int busy = 0;
  dim3 grid; int n = 0; size_t shmem = 0; const std::function<void()>* body = nullptr;
  std::atomic<long> next{0};
  long total = 0;
};
// Heap-allocated and never destroyed: glibc's pthread_cond_destroy blocks while a waiter exists, which would hang interpreter exit.
Pool& P = *new Pool;

void drain(Worker& w) {
#if HIPEMU_ASAN
  // dynamic LDS beyond what the launch asked for is poisoned: an overrun of the requested size is reported like any other
  __asan_unpoison_memory_region(cg_dyn_lds, sizeof(cg_dyn_lds));
  const size_t used = (P.shmem + 15) & ~(size_t)15;
  if (used < sizeof(cg_dyn_lds)) __asan_poison_memory_region(cg_dyn_lds + used, sizeof(cg_dyn_lds) - used);
#endif
  for (;;) {
    const long i = P.next.fetch_add(1);
    if (i >= P.total) return;
    const unsigned bx = (unsigned)(i % P.grid.x), by = (unsigned)((i / P.grid.x) % P.grid.y), bz = (unsigned)(i / ((long)P.grid.x * P.grid.y));
    run_block(w, dim3(bx, by, bz), P.n, (i & 1) != 0, *P.body);
  }
}

void pool_thread() {
  W = new Worker;
  unsigned long seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(P.m);
      P.cv_start.wait(lk, [&] { return P.job != seen; });
      seen = P.job;
    }
    drain(*W);
    {
      std::unique_lock<std::mutex> lk(P.m);
      if (--P.busy == 0) P.cv_done.notify_one();
    }
  }
}

int pool_size() {
  static const int n = [] {
    if (HIPEMU_ASAN) return 1;      // statics stand in for LDS in the sanitizer build (hip_runtime.h): one workgroup at a time
    const char* e = getenv("HIPEMU_WORKERS");
    int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(v, 16));
  }();
  return n;
}
}  // namespace

void syncthreads() { Worker& w = *W; wait(w, w.block_bar, w.n, "workgroup"); }
void wave_barrier(int wave) { Worker& w = *W; wait(w, w.wave_bar[wave], w.wave_count[wave], "wave"); }
void* wave_slot(int wave, int lane) { return W->slots[wave][lane]; }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  std::lock_guard<std::mutex> guard(P.launch_mutex);
  const int n = (int)(block.x * block.y * block.z);
  if (n > kMaxThreads || n < 1 || shmem > sizeof(cg_dyn_lds) || block.y != 1 || block.z != 1) { fprintf(stderr, "hipemu: unsupported launch\n"); abort(); }
  const long total = (long)grid.x * grid.y * grid.z;
  if (total <= 0) return;
  blockDim = block; gridDim = grid;
  const int helpers = (int)std::min<long>(pool_size(), total);
  while ((int)P.threads.size() < helpers) { P.threads.emplace_back(pool_thread); P.threads.back().detach(); }
  std::unique_lock<std::mutex> lk(P.m);
  P.grid = grid; P.n = n; P.shmem = shmem; P.body = &body; P.total = total; P.next.store(0);
  P.busy = (int)P.threads.size(); ++P.job;
  P.cv_start.notify_all();
  P.cv_done.wait(lk, [&] { return P.busy == 0; });
}
}  // namespace hipemu
