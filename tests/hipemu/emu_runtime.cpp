// TEST-ONLY runtime of the HIP shim (see hip/hip_runtime.h): a pool of OS threads executes one
// workgroup at a time; __syncthreads() and the wave exchange barriers are condition-variable
// barriers so that 256 "GPU threads" can share 8 CPU cores without spinning.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

thread_local dim3 threadIdx;
dim3 blockIdx, blockDim, gridDim;
alignas(16) unsigned char cg_dyn_lds[160 * 1024];

namespace hipemu {
namespace {
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int count = 0, waiting = 0;
  unsigned long gen = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long g = gen;
    if (++waiting == count) { waiting = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

constexpr int kMaxThreads = 1024;
// Everything the detached workers wait on lives on the heap and is never destroyed: glibc's
// pthread_cond_destroy blocks while a waiter exists, which would hang interpreter exit.
struct State {
  Barrier block_barrier;
  Barrier wave_barrier[kMaxThreads / 64];
  std::mutex m;
  std::condition_variable cv_start[kMaxThreads / 64], cv_done;      // one start signal per wave: a 64-thread block wakes 64 workers
  std::vector<std::thread> pool;
  std::mutex launch_mutex;
};
State& S = *new State;
#define g_block_barrier S.block_barrier
#define g_wave_barrier S.wave_barrier
#define g_m S.m
#define g_cv_start S.cv_start
#define g_cv_done S.cv_done
#define g_pool S.pool
#define g_launch_mutex S.launch_mutex
alignas(16) unsigned char g_wave_slots[kMaxThreads / 64][64][16];
unsigned long g_job = 0;
int g_active = 0, g_done = 0;
const std::function<void()>* g_body = nullptr;

void worker(int id) {
  unsigned long seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lk(g_m);
      g_cv_start[id / 64].wait(lk, [&] { return g_job != seen && id < g_active; });
      seen = g_job;
    }
    threadIdx = dim3((unsigned)id, 0, 0);
    (*g_body)();
    {
      std::unique_lock<std::mutex> lk(g_m);
      if (++g_done == g_active) g_cv_done.notify_one();
    }
  }
}
}  // namespace

void syncthreads() { g_block_barrier.wait(); }
void wave_barrier(int wave) { g_wave_barrier[wave].wait(); }
void* wave_slot(int wave, int lane) { return g_wave_slots[wave][lane]; }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  std::lock_guard<std::mutex> guard(g_launch_mutex);
  const int n = (int)(block.x * block.y * block.z);
  if (n > kMaxThreads || shmem > sizeof(cg_dyn_lds) || block.y != 1 || block.z != 1) { fprintf(stderr, "hipemu: unsupported launch\n"); abort(); }
  while ((int)g_pool.size() < n) { const int id = (int)g_pool.size(); g_pool.emplace_back(worker, id); g_pool.back().detach(); }
  blockDim = block; gridDim = grid;
  g_block_barrier.count = n;
  for (int w = 0; w * 64 < n; ++w) g_wave_barrier[w].count = std::min(64, n - w * 64);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = dim3(bx, by, bz);
        std::unique_lock<std::mutex> lk(g_m);
        g_body = &body; g_active = n; g_done = 0; ++g_job;
        for (int w = 0; w * 64 < n; ++w) g_cv_start[w].notify_all();
        g_cv_done.wait(lk, [&] { return g_done == n; });
      }
}
}  // namespace hipemu
