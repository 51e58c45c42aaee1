// TEST-ONLY shim: lets the CPU-only build container execute the *same* .hip kernel sources that
// ship for gfx950, by mapping every HIP thread of a workgroup to a fiber of its own (real barriers, real
// shared memory semantics, wave-level shuffles through a per-wave exchange slot; see emu_runtime.cpp).  It exists so
// that indexing / synchronisation bugs surface in `pytest -m "not gpu"`; it is never shipped,
// never loaded by the product package, and has nothing to do with performance.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; } __attribute__((aligned(16)));
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
struct uint4 { unsigned x, y, z, w; } __attribute__((aligned(16)));
struct int4 { int x, y, z, w; } __attribute__((aligned(16)));
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

extern thread_local dim3 threadIdx, blockIdx;      // per OS thread: one workgroup at a time runs on each (its HIP threads are fibers)
extern dim3 blockDim, gridDim;

#define CG_OPAQUE_V(x) ((void)(x))
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }      // only ever applied to wave-uniform values
struct float2 { float x, y; } __attribute__((aligned(8)));
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#if defined(__SANITIZE_ADDRESS__)
#define __shared__ static                   // sanitizer build: ONE workgroup in flight (emu_runtime.cpp), plain statics have redzones (thread_local ones do not)
#else
#define __shared__ static thread_local      // one copy per OS thread = per workgroup in flight
#endif
#define HIP_DYNAMIC_SHARED(type, var) extern thread_local type var[];

typedef int hipError_t;
enum { hipSuccess = 0 };
typedef void* hipStream_t;
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }

using std::max;
using std::min;

namespace hipemu {
void syncthreads();
void wave_barrier(int wave);
void* wave_slot(int wave, int lane);   // 16 bytes per lane
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hipemu

static inline void __syncthreads() { hipemu::syncthreads(); }
// wave-level barrier / fence (ordering of LDS accesses between the lanes of a wave): a real barrier of the wave's 64 fibers here
static inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_barrier((int)threadIdx.x / 64); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)

template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  static_assert(sizeof(T) <= 16, "shuffle payload");
  const int lin = (int)threadIdx.x, lane = lin % 64, wave = lin / 64;
  *reinterpret_cast<T*>(hipemu::wave_slot(wave, lane)) = v;
  hipemu::wave_barrier(wave);
  const int src = lane + (int)delta;
  T r = v;
  if (src < 64 && (src / width) == (lane / width) && wave * 64 + src < (int)blockDim.x) r = *reinterpret_cast<T*>(hipemu::wave_slot(wave, src));
  hipemu::wave_barrier(wave);
  return r;
}

template <typename T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  static_assert(sizeof(T) <= 16, "shuffle payload");
  const int lin = (int)threadIdx.x, lane = lin % 64, wave = lin / 64;
  *reinterpret_cast<T*>(hipemu::wave_slot(wave, lane)) = v;
  hipemu::wave_barrier(wave);
  const int src = lane - (int)delta;
  T r = v;
  if (src >= 0 && (src / width) == (lane / width)) r = *reinterpret_cast<T*>(hipemu::wave_slot(wave, src));
  hipemu::wave_barrier(wave);
  return r;
}

// DPP lane exchange inside a row of 16 lanes: row_shl:n (0x100 + n: lane i takes lane i + n), row_shr:n (0x110 + n: lane i - n),
// row_ror:n (0x120 + n: rotation).  A lane whose source falls outside its row keeps `old` (bound_ctrl off) or takes 0.
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool bound_ctrl) {
  const int lin = (int)threadIdx.x, lane = lin % 64, wave = lin / 64, l15 = lane & 15, n = ctrl & 15, kind = ctrl & 0x1f0;
  *reinterpret_cast<int*>(hipemu::wave_slot(wave, lane)) = src;
  hipemu::wave_barrier(wave);
  int sl = -1;
  if (kind == 0x100) sl = l15 + n < 16 ? l15 + n : -1;
  else if (kind == 0x110) sl = l15 - n;
  else if (kind == 0x120) sl = (l15 - n) & 15;
  else { fprintf(stderr, "hipemu: unsupported dpp control 0x%x\n", ctrl); abort(); }
  int r = bound_ctrl ? 0 : old;
  if (sl >= 0 && wave * 64 + (lane - l15) + sl < (int)blockDim.x) r = *reinterpret_cast<int*>(hipemu::wave_slot(wave, (lane - l15) + sl));
  hipemu::wave_barrier(wave);
  return r;
}

template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  static_assert(sizeof(T) <= 16, "shuffle payload");
  const int lin = (int)threadIdx.x, lane = lin % 64, wave = lin / 64;
  *reinterpret_cast<T*>(hipemu::wave_slot(wave, lane)) = v;
  hipemu::wave_barrier(wave);
  const int src = lane ^ mask;
  T r = v;
  if (src < 64 && (src / width) == (lane / width) && wave * 64 + src < (int)blockDim.x) r = *reinterpret_cast<T*>(hipemu::wave_slot(wave, src));
  hipemu::wave_barrier(wave);
  return r;
}

// v_mfma_f32_16x16x4_f32 on one "wave": every lane contributes A[i = lane & 15][k = lane >> 4] and
// B[k = lane >> 4][j = lane & 15]; lane receives D[row = 4 * (lane >> 4) + reg][col = lane & 15].
typedef float hipemu_f32x4 __attribute__((vector_size(16)));
static inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  const int lin = (int)threadIdx.x, lane = lin % 64, wave = lin / 64;
  float* slot = reinterpret_cast<float*>(hipemu::wave_slot(wave, lane));
  slot[0] = a; slot[1] = b;
  hipemu::wave_barrier(wave);
  const int col = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (lane >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      const float av = reinterpret_cast<float*>(hipemu::wave_slot(wave, row + 16 * k))[0];
      const float bv = reinterpret_cast<float*>(hipemu::wave_slot(wave, col + 16 * k))[1];
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  hipemu::wave_barrier(wave);
  return c;
}

template <typename T>
static inline T hipemu_atomic_add_fp(T* p, T v) {
  using U = typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type;
  U* up = reinterpret_cast<U*>(p);
  U old = __atomic_load_n(up, __ATOMIC_RELAXED);
  for (;;) {
    T cur; memcpy(&cur, &old, sizeof(T));
    T nv = cur + v; U nu; memcpy(&nu, &nv, sizeof(T));
    if (__atomic_compare_exchange_n(up, &old, nu, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) return cur;
  }
}
static inline float atomicAdd(float* p, float v) { return hipemu_atomic_add_fp(p, v); }
static inline double atomicAdd(double* p, double v) { return hipemu_atomic_add_fp(p, v); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })
