// Shared device helpers for the CIST-GCN gfx950 kernels.
// Wavefront = 64 lanes (CDNA4); every reduction below is written for that width.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CG_WAVE 64

// Makes a per-lane value opaque to the optimiser (no instruction is emitted).  Used at the top of long loop bodies so that the
// address arithmetic derived from it is recomputed per iteration instead of being hoisted into dozens of live registers.
// The CPU test shim (tests/hipemu) defines it as a no-op before this header is read.
#ifndef CG_OPAQUE_V
#define CG_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif

// ---- status codes returned through the C ABI (0 = ok, >0 = hipError_t, <0 = argument check) ----
#define CG_OK 0
#define CG_EARG (-1)
#define CG_ESHAPE (-2)

// BatchNorm channel sums are accumulated with f64 atomics by every workgroup of the producing kernel; with
// thousands of workgroups per channel one address serialises (~tens of microseconds).  Producers therefore add
// into replica (block id mod CG_STAT_REPLICAS) of a [CG_STAT_REPLICAS][C][2] buffer, the consumer sums the replicas.
#define CG_STAT_REPLICAS 16

// 4-D strided view: dims n[0..3] (n[0] = batch rows, n[1] = channel), element strides s[0..3].
// Layout-agnostic: NCTV, NTCV and (N,3,V,T) views of one buffer differ only in s[].
struct CgView4 {
  long long n[4];
  long long s[4];
};

// Zero fill by a kernel (rowops.hip).  hipMemsetAsync is NOT used anywhere in this library: as a node of a captured HIP
// graph it was not reliably ordered against the neighbouring kernels (see cg_zero in rowops.hip).
int cg_zero_fill(void* p, long long bytes, hipStream_t stream);

// Dynamic LDS above the default limit needs hipFuncAttributeMaxDynamicSharedMemorySize raised on the kernel.  That is a driver call:
// it used to run in front of EVERY launch of the LDS-heavy kernels (invisible under graph replay, pure host overhead in eager
// small-batch steps).  The limit a kernel has been given is remembered here (per translation unit: kernels are file-local), the driver
// is called only when a launch needs more than was granted before.
#include <mutex>
static inline hipError_t cg_lds_limit(const void* fn, size_t bytes) {
  struct Slot { const void* fn; size_t bytes; };
  static Slot slots[64];
  static int used = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int at = -1;
  for (int i = 0; i < used; ++i) if (slots[i].fn == fn) { at = i; break; }
  if (at >= 0 && slots[at].bytes >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  if (at < 0 && used < 64) at = used++;
  if (at >= 0) { slots[at].fn = fn; slots[at].bytes = bytes; }
  return hipSuccess;
}

static inline int cg_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CG_OK : (int)e;
}

// ---- wave / block reductions -------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T cg_wave_sum(T v) {
#pragma unroll
  for (int off = CG_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, CG_WAVE);
  return v;
}

// Sum over the whole workgroup; result valid in thread 0. `scratch` holds >= 16 T's in LDS.
// Contains two barriers; every thread of the block must call it.
template <typename T>
__device__ __forceinline__ T cg_block_sum(T v, T* scratch) {
  const int lane = threadIdx.x & (CG_WAVE - 1);
  const int wave = threadIdx.x / CG_WAVE;
  const int nw = (blockDim.x + CG_WAVE - 1) / CG_WAVE;
  v = cg_wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  T r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) r += scratch[w];
  return r;
}

// ---- counter-based RNG for dropout: splitmix64 of (seed, site salt, element index) ---------------
__device__ __forceinline__ uint32_t cg_rand_u32(unsigned long long seed, unsigned int salt, unsigned long long idx) {
  unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ull + ((unsigned long long)salt << 40) + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

// Dropout keep-scale: 0 (dropped) or 1/(1-p).  One 64-bit hash serves four consecutive elements (16 bits each), so
// the vectorised row kernels draw once per float4; the scalar path derives the same bits from (idx >> 2, idx & 3).
// The drop probability is therefore quantised to 1/65536.
// ---- lane exchange inside a row of 16 lanes on the VALU (DPP), no LDS round trip -----------------------------------------------
// CTRL: 0x100 + n row_shl (lane i takes lane i + n), 0x110 + n row_shr (lane i takes lane i - n), 0x120 + n row_ror (rotation).
// A lane whose source falls outside its row keeps `old`.
template <int CTRL>
__device__ __forceinline__ float cg_dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int cg_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
// sum over the 16 lanes of a row, in every lane of the row
__device__ __forceinline__ float cg_row16_sum(float v) {
  v += cg_dpp<0x128>(0.f, v); v += cg_dpp<0x124>(0.f, v); v += cg_dpp<0x122>(0.f, v); v += cg_dpp<0x121>(0.f, v);
  return v;
}
// the same in f64 (BatchNorm channel sums are kept in f64 end to end: the variance is formed as E[x^2] - E[x]^2): the two 32-bit
// halves travel through the DPP rotation, the additions are f64
template <int CTRL>
__device__ __forceinline__ double cg_dpp_f64(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = cg_dpp<CTRL>(0, (int)(unsigned)(u & 0xFFFFFFFFull)), hi = cg_dpp<CTRL>(0, (int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ double cg_row16_sum(double v) {
  v += cg_dpp_f64<0x128>(v); v += cg_dpp_f64<0x124>(v); v += cg_dpp_f64<0x122>(v); v += cg_dpp_f64<0x121>(v);
  return v;
}

// The gradient of a shared PReLU slope is a sum over the whole tensor.  One f64 word would take an atomic from every
// workgroup: same-address atomics serialise at ~20 ns each on MI355X (measured, profiles/README.md), 4096 of them cost more than
// the kernel's memory traffic.  Its partial sums are therefore spread over CG_ALPHA_SLOTS words behind the channel sums.
#define CG_ALPHA_SLOTS 64
__device__ __forceinline__ double cg_alpha_sum(const double* slots) {
  double s = 0.0;
  for (int i = 0; i < CG_ALPHA_SLOTS; ++i) s += slots[i];
  return s;
}

__device__ __forceinline__ unsigned long long cg_drop_bits(unsigned long long seed, unsigned int salt, unsigned long long quad) {
  unsigned long long z = quad + seed * 0x9E3779B97F4A7C15ull + ((unsigned long long)salt << 40) + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float cg_drop_pick(unsigned long long bits, int j, float p) {
  const uint32_t thr = (uint32_t)(p * 65536.0f);
  return ((uint32_t)(bits >> (16 * j)) & 0xFFFFu) >= thr ? 1.0f / (1.0f - p) : 0.0f;
}
__device__ __forceinline__ float cg_drop_scale(float p, unsigned long long seed, unsigned int salt, unsigned long long idx) {
  return cg_drop_pick(cg_drop_bits(seed, salt, idx >> 2), (int)(idx & 3), p);
}
