// Helpers shared by the phase kernels (dstd_tail.hip, map2adj_tail.hip): BatchNorm constants of a chain, PReLU, matrix-core
// fragment reads.
#pragma once
#include "cg_common.h"
#include "dstd_tail.h"

typedef float cg_f32x4 __attribute__((vector_size(16)));

// ---- per-channel constants -------------------------------------------------------------------------------------------
struct CgAff { float mean, rstd, gamma, beta; };

// train: batch statistics from the replicated f64 sums (and, by the block that owns channel bookkeeping, save + running
// statistics exactly like nn.BatchNorm); eval: running statistics.  backward: the saved pair.
static __device__ __forceinline__ CgAff cg_tail_aff(const CgTailBN& bn, int c, int C, double cnt, int train, bool backward, bool owner) {
  CgAff a;
  a.gamma = bn.gamma[c]; a.beta = bn.beta[c];
  if (backward) { a.mean = bn.save[c]; a.rstd = bn.save[C + c]; return a; }
  if (train) {
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < CG_STAT_REPLICAS; ++r) { s1 += bn.stats[((long long)r * C + c) * 2]; s2 += bn.stats[((long long)r * C + c) * 2 + 1]; }
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    a.mean = (float)mean;
    a.rstd = (float)(1.0 / sqrt(var + (double)bn.eps));
    if (owner) {
      bn.save[c] = a.mean; bn.save[C + c] = a.rstd;
      if (bn.running_mean) {
        const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        bn.running_mean[c] = (1.f - bn.momentum) * bn.running_mean[c] + bn.momentum * (float)mean;
        bn.running_var[c] = (1.f - bn.momentum) * bn.running_var[c] + bn.momentum * (float)unb;
        if (c == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
      }
    }
  } else {
    a.mean = bn.running_mean[c];
    a.rstd = 1.0f / sqrtf(bn.running_var[c] + bn.eps);
    if (owner) { bn.save[c] = a.mean; bn.save[C + c] = a.rstd; }
  }
  return a;
}

static __device__ __forceinline__ float cg_bn(const CgAff& a, float v) { return (v - a.mean) * (a.gamma * a.rstd) + a.beta; }
static __device__ __forceinline__ float cg_prelu(float u, float alpha) { return u > 0.f ? u : alpha * u; }


// ======================================================================================================================
// matrix-core helpers (same fragment scheme as stgcn_domain_mfma.hip: inside a 16-wide k chunk step s takes k = 4*slot + s)
// ======================================================================================================================
template <int KIND>
__device__ __forceinline__ void cg_tfrag(const float* __restrict__ p, int rs, int k0, float v[4]) {
  if (KIND == 0) {
    const float4 t = *reinterpret_cast<const float4*>(p + k0);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    const float* q = p + k0 * rs;
    v[0] = q[0]; v[1] = q[rs]; v[2] = q[2 * rs]; v[3] = q[3 * rs];
  }
}
template <int KIND>
__device__ __forceinline__ const float* cg_tfrag_ptr(const float* base, int rs, int l15, int slot) {
  return KIND == 0 ? base + l15 * rs + 4 * slot : base + l15 + 4 * slot * rs;
}


// keep factors of the four consecutive elements idx0 .. idx0 + 3 (idx0 % 4 == 0): one hash, as cg_norm_act's float4 path
static __device__ __forceinline__ void cg_keep4(bool on, float p, unsigned long long seed, unsigned int salt, unsigned long long idx0, float keep[4]) {
  if (!on) { keep[0] = keep[1] = keep[2] = keep[3] = 1.f; return; }
  const unsigned long long bits = cg_drop_bits(seed, salt, idx0 >> 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) keep[j] = cg_drop_pick(bits, j, p);
}
