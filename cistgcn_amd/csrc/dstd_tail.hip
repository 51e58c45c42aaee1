// Tail of a DSTD_GC block as phase kernels (reference: CISTGCN.py:266-269 `tcn` BatchNorm + Dropout + residual + PReLU of both
// Domain_GCNN layers, :388 `PReLU(BN(w * x))` of both branches + cat, :305-309 compressor 1x1 conv + BN + PReLU + SELayer2d,
// :390 block residual).  In train mode every BatchNorm needs the batch statistics of its input before the next operation
// can run, so the chain is cut exactly at those barriers and nowhere else; nothing but the compressor's pre-activation `h0`
// and the tensors autograd has to hand on (gradients at the barriers) touches HBM.
//
//   forward   F1  sums of z_i = w_i * PReLU_d(Dropout(BN_t(y_i)) + r_i)                    (reads y_i, r_i)
//             F2  a_i = PReLU_p(BN_p(z_i)) recomputed per tile -> h0 = Wc [a_1; a_2] on the matrix cores + sums of h0
//             F3  pooled[b,c] = mean_{t,v} PReLU_c(BN_c(h0))                                  (reads h0)
//             --  SELayer excitation (cg_se_gate_fwd)
//             F4  out = PReLU_c(BN_c(h0)) * gate + block residual, + sums of out for the next block's BatchNorm
//   backward  K1  dgate[b,c] = sum_{t,v} dout * h                                              (reads dout, h0)
//             --  SELayer excitation backward (cg_se_gate_bwd) -> dpooled
//             K2  sums of g_c = (dout * gate + dpooled / P) * PReLU_c'  (and g_c * h0_hat, d alpha_c)
//             K3  dh0 = BN_c'(g_c) per tile; d[a] = Wc^T dh0 and dWc += dh0 a^T on the matrix cores; g_p = d a * PReLU_p'
//                 -> HBM, + its BatchNorm sums
//             K4  dz = BN_p'(g_p); dw = sum dz * x; g_d = w * dz * PReLU_d' -> HBM (= gradient of the residual addend),
//                 + sums of the `tcn` BatchNorm
//             K5  dy_i = BN_t'(g_d * keep)
// Statistics are f64 sums in replicated slots (cg_common.h), reductions of the backward in f64 as in rowops.hip; the
// element arithmetic is f32 with the mean subtracted first, exactly as cg_norm_act does it, so the two paths agree to
// rounding.  Dropout draws are those of cg_norm_act (same counter-based hash of seed, site and element index).
#include "cg_common.h"
#include "dstd_tail.h"
#include "cg_phase.h"
#include <stdlib.h>

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

// rows of one channel per workgroup, as rowops.hip
static int cg_tail_rows(long long B, long long C, long long P) {
  long long target = (B * C * P) / 2048;
  target = target < 4096 ? 4096 : (target > 32768 ? 32768 : target);
  long long rb = target / (P > 0 ? P : 1);
  if (rb < 1) rb = 1;
  if (rb > B) rb = B;
  while (rb > 1 && C * ((B + rb - 1) / rb) < 512) rb = (rb + 1) / 2;
  return (int)rb;
}

// x_i = PReLU_d(Dropout(BN_t(y)) + r) for one element; also returns the pre-activation u and the keep factor
__device__ __forceinline__ float cg_tail_x(const CgAff& at, float alpha_d, float y, float r, float keep, float& u) {
  u = cg_bn(at, y) * keep + r;
  return cg_prelu(u, alpha_d);
}

__device__ __forceinline__ float cg_tail_keep(const CgDstdTail& t, int i, unsigned long long seed, int b, int c, int p) {
  if (!(t.train && t.drop_p > 0.f)) return 1.f;
  const long long P = (long long)t.T * t.V;
  return cg_drop_scale(t.drop_p, seed, t.salt[i], ((unsigned long long)b * t.C + c) * P + p);
}

__device__ __forceinline__ void cg_tail_keep4(const CgDstdTail& t, int i, unsigned long long seed, unsigned long long idx0, float keep[4]);

// ======================================================================================================================
// F1: channel sums of z_i = w_i * x_i     grid (C, batch chunks, 2)
// ======================================================================================================================
__global__ void cg_tail_f1_kernel(CgDstdTail t, int rb) {
  __shared__ double red[32];
  const int i = blockIdx.z, c = blockIdx.x, b0 = blockIdx.y * rb;
  if (b0 >= t.B) return;
  const int P = t.T * t.V, nb = min(rb, t.B - b0), C = t.C;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const double cnt = (double)t.B * P;
  const CgAff at = cg_tail_aff(t.bn_t[i], c, C, cnt, t.train, false, blockIdx.y == 0 && threadIdx.x == 0);
  const float ad = t.alpha_d[i][0];
  const unsigned long long seed = (t.train && t.drop_p > 0.f) ? *t.seed : 0ull;
  const float* __restrict__ y = t.y[i]; const float* __restrict__ r = t.r[i];
  float* const tap = t.tap_x[i];
  double s = 0.0, q = 0.0;
  // a wave per row (b, c), four consecutive positions per lane and step: 16-byte loads, one dropout hash per quad (the first form of
  // this kernel walked the elements one by one, a division and a hash each: 3.2 TB/s against 5.2 for K4, which was written this way)
  for (int br = wave; br < nb; br += nw) {
    const int b = b0 + br;
    const long long base = ((long long)b * C + c) * P;
    const float wv = t.w[i][(long long)b * C + c];
    if ((P & 3) == 0) {
      for (int p = 4 * lane; p < P; p += 256) {
        float keep[4];
        cg_tail_keep4(t, i, seed, (unsigned long long)(base + p), keep);
        const float4 y4 = *reinterpret_cast<const float4*>(y + base + p), r4 = *reinterpret_cast<const float4*>(r + base + p);
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
        float xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u;
          xv[j] = cg_tail_x(at, ad, yv[j], rv[j], keep[j], u);
          const double z = (double)(wv * xv[j]);
          s += z; q += z * z;                              // f64 per element, as every BatchNorm sum of this library (a quad summed in f32 first
        }                                                  // moved a cancelling bias gradient two layers down by 3e-3 of its size)
        if (tap) *reinterpret_cast<float4*>(tap + base + p) = make_float4(xv[0], xv[1], xv[2], xv[3]);
      }
    } else {
      for (int p = lane; p < P; p += 64) {
        float u;
        const float x = cg_tail_x(at, ad, y[base + p], r[base + p], cg_tail_keep(t, i, seed, b, c, p), u);
        const float z = wv * x;
        if (tap) tap[base + p] = x;
        s += (double)z; q += (double)z * (double)z;
      }
    }
  }
  s = cg_block_sum(s, red);
  q = cg_block_sum(q, red + 16);
  if (threadIdx.x == 0) {
    double* rep = t.bn_p[i].stats + (long long)(blockIdx.y % CG_STAT_REPLICAS) * 2 * C;
    atomicAdd(&rep[2 * c], s); atomicAdd(&rep[2 * c + 1], q);
  }
}

#define CG_TAIL_PT 64                 // positions per tile
#define CG_TAIL_PS (CG_TAIL_PT + 4)   // row stride of the [channel][position] images (== 4 mod 8)
#define CG_TAIL_THREADS 256
#define CG_TAIL_PT3 32                // positions per tile of the backward GEMM phase (two workgroups per CU)
#define CG_TAIL_PS3 (CG_TAIL_PT3 + 4)
#define CG_TAIL_K3_TASKS 2            // d a tasks of a wave per tile: (2C / 16 <= 8 channel tiles) * (PT3 / 32) / 4 waves

// per-channel constants of the forward chain of branch i, channel c, staged in LDS: [2C][8]
//   0 mean_t  1 scale_t  2 beta_t  3 alpha_d  4 mean_p  5 rstd_p  6 gamma_p  7 beta_p
__device__ __forceinline__ void cg_tail_consts(const CgDstdTail& t, float* sK, bool backward, bool owner) {
  const int C = t.C;
  const double cnt = (double)t.B * t.T * t.V;
  for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) {
    const int i = e / C, c = e - i * C;
    // train: F1 saved the tcn statistics; eval: F1 does not run, the running statistics are taken (and saved) here
    const CgAff at = cg_tail_aff(t.bn_t[i], c, C, cnt, t.train, backward || t.train != 0, owner);
    const CgAff ap = cg_tail_aff(t.bn_p[i], c, C, cnt, t.train, backward, owner);
    float* k = sK + 8 * e;
    k[0] = at.mean; k[1] = at.gamma * at.rstd; k[2] = at.beta; k[3] = t.alpha_d[i][0];
    k[4] = ap.mean; k[5] = ap.rstd; k[6] = ap.gamma; k[7] = ap.beta;
  }
}

// What the staging code of the matrix phases reads of the argument block for every quad, fetched ONCE per workgroup: the block lives in
// kernel-argument memory, and the slopes behind pointers; read inside the staging loops they cost a scalar (or global) load and a full
// wait per quad - found in the ISA in round 4.
struct CgTailHot {
  int C, P, train;
  float drop_p;
  unsigned int salt[2];
  float alpha_p[2];
  float* tap_a[2]; float* tap_x[2];
};
__device__ __forceinline__ CgTailHot cg_tail_hot(const CgDstdTail& t) {
  CgTailHot h;
  h.C = t.C; h.P = t.T * t.V; h.train = t.train; h.drop_p = t.drop_p;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    h.salt[i] = t.salt[i];
    h.alpha_p[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t.alpha_p[i][0])));
    h.tap_a[i] = t.tap_a[i]; h.tap_x[i] = t.tap_x[i];
  }
  return h;
}
__device__ __forceinline__ void cg_tail_keep4(const CgTailHot& h, int i, unsigned long long seed, unsigned long long idx0, float keep[4]) {
  if (!(h.train && h.drop_p > 0.f)) { keep[0] = keep[1] = keep[2] = keep[3] = 1.f; return; }
  const unsigned long long bits = cg_drop_bits(seed, h.salt[i], idx0 >> 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) keep[j] = cg_drop_pick(bits, j, h.drop_p);
}

// keep factors of the four consecutive elements idx0 .. idx0 + 3 (idx0 % 4 == 0): one hash, as cg_norm_act's float4 path
__device__ __forceinline__ void cg_tail_keep4(const CgDstdTail& t, int i, unsigned long long seed, unsigned long long idx0, float keep[4]) {
  if (!(t.train && t.drop_p > 0.f)) { keep[0] = keep[1] = keep[2] = keep[3] = 1.f; return; }
  const unsigned long long bits = cg_drop_bits(seed, t.salt[i], idx0 >> 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) keep[j] = cg_drop_pick(bits, j, t.drop_p);
}

// stage the activations of a tile: image[c2][p] for c2 in [0, 2C), p in [0, PT): what = 0: a = PReLU_p(BN_p(z)), 1: zhat.
// Four positions per work item (one 16-byte load of y and of r, one dropout hash) when the rows allow it (P % 4 == 0).
// Software-pipelined form of the staging below for rows of whole quads (P % 4 == 0): `load` issues every 16-byte read of a
// tile (PT / 8 quads of y and of r per thread at C <= 64) before anything waits, `finish` turns them into the LDS image.  The
// GEMM phases call `load` for tile k+1 right before the matrix-core work of tile k, so the reads travel while the MFMAs run.
template <int PT>
__device__ __forceinline__ void cg_tail_act_load(const CgDstdTail& t, int b, int p0, int np, float4 yq[PT / 8], float4 rq[PT / 8], float wq[PT / 8]) {
  // quads q < PT / 16 belong to branch 0, the others to branch 1: the branch (and with it the base pointers) is uniform per
  // quad slot, so the addresses are scalar base + 32-bit lane offset (C <= 64: C * PT / 4 work items per branch fit PT / 16 slots)
  const int C = t.C, P = t.T * t.V;
#pragma unroll
  for (int q = 0; q < PT / 8; ++q) {
    const int i = q / (PT / 16), e = threadIdx.x + (q - i * (PT / 16)) * CG_TAIL_THREADS;
    const int c = e / (PT / 4), pp = 4 * (e - c * (PT / 4));
    yq[q] = make_float4(0.f, 0.f, 0.f, 0.f); rq[q] = yq[q]; wq[q] = 0.f;
    if (c < C && pp < np) {
      const float* yb = t.y[i] + (long long)b * C * P + p0;
      const float* rb = t.r[i] + (long long)b * C * P + p0;
      const int off = c * P + pp;
      yq[q] = *reinterpret_cast<const float4*>(yb + off); rq[q] = *reinterpret_cast<const float4*>(rb + off);
      wq[q] = t.w[i][(long long)b * C + c];               // the row's gate travels with the row (a load inside `finish` is a round trip in front of its arithmetic)
    }
  }
}

template <int PT>
__device__ __forceinline__ void cg_tail_act_finish(const CgTailHot& t, const float* sK, unsigned long long seed, int b, int p0, int np,
                                                   const float4 yq[PT / 8], const float4 rq[PT / 8], const float wq[PT / 8], float* img, int what) {
  constexpr int PS = PT + 4;
  const int C = t.C, P = t.P;
#pragma unroll
  for (int q = 0; q < PT / 8; ++q) {
    const int i = q / (PT / 16), e = threadIdx.x + (q - i * (PT / 16)) * CG_TAIL_THREADS;       // same slots as cg_tail_act_load
    const int c = e / (PT / 4), pp = 4 * (e - c * (PT / 4)), c2 = i * C + c;
    if (c >= C) continue;
    float val[4] = {0.f, 0.f, 0.f, 0.f};
    if (pp < np) {
      const float4 k0 = *reinterpret_cast<const float4*>(sK + 8 * c2), k1 = *reinterpret_cast<const float4*>(sK + 8 * c2 + 4);
      const long long off = ((long long)b * C + c) * P + p0 + pp;
      const float yv[4] = {yq[q].x, yq[q].y, yq[q].z, yq[q].w}, rv[4] = {rq[q].x, rq[q].y, rq[q].z, rq[q].w};
      float keep[4];
      cg_tail_keep4(t, i, seed, (unsigned long long)off, keep);
      const float wv = wq[q], ap = t.alpha_p[i], scale_p = k1.z * k1.y;
      float xv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = ((yv[j] - k0.x) * k0.y + k0.z) * keep[j] + rv[j];
        xv[j] = cg_prelu(u, k0.w);
        const float z = wv * xv[j];
        val[j] = what == 0 ? cg_prelu((z - k1.x) * scale_p + k1.w, ap) : (z - k1.x) * k1.y;
      }
      if (what == 0) {
        if (t.tap_a[i]) *reinterpret_cast<float4*>(t.tap_a[i] + off) = make_float4(val[0], val[1], val[2], val[3]);
        if (!t.train && t.tap_x[i]) *reinterpret_cast<float4*>(t.tap_x[i] + off) = make_float4(xv[0], xv[1], xv[2], xv[3]);
      }
    }
    *reinterpret_cast<float4*>(img + c2 * PS + pp) = make_float4(val[0], val[1], val[2], val[3]);
  }
}

template <int PT>
__device__ __forceinline__ void cg_tail_stage_act(const CgDstdTail& t, const float* sK, unsigned long long seed, int b, int p0, int np,
                                                  float* img, int what) {
  constexpr int PS = PT + 4;
  const int C = t.C, P = t.T * t.V;
  if ((P & 3) == 0) {
    // branch by branch: the branch index (base pointers, gate row, slopes) is uniform, addresses are scalar base + 32-bit lane offset
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll 2
    for (int e = threadIdx.x; e < C * (PT / 4); e += CG_TAIL_THREADS) {
      const int c = e / (PT / 4), pp = 4 * (e - c * (PT / 4)), c2 = i * C + c;
      float val[4] = {0.f, 0.f, 0.f, 0.f};
      if (pp < np) {                                       // np % 4 == 0 here: a quad is inside the tile or outside
        const int p = p0 + pp;
        const float4 k0 = *reinterpret_cast<const float4*>(sK + 8 * c2), k1 = *reinterpret_cast<const float4*>(sK + 8 * c2 + 4);
        const long long off = ((long long)b * C + c) * P + p;
        const float4 y4 = *reinterpret_cast<const float4*>(t.y[i] + off), r4 = *reinterpret_cast<const float4*>(t.r[i] + off);
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
        float keep[4];
        cg_tail_keep4(t, i, seed, (unsigned long long)off, keep);
        const float wv = t.w[i][(long long)b * C + c], ap = t.alpha_p[i][0], scale_p = k1.z * k1.y;
        float xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = ((yv[j] - k0.x) * k0.y + k0.z) * keep[j] + rv[j];
          xv[j] = cg_prelu(u, k0.w);
          const float z = wv * xv[j];
          val[j] = what == 0 ? cg_prelu((z - k1.x) * scale_p + k1.w, ap) : (z - k1.x) * k1.y;
        }
        if (what == 0) {
          if (t.tap_a[i]) *reinterpret_cast<float4*>(t.tap_a[i] + off) = make_float4(val[0], val[1], val[2], val[3]);
          if (!t.train && t.tap_x[i]) *reinterpret_cast<float4*>(t.tap_x[i] + off) = make_float4(xv[0], xv[1], xv[2], xv[3]);
        }
      }
      *reinterpret_cast<float4*>(img + c2 * PS + pp) = make_float4(val[0], val[1], val[2], val[3]);
    }
    return;
  }
  for (int e = threadIdx.x; e < 2 * C * PT; e += CG_TAIL_THREADS) {
    const int c2 = e / PT, pp = e - c2 * PT;
    float val = 0.f;
    if (pp < np) {
      const int i = c2 / C, c = c2 - i * C, p = p0 + pp;
      const float* k = sK + 8 * c2;
      const long long off = ((long long)b * C + c) * P + p;
      const float keep = cg_tail_keep(t, i, seed, b, c, p);
      const float u = ((t.y[i][off] - k[0]) * k[1] + k[2]) * keep + t.r[i][off];
      const float z = t.w[i][(long long)b * C + c] * cg_prelu(u, k[3]);
      if (what == 0) {
        val = cg_prelu((z - k[4]) * (k[6] * k[5]) + k[7], t.alpha_p[i][0]);
        if (t.tap_a[i]) t.tap_a[i][off] = val;
        if (!t.train && t.tap_x[i]) t.tap_x[i][off] = cg_prelu(u, k[3]);       // eval: F1 does not run
      }
      else val = (z - k[4]) * k[5];                                    // zhat = (z - mean) * rstd
    }
    img[c2 * PS + pp] = val;
  }
}

// ======================================================================================================================
// F2: h0[b][co][p] = sum_c2 Wc[co][c2] a[c2][p]  + channel sums of h0.   Persistent workgroups over (sample, 64 positions)
// ======================================================================================================================
// MT_ / NT2_: 16-row tiles of the C output channels / of the 2C stacked input channels as compile-time constants (0: taken from t.C at
// run time - any width).  With the widths of the shipped configurations known to the compiler every matrix-core loop below unrolls with
// constant LDS offsets and independent accumulator chains; the run-time form left them rolled: LDS read, full wait, four dependent
// MFMAs per iteration (round 4, found in the ISA).
template <int MT_, int NT2_>
__global__ __launch_bounds__(CG_TAIL_THREADS, 2) void cg_tail_f2_kernel(CgDstdTail t, int tiles_per_sample, int total, int per) {
  const int C = t.C, C2 = 2 * C, CM = MT_ ? 16 * MT_ : (C + 15) & ~15, C2M = NT2_ ? 16 * NT2_ : (C2 + 15) & ~15, P = t.T * t.V;
  const int WS = C2M + 4;
  float* sAct = reinterpret_cast<float*>(cg_dyn_lds);            // [C2M][PS]
  float* sW = sAct + C2M * CG_TAIL_PS;                            // [CM][WS]
  float* sK = sW + CM * WS;                                       // [2C][8]
  double* sStat = reinterpret_cast<double*>(sK + 8 * C2M);        // [CM][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  const int wg = blockIdx.x;
  if (wg * per >= total) return;
  for (int e = tid; e < C2M * CG_TAIL_PS + CM * WS; e += CG_TAIL_THREADS) sAct[e] = 0.f;
  for (int e = tid; e < 2 * CM; e += CG_TAIL_THREADS) sStat[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < C * C2; e += CG_TAIL_THREADS) { const int co = e / C2, c2 = e - co * C2; sW[co * WS + c2] = t.Wc[e]; }
  cg_tail_consts(t, sK, false, wg == 0);
  const unsigned long long seed = (t.train && t.drop_p > 0.f) ? *t.seed : 0ull;
  const CgTailHot hot = cg_tail_hot(t);
  const int MT = CM / 16;
  constexpr int nw = CG_TAIL_THREADS / 64;
  const bool vec = (P & 3) == 0;
  float4 yq[CG_TAIL_PT / 8], rq[CG_TAIL_PT / 8];
  float wq[CG_TAIL_PT / 8];
  double fsum[MT_ ? (2 * MT_ + nw - 1) / nw : 1][2];
#pragma unroll
  for (int i = 0; i < (MT_ ? (2 * MT_ + nw - 1) / nw : 1); ++i) fsum[i][0] = fsum[i][1] = 0.0;
  if (vec) {
    const int lid = wg * per, b = lid / tiles_per_sample, p0 = (lid - b * tiles_per_sample) * CG_TAIL_PT;
    cg_tail_act_load<CG_TAIL_PT>(t, b, p0, min(CG_TAIL_PT, P - p0), yq, rq, wq);
  }
  for (int it = 0; it < per; ++it) {
    const int lid = wg * per + it;
    if (lid >= total) break;
    const int b = lid / tiles_per_sample, tile = lid - b * tiles_per_sample, p0 = tile * CG_TAIL_PT, np = min(CG_TAIL_PT, P - p0);
    __syncthreads();
    if (vec) cg_tail_act_finish<CG_TAIL_PT>(hot, sK, seed, b, p0, np, yq, rq, wq, sAct, 0);
    else cg_tail_stage_act<CG_TAIL_PT>(t, sK, seed, b, p0, np, sAct, 0);
    __syncthreads();
    if (vec && it + 1 < per && lid + 1 < total) {          // the next tile's reads travel while the matrix cores work
      const int l2 = lid + 1, b2 = l2 / tiles_per_sample, q0 = (l2 - b2 * tiles_per_sample) * CG_TAIL_PT;
      cg_tail_act_load<CG_TAIL_PT>(t, b2, q0, min(CG_TAIL_PT, P - q0), yq, rq, wq);
    }
    float* hb = t.h0 + (long long)b * C * P + p0;
    // (co tile, pair of position tiles): MT * 2 tasks.  Known widths: the (up to two) tasks of a wave - co tiles w >> 1 and (w >> 1) + 2 on
    // the SAME position pair - advance through k together: the activation fragments are read once, four independent MFMA chains
    constexpr int NTASK = MT_ ? (2 * MT_ + nw - 1) / nw : 1;
    cg_f32x4 facc[NTASK][2];
    if (MT_) {
#pragma unroll
      for (int ti = 0; ti < NTASK; ++ti) facc[ti][0] = facc[ti][1] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
      if (2 * MT_ >= nw || wave < 2 * MT_) {
        const int n0 = 32 * (wave & 1);
#pragma unroll
        for (int k0 = 0; k0 < 16 * NT2_; k0 += 16) {                        // rows >= 2C of sAct and columns >= 2C of sW are zero
          float b0v[4], b1v[4];
          cg_tfrag<1>(cg_tfrag_ptr<1>(sAct + n0, CG_TAIL_PS, l15, slot), CG_TAIL_PS, k0, b0v);
          cg_tfrag<1>(cg_tfrag_ptr<1>(sAct + n0 + 16, CG_TAIL_PS, l15, slot), CG_TAIL_PS, k0, b1v);
#pragma unroll
          for (int ti = 0; ti < NTASK; ++ti) {
            float av[4];
            cg_tfrag<0>(cg_tfrag_ptr<0>(sW + 16 * ((wave >> 1) + (nw / 2) * ti) * WS, WS, l15, slot), WS, k0, av);
#pragma unroll
            for (int s = 0; s < 4; ++s) {                                   // C[position 4 * slot + q][output channel l15]
              facc[ti][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0v[s], av[s], facc[ti][0], 0, 0, 0);
              facc[ti][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1v[s], av[s], facc[ti][1], 0, 0, 0);
            }
          }
        }
      }
    }
#pragma unroll
    for (int ti = 0; ti < (MT_ ? NTASK : 2); ++ti) {                      // C <= 64: at most eight tasks
      const int w = wave + nw * ti;
      if (w >= MT * 2) break;
      const int mt = w >> 1, n0 = 32 * (w & 1), n1 = n0 + 16;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      if (MT_) { c0 = facc[MT_ ? ti : 0][0]; c1 = facc[MT_ ? ti : 0][1]; }
      else {
        const float* ap = cg_tfrag_ptr<0>(sW + 16 * mt * WS, WS, l15, slot);
        const float* bp0 = cg_tfrag_ptr<1>(sAct + n0, CG_TAIL_PS, l15, slot);
        const float* bp1 = cg_tfrag_ptr<1>(sAct + n1, CG_TAIL_PS, l15, slot);
        for (int k0 = 0; k0 < C2M; k0 += 16) {                              // rows >= 2C of sAct and columns >= 2C of sW are zero
          float av[4], b0v[4], b1v[4];
          cg_tfrag<0>(ap, WS, k0, av); cg_tfrag<1>(bp0, CG_TAIL_PS, k0, b0v); cg_tfrag<1>(bp1, CG_TAIL_PS, k0, b1v);
#pragma unroll
          for (int s = 0; s < 4; ++s) {                                     // C[position 4 * slot + q][output channel l15]
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0v[s], av[s], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1v[s], av[s], c1, 0, 0, 0);
          }
        }
      }
      // position-major tiles: a lane holds four consecutive positions of ONE output channel -> 16-byte stores of h0, channel sums
      // per lane, combined over the four lanes l15 + 16 * slot of a channel by two exchanges
      const int co = 16 * mt + l15;
      const bool cok = co < C;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pq = (h ? n1 : n0) + 4 * slot;
        const cg_f32x4 cc = h ? c1 : c0;
        if (cok && pq < np) {
          float* dst = hb + (long long)co * P + pq;
          if (vec) *reinterpret_cast<float4*>(dst) = make_float4(cc[0], cc[1], cc[2], cc[3]);      // np % 4 == 0: the quad is inside the row
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (pq + q < np) {
              if (!vec) dst[q] = cc[q];
              s1 += cc[q]; s2 += cc[q] * cc[q];
            }
          }
        }
      }
      if (t.train) {
        if (MT_) { fsum[MT_ ? ti : 0][0] += (double)s1; fsum[MT_ ? ti : 0][1] += (double)s2; }      // known widths: task ti of a wave is the same channel tile in every tile
        else {
          s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
          s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
          if (slot == 0 && cok) { atomicAdd(&sStat[2 * co], (double)s1); atomicAdd(&sStat[2 * co + 1], (double)s2); }
        }
      }
    }
  }
  if (t.train && MT_) {
    // the channel sums of a lane's tasks, kept in registers over all tiles: one pair of LDS atomics per task and workgroup instead of
    // two exchanges and a pair of atomics per task and TILE
#pragma unroll
    for (int ti = 0; ti < (MT_ ? (2 * MT_ + nw - 1) / nw : 1); ++ti) {
      const int w = wave + nw * ti, co = 16 * (w >> 1) + l15;
      if (w < MT * 2 && co < C) { atomicAdd(&sStat[2 * co], fsum[ti][0]); atomicAdd(&sStat[2 * co + 1], fsum[ti][1]); }
    }
  }
  if (t.train) {
    __syncthreads();
    double* rep = t.bn_c.stats + (long long)(blockIdx.x % CG_STAT_REPLICAS) * 2 * C;
    for (int e = tid; e < 2 * C; e += CG_TAIL_THREADS) atomicAdd(&rep[e], sStat[e]);
  }
}

// ======================================================================================================================
// F3: pooled[b,c] = mean_p PReLU_c(BN_c(h0))     one workgroup per (c, b) row
// F4: out = h * gate[b,c] + bres   (+ channel sums of out)
// ======================================================================================================================
__global__ void cg_tail_f3_kernel(CgDstdTail t, int rb) {
  const int c = blockIdx.x, b0 = blockIdx.y * rb, P = t.T * t.V, C = t.C;
  if (b0 >= t.B) return;
  const int nb = min(rb, t.B - b0), lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const CgAff ac = cg_tail_aff(t.bn_c, c, C, (double)t.B * P, t.train, false, blockIdx.y == 0 && threadIdx.x == 0);
  const float alpha = t.alpha_c[0];
  for (int br = wave; br < nb; br += nw) {                 // a wave per row: its mean needs no LDS (one workgroup per row was 16 K tiny workgroups)
    const int b = b0 + br;
    const float* __restrict__ h0 = t.h0 + ((long long)b * C + c) * P;
    float s = 0.f;
    if ((P & 3) == 0) {
      for (int p = 4 * lane; p < P; p += 256) {
        const float4 h4 = *reinterpret_cast<const float4*>(h0 + p);
        s += (cg_prelu(cg_bn(ac, h4.x), alpha) + cg_prelu(cg_bn(ac, h4.y), alpha)) + (cg_prelu(cg_bn(ac, h4.z), alpha) + cg_prelu(cg_bn(ac, h4.w), alpha));
      }
    } else {
      for (int p = lane; p < P; p += 64) s += cg_prelu(cg_bn(ac, h0[p]), alpha);
    }
    const double sd = cg_wave_sum((double)s);
    if (lane == 0) t.pooled[(long long)b * C + c] = (float)(sd / (double)P);
  }
}

__global__ void cg_tail_f4_kernel(CgDstdTail t, int rb) {
  __shared__ double red[32];
  const int c = blockIdx.x, b0 = blockIdx.y * rb;
  if (b0 >= t.B) return;
  const int P = t.T * t.V, nb = min(rb, t.B - b0), C = t.C;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const CgAff ac = cg_tail_aff(t.bn_c, c, C, (double)t.B * P, t.train, true, false);      // saved by F3
  const float alpha = t.alpha_c[0];
  float* const tap = t.tap_h;
  double s = 0.0, q = 0.0;
  for (int br = wave; br < nb; br += nw) {
    const int b = b0 + br;
    const long long base = ((long long)b * C + c) * P;
    const float gate = t.gate[(long long)b * C + c];
    if ((P & 3) == 0) {
      for (int p = 4 * lane; p < P; p += 256) {
        const float4 h4 = *reinterpret_cast<const float4*>(t.h0 + base + p), r4 = *reinterpret_cast<const float4*>(t.bres + base + p);
        const float hv[4] = {h4.x, h4.y, h4.z, h4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
        float h[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = cg_prelu(cg_bn(ac, hv[j]), alpha); v[j] = h[j] * gate + rv[j]; s += (double)v[j]; q += (double)v[j] * (double)v[j]; }
        if (tap) *reinterpret_cast<float4*>(tap + base + p) = make_float4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<float4*>(t.out + base + p) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
      for (int p = lane; p < P; p += 64) {
        const float h = cg_prelu(cg_bn(ac, t.h0[base + p]), alpha);
        if (tap) tap[base + p] = h;
        const float v = h * gate + t.bres[base + p];
        t.out[base + p] = v;
        s += (double)v; q += (double)v * (double)v;
      }
    }
  }
  if (t.ostats) {
    s = cg_block_sum(s, red);
    q = cg_block_sum(q, red + 16);
    if (threadIdx.x == 0) {
      double* rep = t.ostats + (long long)(blockIdx.y % CG_STAT_REPLICAS) * 2 * C;
      atomicAdd(&rep[2 * c], s); atomicAdd(&rep[2 * c + 1], q);
    }
  }
}

// ======================================================================================================================
// backward
// ======================================================================================================================
// K1: dgate[b,c] = sum_p dout * h
__global__ void cg_tail_k1_kernel(CgDstdTail t, int rb) {
  const int c = blockIdx.x, b0 = blockIdx.y * rb, P = t.T * t.V, C = t.C;
  if (b0 >= t.B) return;
  const int nb = min(rb, t.B - b0), lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const CgAff ac = cg_tail_aff(t.bn_c, c, C, 0.0, t.train, true, false);
  const float alpha = t.alpha_c[0];
  for (int br = wave; br < nb; br += nw) {
    const int b = b0 + br;
    const long long base = ((long long)b * C + c) * P;
    float s = 0.f;
    if ((P & 3) == 0) {
      for (int p = 4 * lane; p < P; p += 256) {
        const float4 d4 = *reinterpret_cast<const float4*>(t.dout + base + p), h4 = *reinterpret_cast<const float4*>(t.h0 + base + p);
        s += (d4.x * cg_prelu(cg_bn(ac, h4.x), alpha) + d4.y * cg_prelu(cg_bn(ac, h4.y), alpha)) +
             (d4.z * cg_prelu(cg_bn(ac, h4.z), alpha) + d4.w * cg_prelu(cg_bn(ac, h4.w), alpha));
      }
    } else {
      for (int p = lane; p < P; p += 64) s += t.dout[base + p] * cg_prelu(cg_bn(ac, t.h0[base + p]), alpha);
    }
    const double sd = cg_wave_sum((double)s);
    if (lane == 0) t.dgate[(long long)b * C + c] = (float)sd;
  }
}

// gradient in front of the compressor's PReLU for one element: g_c = (dout * gate + dpooled / P) * PReLU_c'(u); returns g_c, u
__device__ __forceinline__ float cg_tail_gc(const CgDstdTail& t, const CgAff& ac, float alpha, long long off, int b, int c, float invP, float& u) {
  u = cg_bn(ac, t.h0[off]);
  const float dh = t.dout[off] * t.gate[(long long)b * t.C + c] + t.dpooled[(long long)b * t.C + c] * invP;
  return u > 0.f ? dh : alpha * dh;
}

// K2: red_c[c] = { sum g_c, sum g_c * h0hat }, red_c[2C + slot] += sum_{u <= 0} dh * u   (d alpha_c, CG_ALPHA_SLOTS partial sums)
__global__ void cg_tail_k2_kernel(CgDstdTail t, int rb) {
  __shared__ double red[48];
  const int c = blockIdx.x, b0 = blockIdx.y * rb;
  if (b0 >= t.B) return;
  const int P = t.T * t.V, nb = min(rb, t.B - b0), C = t.C;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const CgAff ac = cg_tail_aff(t.bn_c, c, C, 0.0, t.train, true, false);
  const float alpha = t.alpha_c[0], invP = 1.f / (float)P;
  double s1 = 0.0, s2 = 0.0, sa = 0.0;
  for (int br = wave; br < nb; br += nw) {
    const int b = b0 + br;
    const long long base = ((long long)b * C + c) * P;
    const float gate = t.gate[(long long)b * C + c], dp = t.dpooled[(long long)b * C + c] * invP;
    if ((P & 3) == 0) {
      for (int p = 4 * lane; p < P; p += 256) {
        const float4 d4 = *reinterpret_cast<const float4*>(t.dout + base + p), h4 = *reinterpret_cast<const float4*>(t.h0 + base + p);
        const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = cg_bn(ac, hv[j]), dh = dv[j] * gate + dp;
          const float g = u > 0.f ? dh : alpha * dh;
          s1 += (double)g; s2 += (double)g * (double)((hv[j] - ac.mean) * ac.rstd);
          sa += u > 0.f ? 0.0 : (double)dh * (double)u;
        }
      }
    } else {
      for (int p = lane; p < P; p += 64) {
        float u;
        const float g = cg_tail_gc(t, ac, alpha, base + p, b, c, invP, u);
        s1 += (double)g;
        s2 += (double)g * (double)((t.h0[base + p] - ac.mean) * ac.rstd);
        if (!(u > 0.f)) sa += (double)(t.dout[base + p] * gate + dp) * (double)u;
      }
    }
  }
  s1 = cg_block_sum(s1, red); s2 = cg_block_sum(s2, red + 16); sa = cg_block_sum(sa, red + 32);
  if (threadIdx.x == 0) {
    atomicAdd(&t.red_c[2 * c], s1); atomicAdd(&t.red_c[2 * c + 1], s2); atomicAdd(&t.red_c[2 * C + ((c + 5 * (int)blockIdx.y) & (CG_ALPHA_SLOTS - 1))], sa);
  }
}

// Diagnostic build only (tools/stamps_tail.py compiles a private copy of the library with -DCG_TAIL_STAMPS): thread 0 of every
// workgroup stores the shader clock at the phase boundaries of K3; nothing depends on it, the shipped library has no stamp.
#ifdef CG_TAIL_STAMPS
__device__ unsigned long long* cg_tail_stamp_buf = nullptr;
extern "C" int cg_tail_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(cg_tail_stamp_buf), &p, sizeof(p)); }
#define CG_TSTAMP()                                                                                     \
  do {                                                                                                  \
    if (threadIdx.x == 0 && cg_tail_stamp_buf && nst < 255) cg_tail_stamp_buf[blockIdx.x * 256 + (++nst)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define CG_TSTAMP_END() do { if (threadIdx.x == 0 && cg_tail_stamp_buf) cg_tail_stamp_buf[blockIdx.x * 256] = nst; } while (0)
#else
#define CG_TSTAMP() do { } while (0)
#define CG_TSTAMP_END() do { } while (0)
#endif

// K3: per tile: dh0 (BatchNorm backward of g_c), d a = Wc^T dh0, dWc += dh0 a^T, g_p = d a * PReLU_p' -> HBM + its sums
template <int MT_, int NT2_>
__global__ __launch_bounds__(CG_TAIL_THREADS, 2) void cg_tail_k3_kernel(CgDstdTail t, int tiles_per_sample, int total, int per, int replicas) {
  const int C = t.C, C2 = 2 * C, CM = MT_ ? 16 * MT_ : (C + 15) & ~15, C2M = NT2_ ? 16 * NT2_ : (C2 + 15) & ~15, P = t.T * t.V;
  const int WS = C2M + 4;
  float* sZ = reinterpret_cast<float*>(cg_dyn_lds);              // [C2M][PS]  zhat of both branches
  float* sDH = sZ + C2M * CG_TAIL_PS3;                             // [CM][PS]   dh0
  float* sW = sDH + CM * CG_TAIL_PS3;                              // [CM][WS]
  float* sK = sW + CM * WS;                                       // [2C][8]
  float* sKc = sK + 8 * C2M;                                      // [CM][8] compressor BatchNorm backward constants
  double* sRed = reinterpret_cast<double*>(sKc + 8 * CM);         // [C2M][2] sums of g_p, g_p * zhat ; then [2] d alpha_p
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4, nw = CG_TAIL_THREADS / 64;
  const int wg = blockIdx.x;
  if (wg * per >= total) return;
  int nst = 0; (void)nst;
  CG_TSTAMP();
  for (int e = tid; e < C2M * CG_TAIL_PS3 + CM * CG_TAIL_PS3 + CM * WS; e += CG_TAIL_THREADS) sZ[e] = 0.f;
  for (int e = tid; e < 2 * C2M + 2; e += CG_TAIL_THREADS) sRed[e] = 0.0;
  __syncthreads();
  for (int e = tid; e < C * C2; e += CG_TAIL_THREADS) { const int co = e / C2, c2 = e - co * C2; sW[co * WS + c2] = t.Wc[e]; }
  cg_tail_consts(t, sK, true, false);
  const double cnt = (double)t.B * P;
  for (int c = tid; c < C; c += CG_TAIL_THREADS) {
    const CgAff ac = cg_tail_aff(t.bn_c, c, C, 0.0, t.train, true, false);
    float* kc = sKc + 8 * c;
    kc[0] = ac.mean; kc[1] = ac.rstd; kc[2] = ac.gamma * ac.rstd; kc[3] = ac.beta;
    kc[4] = t.train ? (float)(t.red_c[2 * c] / cnt) : 0.f; kc[5] = t.train ? (float)(t.red_c[2 * c + 1] / cnt) : 0.f;
  }
  const unsigned long long seed = (t.train && t.drop_p > 0.f) ? *t.seed : 0ull;
  const CgTailHot hot = cg_tail_hot(t);
  const bool train = t.train != 0;
  float* const gp0 = t.gp[0]; float* const gp1 = t.gp[1];
  const int MT = CM / 16, NT2 = C2M / 16;
  const float invP = 1.f / (float)P;
  // dWc accumulators: tile (mt, n2) for id = u * nw + wave, kept in registers across all tiles of this workgroup
  cg_f32x4 wacc[CG_TAIL_MAXW];
#pragma unroll
  for (int u = 0; u < CG_TAIL_MAXW; ++u) wacc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  const float alpha_c = t.alpha_c[0];
  const float alpha_p0 = t.alpha_p[0][0], alpha_p1 = t.alpha_p[1][0];      // once: a global load inside the matrix phases stalls every task
  float racc[CG_TAIL_K3_TASKS][3];                 // per lane: sums of g_p, g_p * zhat and the d alpha_p terms of its channel in the wave's d a tasks
#pragma unroll
  for (int i = 0; i < CG_TAIL_K3_TASKS * 3; ++i) (&racc[0][0])[i] = 0.f;
  CG_TSTAMP();
  // Loads of a tile travel while something else runs: y / r / h0 / dout / gate / dpooled of tile k+1 are issued in front of the
  // matrix phases of tile k (registers yq / rq / h4 / d4 / gt / dp)
  const bool vec = (P & 3) == 0;
  constexpr int DQ = CG_TAIL_PT3 / 16;               // quads of the dh0 image per thread (C <= 64)
  float4 yq[CG_TAIL_PT3 / 8], rq[CG_TAIL_PT3 / 8];
  float wq[CG_TAIL_PT3 / 8];
  float4 h4[DQ], d4[DQ];
  float gt[DQ], dp[DQ];
  auto dh_load = [&](int b, int p0, int np) {            // h0 / dout quads and the per-row gate terms of a tile's dh0 image
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      const int e = tid + q * CG_TAIL_THREADS, c = e / (CG_TAIL_PT3 / 4), pp = 4 * (e - c * (CG_TAIL_PT3 / 4));
      h4[q] = make_float4(0.f, 0.f, 0.f, 0.f); d4[q] = h4[q]; gt[q] = 0.f; dp[q] = 0.f;
      if (c < C && pp < np) {
        const long long off = ((long long)b * C + c) * P + p0 + pp;
        h4[q] = *reinterpret_cast<const float4*>(t.h0 + off); d4[q] = *reinterpret_cast<const float4*>(t.dout + off);
        gt[q] = t.gate[(long long)b * C + c]; dp[q] = t.dpooled[(long long)b * C + c];     // raw: arithmetic on a loaded value would wait for it here
      }
    }
  };
  if (vec) {
    const int lid = wg * per, b = lid / tiles_per_sample, p0 = (lid - b * tiles_per_sample) * CG_TAIL_PT3;
    cg_tail_act_load<CG_TAIL_PT3>(t, b, p0, min(CG_TAIL_PT3, P - p0), yq, rq, wq);
    dh_load(b, p0, min(CG_TAIL_PT3, P - p0));
  }
  for (int it = 0; it < per; ++it) {
    const int lid = wg * per + it;
    if (lid >= total) break;
    const int b = lid / tiles_per_sample, tile = lid - b * tiles_per_sample, p0 = tile * CG_TAIL_PT3, np = min(CG_TAIL_PT3, P - p0);
    __syncthreads();
    CG_TSTAMP();
    if (vec) {
      cg_tail_act_finish<CG_TAIL_PT3>(hot, sK, seed, b, p0, np, yq, rq, wq, sZ, 1);
      CG_TSTAMP();
      // dh0 = gamma_c * rstd * (g_c - mean(g_c) - h0hat * mean(g_c h0hat)); per-channel constants from sKc
#pragma unroll
      for (int q = 0; q < DQ; ++q) {
        const int e = tid + q * CG_TAIL_THREADS, c = e / (CG_TAIL_PT3 / 4), pp = 4 * (e - c * (CG_TAIL_PT3 / 4));
        if (c >= C) continue;
        float val[4] = {0.f, 0.f, 0.f, 0.f};
        if (pp < np) {
          const float4 kcA = *reinterpret_cast<const float4*>(sKc + 8 * c);       // mean, rstd, scale = gamma * rstd, beta
          const float2 kcB = *reinterpret_cast<const float2*>(sKc + 8 * c + 4);   // m1, m2
          const float hv[4] = {h4[q].x, h4[q].y, h4[q].z, h4[q].w}, dv[4] = {d4[q].x, d4[q].y, d4[q].z, d4[q].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float u = (hv[j] - kcA.x) * kcA.z + kcA.w;
            const float dh = dv[j] * gt[q] + dp[q] * invP;
            const float g = u > 0.f ? dh : alpha_c * dh;
            val[j] = train ? kcA.z * (g - kcB.x - (hv[j] - kcA.x) * kcA.y * kcB.y) : g * kcA.z;
          }
        }
        *reinterpret_cast<float4*>(sDH + c * CG_TAIL_PS3 + pp) = make_float4(val[0], val[1], val[2], val[3]);
      }
    } else {
      cg_tail_stage_act<CG_TAIL_PT3>(t, sK, seed, b, p0, np, sZ, 1);
      CG_TSTAMP();
      for (int e = tid; e < C * CG_TAIL_PT3; e += CG_TAIL_THREADS) {
        const int c = e / CG_TAIL_PT3, pp = e - c * CG_TAIL_PT3;
        float val = 0.f;
        if (pp < np) {
          const float* kc = sKc + 8 * c;
          const long long off = ((long long)b * C + c) * P + p0 + pp;
          const float hv = t.h0[off];
          const float u = (hv - kc[0]) * kc[2] + kc[3];
          const float dh = t.dout[off] * t.gate[(long long)b * C + c] + t.dpooled[(long long)b * C + c] * invP;
          const float g = u > 0.f ? dh : alpha_c * dh;
          val = t.train ? kc[2] * (g - kc[4] - (hv - kc[0]) * kc[1] * kc[5]) : g * kc[2];
        }
        sDH[c * CG_TAIL_PS3 + pp] = val;
      }
    }
    __syncthreads();
    if (vec && it + 1 < per && lid + 1 < total) {
      const int b2 = (lid + 1) / tiles_per_sample, q0 = (lid + 1 - b2 * tiles_per_sample) * CG_TAIL_PT3;
      cg_tail_act_load<CG_TAIL_PT3>(t, b2, q0, min(CG_TAIL_PT3, P - q0), yq, rq, wq);
      dh_load(b2, q0, min(CG_TAIL_PT3, P - q0));
    }
    CG_TSTAMP();
    // dWc[co][c2] += sum_p dh0[co][p] a[c2][p],  a = PReLU_p(gamma zhat + beta) rebuilt from zhat in the B fragments
    if (MT_) {
      // known widths: tile id = u * nw + wave = mt * NT2 + n2 with n2 = g * nw + wave (g < G), u = mt * G + g: the G activation fragments of a
      // wave are read and activated ONCE per step and meet every dh0 fragment: G + MT reads for 4 G MT independent MFMAs
      constexpr int G = NT2_ >= nw ? NT2_ / nw : 1;
      static_assert(NT2_ == 0 || NT2_ % nw == 0 || (NT2_ < nw && MT_ == 1), "tile split of the four waves");
      if (NT2_ >= nw || wave < NT2_) {
#pragma unroll
        for (int k0 = 0; k0 < CG_TAIL_PT3; k0 += 16) {
          float afr[G][4];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int n2 = g * nw + wave, c2 = 16 * n2 + l15;
            const bool cok = c2 < C2;
            const float gam = cok ? sK[8 * c2 + 6] : 0.f, bet = cok ? sK[8 * c2 + 7] : 0.f, alp = cok ? (c2 >= C ? alpha_p1 : alpha_p0) : 0.f;
            float bv[4];
            cg_tfrag<0>(cg_tfrag_ptr<0>(sZ + 16 * n2 * CG_TAIL_PS3, CG_TAIL_PS3, l15, slot), CG_TAIL_PS3, k0, bv);
            // positions >= np hold zhat = 0 but a = PReLU(beta) != 0 there; dh0 is zero at those positions, so the product vanishes
#pragma unroll
            for (int s = 0; s < 4; ++s) afr[g][s] = cg_prelu(gam * bv[s] + bet, alp);
          }
#pragma unroll
          for (int mt = 0; mt < (MT_ ? MT_ : 1); ++mt) {
            float av[4];
            cg_tfrag<0>(cg_tfrag_ptr<0>(sDH + 16 * mt * CG_TAIL_PS3, CG_TAIL_PS3, l15, slot), CG_TAIL_PS3, k0, av);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
              for (int s = 0; s < 4; ++s)
                wacc[mt * G + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], afr[g][s], wacc[mt * G + g], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
    for (int u = 0; u < CG_TAIL_MAXW; ++u) {      const int id = u * nw + wave;
      if (id < MT * NT2) {
        const int mt = id / NT2, n2 = id - mt * NT2, c2 = 16 * n2 + l15;
        const int i = c2 >= C ? 1 : 0, c = c2 - i * C;
        const bool cok = c2 < C2;
        const float gam = cok ? sK[8 * c2 + 6] : 0.f, bet = cok ? sK[8 * c2 + 7] : 0.f, alp = cok ? (i ? alpha_p1 : alpha_p0) : 0.f;
        (void)c;
        const float* ap = cg_tfrag_ptr<0>(sDH + 16 * mt * CG_TAIL_PS3, CG_TAIL_PS3, l15, slot);
        const float* bp = cg_tfrag_ptr<0>(sZ + 16 * n2 * CG_TAIL_PS3, CG_TAIL_PS3, l15, slot);
        for (int k0 = 0; k0 < CG_TAIL_PT3; k0 += 16) {
          float av[4], bv[4];
          cg_tfrag<0>(ap, CG_TAIL_PS3, k0, av); cg_tfrag<0>(bp, CG_TAIL_PS3, k0, bv);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            // positions >= np hold zhat = 0 but a = PReLU(beta) != 0 there; dh0 is zero at those positions, so the product vanishes
            const float a = cg_prelu(gam * bv[s] + bet, alp);
            wacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], a, wacc[u], 0, 0, 0);
          }
        }
      }
    }
    }
    CG_TSTAMP();
    // d a[c2][p] = sum_co Wc[co][c2] dh0[co][p];  g_p = d a * PReLU_p'(gamma zhat + beta) -> HBM, sums of g_p and g_p * zhat.
    // Result tiles are position-major (dh0 fragments as the A operand): a lane holds four consecutive positions of ONE channel -
    // zhat comes as one 16-byte LDS read, g_p leaves as one 16-byte store, the channel constants are per lane, and the sums of a
    // channel are three registers per task over all tiles of the workgroup (a wave owns the same c2 tiles in every tile)
    static_assert(CG_TAIL_PT3 == 32, "one pair of position tiles per task: task w = channel tile w");
    cg_f32x4 dacc[CG_TAIL_K3_TASKS][2];
    if (MT_) {
      // known widths: the tasks of a wave (channel tiles wave, wave + 4) share the dh0 fragments and advance through k together
#pragma unroll
      for (int ti = 0; ti < CG_TAIL_K3_TASKS; ++ti) dacc[ti][0] = dacc[ti][1] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
      if (NT2_ >= nw || wave < NT2_) {
#pragma unroll
        for (int k0 = 0; k0 < 16 * (MT_ ? MT_ : 1); k0 += 16) {
          float b0v[4], b1v[4];
          cg_tfrag<1>(cg_tfrag_ptr<1>(sDH, CG_TAIL_PS3, l15, slot), CG_TAIL_PS3, k0, b0v);
          cg_tfrag<1>(cg_tfrag_ptr<1>(sDH + 16, CG_TAIL_PS3, l15, slot), CG_TAIL_PS3, k0, b1v);
#pragma unroll
          for (int ti = 0; ti < CG_TAIL_K3_TASKS; ++ti) {
            if (nw * ti >= NT2_ && ti > 0) break;              // constant per instantiation
            float av[4];
            cg_tfrag<1>(cg_tfrag_ptr<1>(sW + 16 * (wave + nw * ti), WS, l15, slot), WS, k0, av);
#pragma unroll
            for (int s = 0; s < 4; ++s) {                       // C[position 4 * slot + q][channel l15]
              dacc[ti][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0v[s], av[s], dacc[ti][0], 0, 0, 0);
              dacc[ti][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1v[s], av[s], dacc[ti][1], 0, 0, 0);
            }
          }
        }
      }
    }
#pragma unroll
    for (int ti = 0; ti < CG_TAIL_K3_TASKS; ++ti) {
      const int w = wave + nw * ti;
      if (w >= NT2 * (CG_TAIL_PT3 / 32)) break;
      const int mt = w / (CG_TAIL_PT3 / 32), n0 = 32 * (w % (CG_TAIL_PT3 / 32)), n1 = n0 + 16;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      if (MT_) { c0 = dacc[ti][0]; c1 = dacc[ti][1]; }
      else {
        const float* ap = cg_tfrag_ptr<1>(sW + 16 * mt, WS, l15, slot);
        const float* bp0 = cg_tfrag_ptr<1>(sDH + n0, CG_TAIL_PS3, l15, slot);
        const float* bp1 = cg_tfrag_ptr<1>(sDH + n1, CG_TAIL_PS3, l15, slot);
        for (int k0 = 0; k0 < CM; k0 += 16) {
          float av[4], b0v[4], b1v[4];
          cg_tfrag<1>(ap, WS, k0, av); cg_tfrag<1>(bp0, CG_TAIL_PS3, k0, b0v); cg_tfrag<1>(bp1, CG_TAIL_PS3, k0, b1v);
#pragma unroll
          for (int s = 0; s < 4; ++s) {                       // C[position 4 * slot + q][channel l15]
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0v[s], av[s], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1v[s], av[s], c1, 0, 0, 0);
          }
        }
      }
      const int c2 = 16 * mt + l15;
      const bool cok = c2 < C2;
      const int i = c2 >= C ? 1 : 0, c = cok ? c2 - i * C : 0;
      const float gam = cok ? sK[8 * c2 + 6] : 0.f, bet = cok ? sK[8 * c2 + 7] : 0.f, alp = cok ? (i ? alpha_p1 : alpha_p0) : 0.f;
      float* gpr = (i ? gp1 : gp0) + ((long long)b * C + c) * P + p0;       // a select of two hoisted pointers: t.gp[i] with a per-lane i is a
                                                                                 // vector load from the argument block, and waiting for it waits for the whole prefetch
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pq = (h ? n1 : n0) + 4 * slot;
        const cg_f32x4 cc = h ? c1 : c0;
        if (cok && pq < np) {
          const float4 z4 = *reinterpret_cast<const float4*>(sZ + c2 * CG_TAIL_PS3 + pq);
          const float zh[4] = {z4.x, z4.y, z4.z, z4.w};
          float gq[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v = gam * zh[q] + bet, da = cc[q];
            const bool in = pq + q < np;
            gq[q] = v > 0.f ? da : alp * da;
            if (in) {
              racc[ti][0] += gq[q]; racc[ti][1] += gq[q] * zh[q];
              if (!(v > 0.f)) racc[ti][2] += da * v;
            }
          }
          if (vec) *reinterpret_cast<float4*>(gpr + pq) = make_float4(gq[0], gq[1], gq[2], gq[3]);      // np % 4 == 0: the quad is inside the row
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (pq + q < np) gpr[pq + q] = gq[q];
          }
        }
      }
    }
  }
  CG_TSTAMP();
  // the four lanes l15 + 16 * slot hold the partial sums of one channel: two exchanges, then one LDS atomic per channel and task
#pragma unroll
  for (int ti = 0; ti < CG_TAIL_K3_TASKS; ++ti) {
    const int w = wave + nw * ti;
    if (w >= NT2 * (CG_TAIL_PT3 / 32)) break;
    const int c2 = 16 * (w / (CG_TAIL_PT3 / 32)) + l15;
    float s1 = racc[ti][0], s2 = racc[ti][1], sa = racc[ti][2];
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64); sa += __shfl_xor(sa, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); sa += __shfl_xor(sa, 32, 64);
    if (slot == 0 && c2 < C2) {
      atomicAdd(&sRed[2 * c2], (double)s1); atomicAdd(&sRed[2 * c2 + 1], (double)s2); atomicAdd(&sRed[2 * C2M + (c2 >= C ? 1 : 0)], (double)sa);
    }
  }
  __syncthreads();
  for (int e = tid; e < 2 * C2; e += CG_TAIL_THREADS) {
    const int c2 = e >> 1, i = c2 >= C ? 1 : 0, c = c2 - i * C;
    atomicAdd(&t.red_p[i][2 * c + (e & 1)], sRed[e]);
  }
  if (tid < 2) atomicAdd(&t.red_p[tid][2 * C + ((int)blockIdx.x & (CG_ALPHA_SLOTS - 1))], sRed[2 * C2M + tid]);
  float* dW = t.dWc_ws + (long long)(blockIdx.x % replicas) * C * C2;
#pragma unroll
  for (int u = 0; u < CG_TAIL_MAXW; ++u) {
    const int id = u * nw + wave;
    if (id < MT * NT2) {
      const int mt = id / NT2, n2 = id - mt * NT2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = 16 * mt + 4 * slot + q, c2 = 16 * n2 + l15;
        if (co < C && c2 < C2) atomicAdd(&dW[co * C2 + c2], wacc[u][q]);
      }
    }
  }
  CG_TSTAMP();
  CG_TSTAMP_END();
}

// K4: dz = BN_p'(g_p); dw = sum dz * x; g_d = w * dz * PReLU_d'(u) -> dr (HBM); sums of g_t = g_d * keep for the tcn BatchNorm.
// One WAVE per (b, c) row (the gate gradient is a row sum: shuffles, no workgroup barrier); a workgroup = four waves walking
// the batch rows of one channel, channel sums kept in registers and added once per wave.
__global__ void cg_tail_k4_kernel(CgDstdTail t, int rb) {
  const int i = blockIdx.z, c = blockIdx.x, b0 = blockIdx.y * rb, P = t.T * t.V;
  if (b0 >= t.B) return;
  const int nb = min(rb, t.B - b0), lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const double cnt = (double)t.B * P;
  const CgAff at = cg_tail_aff(t.bn_t[i], c, t.C, 0.0, t.train, true, false);
  const CgAff ap = cg_tail_aff(t.bn_p[i], c, t.C, 0.0, t.train, true, false);
  const float ad = t.alpha_d[i][0], scale_p = ap.gamma * ap.rstd;
  const unsigned long long seed = (t.train && t.drop_p > 0.f) ? *t.seed : 0ull;
  float m1 = 0.f, m2 = 0.f;
  if (t.train) { m1 = (float)(t.red_p[i][2 * c] / cnt); m2 = (float)(t.red_p[i][2 * c + 1] / cnt); }
  double s1 = 0.0, s2 = 0.0, sa = 0.0;
  for (int br = wave; br < nb; br += nw) {
    const int b = b0 + br;
    const float wv = t.w[i][(long long)b * t.C + c];
    const long long base = ((long long)b * t.C + c) * P;
    float sw = 0.f;
    if ((P & 3) == 0) {
      for (int p = 4 * lane; p < P; p += 256) {
        float keep[4];
        cg_tail_keep4(t, i, seed, (unsigned long long)(base + p), keep);
        const float4 y4 = *reinterpret_cast<const float4*>(t.y[i] + base + p), r4 = *reinterpret_cast<const float4*>(t.r[i] + base + p);
        const float4 g4 = *reinterpret_cast<const float4*>(t.gp[i] + base + p);
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float gd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float u;
          const float x = cg_tail_x(at, ad, yv[j], rv[j], keep[j], u);
          const float z = wv * x;
          const float dz = t.train ? scale_p * (gv[j] - m1 - (z - ap.mean) * ap.rstd * m2) : gv[j] * scale_p;
          sw += dz * x;
          const float dx = wv * dz;
          gd[j] = u > 0.f ? dx : ad * dx;
          if (!(u > 0.f)) sa += (double)dx * (double)u;
          const float gt = gd[j] * keep[j];
          s1 += (double)gt; s2 += (double)gt * (double)((yv[j] - at.mean) * at.rstd);
        }
        *reinterpret_cast<float4*>(t.dr[i] + base + p) = make_float4(gd[0], gd[1], gd[2], gd[3]);
      }
    } else {
      for (int p = lane; p < P; p += 64) {
        const float keep = cg_tail_keep(t, i, seed, b, c, p);
        float u;
        const float yv = t.y[i][base + p];
        const float x = cg_tail_x(at, ad, yv, t.r[i][base + p], keep, u);
        const float z = wv * x;
        const float g = t.gp[i][base + p];
        const float dz = t.train ? scale_p * (g - m1 - (z - ap.mean) * ap.rstd * m2) : g * scale_p;
        sw += dz * x;
        const float dx = wv * dz;
        const float gd = u > 0.f ? dx : ad * dx;
        t.dr[i][base + p] = gd;
        if (!(u > 0.f)) sa += (double)dx * (double)u;
        const float gt = gd * keep;
        s1 += (double)gt; s2 += (double)gt * (double)((yv - at.mean) * at.rstd);
      }
    }
    double swd = (double)sw;
    swd = cg_wave_sum(swd);
    if (lane == 0) t.dw[i][(long long)b * t.C + c] = (float)swd;
  }
  s1 = cg_wave_sum(s1); s2 = cg_wave_sum(s2); sa = cg_wave_sum(sa);
  if (lane == 0) {
    atomicAdd(&t.red_t[i][2 * c], s1); atomicAdd(&t.red_t[i][2 * c + 1], s2);
    atomicAdd(&t.red_t[i][2 * t.C + ((c + 5 * (int)blockIdx.y + 17 * wave) & (CG_ALPHA_SLOTS - 1))], sa);
  }
}

// K5: dy = BN_t'(g_d * keep); also the per-channel parameter gradients of all three BatchNorm levels and the PReLU slopes
__global__ void cg_tail_k5_kernel(CgDstdTail t, int rb) {
  const int i = blockIdx.z, c = blockIdx.x, b0 = blockIdx.y * rb;
  if (b0 >= t.B) return;
  const int P = t.T * t.V, nb = min(rb, t.B - b0), C = t.C;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const double cnt = (double)t.B * P;
  const CgAff at = cg_tail_aff(t.bn_t[i], c, C, 0.0, t.train, true, false);
  const unsigned long long seed = (t.train && t.drop_p > 0.f) ? *t.seed : 0ull;
  const bool train = t.train != 0;
  float m1 = 0.f, m2 = 0.f;
  if (train) { m1 = (float)(t.red_t[i][2 * c] / cnt); m2 = (float)(t.red_t[i][2 * c + 1] / cnt); }
  const float scale = at.gamma * at.rstd, rm2 = at.rstd * m2;
  const float* __restrict__ dr = t.dr[i]; const float* __restrict__ y = t.y[i]; float* __restrict__ dy = t.dy[i];
  for (int br = wave; br < nb; br += nw) {                 // a wave per row, 16-byte accesses, one dropout hash per quad (as K4)
    const int b = b0 + br;
    const long long base = ((long long)b * C + c) * P;
    if ((P & 3) == 0) {
      for (int p = 4 * lane; p < P; p += 256) {
        float keep[4];
        cg_tail_keep4(t, i, seed, (unsigned long long)(base + p), keep);
        const float4 g4 = *reinterpret_cast<const float4*>(dr + base + p);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float o[4];
        if (train) {
          const float4 y4 = *reinterpret_cast<const float4*>(y + base + p);
          const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = scale * (gv[j] * keep[j] - m1 - (yv[j] - at.mean) * rm2);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = gv[j] * keep[j] * scale;
        }
        *reinterpret_cast<float4*>(dy + base + p) = make_float4(o[0], o[1], o[2], o[3]);
      }
    } else {
      for (int p = lane; p < P; p += 64) {
        const float gt = dr[base + p] * cg_tail_keep(t, i, seed, b, c, p);
        dy[base + p] = train ? scale * (gt - m1 - (y[base + p] - at.mean) * rm2) : gt * scale;
      }
    }
  }
  if (blockIdx.y == 0 && threadIdx.x == 0) {
    t.dgamma_t[i][c] = (float)t.red_t[i][2 * c + 1]; t.dbeta_t[i][c] = (float)t.red_t[i][2 * c];
    t.dgamma_p[i][c] = (float)t.red_p[i][2 * c + 1]; t.dbeta_p[i][c] = (float)t.red_p[i][2 * c];
    if (i == 0) { t.dgamma_c[c] = (float)t.red_c[2 * c + 1]; t.dbeta_c[c] = (float)t.red_c[2 * c]; }
    if (c == 0) {
      t.dalpha_d[i][0] = (float)cg_alpha_sum(t.red_t[i] + 2 * t.C); t.dalpha_p[i][0] = (float)cg_alpha_sum(t.red_p[i] + 2 * t.C);
      if (i == 0) t.dalpha_c[0] = (float)cg_alpha_sum(t.red_c + 2 * t.C);
    }
  }
}

__global__ void cg_tail_fold_kernel(const float* __restrict__ ws, int replicas, int n, float* __restrict__ dW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int r = 0; r < replicas; ++r) s += ws[(long long)r * n + i];
  dW[i] = s;
}

// ---- host side ---------------------------------------------------------------------------------------------------
#define CG_TAIL_REPLICAS 16

static int cg_tail_check(const CgDstdTail* t) {
  if (!t) return CG_EARG;
  if (t->B <= 0 || t->C <= 0 || t->T <= 0 || t->V <= 0 || t->C > 64) return CG_ESHAPE;
  for (int i = 0; i < 2; ++i)
    if (!t->y[i] || !t->r[i] || !t->w[i] || !t->alpha_d[i] || !t->alpha_p[i] || !t->bn_t[i].gamma || !t->bn_p[i].gamma || !t->bn_t[i].save ||
        !t->bn_p[i].save) return CG_EARG;
  if (!t->Wc || !t->bn_c.gamma || !t->bn_c.save || !t->alpha_c || !t->h0) return CG_EARG;
  if (t->train && t->drop_p > 0.f && !t->seed) return CG_EARG;
  if (t->B > 65535) return CG_ESHAPE;
  return CG_OK;
}

extern "C" long long cg_dstd_tail_ws_floats(int C) { return (long long)CG_TAIL_REPLICAS * C * 2 * C; }

static size_t cg_tail_gemm_lds(int C, bool bwd) {
  const int CM = (C + 15) & ~15, C2M = (2 * C + 15) & ~15, WS = C2M + 4;
  size_t f = (size_t)C2M * (bwd ? CG_TAIL_PS3 : CG_TAIL_PS) + (size_t)CM * WS + (size_t)8 * C2M;
  if (bwd) f += (size_t)CM * CG_TAIL_PS3 + (size_t)8 * CM;
  return f * sizeof(float) + (bwd ? (size_t)(2 * C2M + 2) : (size_t)2 * CM) * sizeof(double) + 16;
}

// include/cistgcn_hip.h : cg_dstd_tail_fwd (phases 1..4) / cg_dstd_tail_bwd (phases 1..5)
extern "C" int cg_dstd_tail_fwd(const CgDstdTail* t, int phase, void* stream_) {
  int st = cg_tail_check(t);
  if (st != CG_OK) return st;
  hipStream_t stream = (hipStream_t)stream_;
  const int P = t->T * t->V;
  const int rb = cg_tail_rows(t->B, t->C, P);
  const dim3 rows((unsigned)t->C, (unsigned)((t->B + rb - 1) / rb), 1);
  if (phase == 1) {
    if (!t->train) return CG_OK;                      // eval: running statistics, nothing to reduce
    if (!t->bn_t[0].stats || !t->bn_t[1].stats || !t->bn_p[0].stats || !t->bn_p[1].stats) return CG_EARG;
    hipLaunchKernelGGL(cg_tail_f1_kernel, dim3(rows.x, rows.y, 2), dim3(256), 0, stream, *t, rb);
  } else if (phase == 2) {
    if (t->train && (!t->bn_p[0].stats || !t->bn_p[1].stats || !t->bn_c.stats)) return CG_EARG;
    const int tps = (P + CG_TAIL_PT - 1) / CG_TAIL_PT, total = t->B * tps;
    const int per = (total + 511) / 512, nwg = (total + per - 1) / per;
    const size_t lds = cg_tail_gemm_lds(t->C, false);
    const int mt = (t->C + 15) / 16, nt2 = (2 * t->C + 15) / 16;
#define CG_TAIL_F2_LAUNCH(M, N)                                                                                               \
    {                                                                                                                        \
      hipError_t e = cg_lds_limit((const void*)cg_tail_f2_kernel<M, N>, lds);                                                \
      if (e != hipSuccess) return (int)e;                                                                                    \
      hipLaunchKernelGGL((cg_tail_f2_kernel<M, N>), dim3((unsigned)nwg), dim3(CG_TAIL_THREADS), lds, stream, *t, tps, total, per); \
    }
    if (mt == 4 && nt2 == 8) CG_TAIL_F2_LAUNCH(4, 8)           // C = 57 .. 64
    else if (mt == 2 && nt2 == 4) CG_TAIL_F2_LAUNCH(2, 4)      // C = 25 .. 32
    else if (mt == 1 && nt2 == 2) CG_TAIL_F2_LAUNCH(1, 2)      // C = 9 .. 16
    else if (mt == 1 && nt2 == 1) CG_TAIL_F2_LAUNCH(1, 1)      // C <= 8
    else CG_TAIL_F2_LAUNCH(0, 0)
#undef CG_TAIL_F2_LAUNCH
  } else if (phase == 3) {
    if (!t->pooled || (t->train && !t->bn_c.stats)) return CG_EARG;
    hipLaunchKernelGGL(cg_tail_f3_kernel, rows, dim3(256), 0, stream, *t, rb);
  } else if (phase == 4) {
    if (!t->gate || !t->bres || !t->out) return CG_EARG;
    hipLaunchKernelGGL(cg_tail_f4_kernel, rows, dim3(256), 0, stream, *t, rb);
  } else return CG_EARG;
  return cg_launch_status();
}

extern "C" int cg_dstd_tail_bwd(const CgDstdTail* t, int phase, void* stream_) {
  int st = cg_tail_check(t);
  if (st != CG_OK) return st;
  if (!t->dout || !t->gate) return CG_EARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int P = t->T * t->V, C = t->C;
  const int rb = cg_tail_rows(t->B, C, P);
  const dim3 rows((unsigned)C, (unsigned)((t->B + rb - 1) / rb), 1);
  if (phase == 1) {
    if (!t->dgate) return CG_EARG;
    hipLaunchKernelGGL(cg_tail_k1_kernel, rows, dim3(256), 0, stream, *t, rb);
  } else if (phase == 2) {
    if (!t->dpooled || !t->red_c) return CG_EARG;
    hipLaunchKernelGGL(cg_tail_k2_kernel, rows, dim3(256), 0, stream, *t, rb);
  } else if (phase == 3) {
    if (!t->dpooled || !t->red_c || !t->gp[0] || !t->gp[1] || !t->red_p[0] || !t->red_p[1] || !t->dWc_ws || !t->dWc) return CG_EARG;
    const int tps = (P + CG_TAIL_PT3 - 1) / CG_TAIL_PT3, total = t->B * tps;
    const int per = (total + 511) / 512, nwg = (total + per - 1) / per;
    const size_t lds = cg_tail_gemm_lds(C, true);
    const int mt = (C + 15) / 16, nt2 = (2 * C + 15) / 16;
#define CG_TAIL_K3_LAUNCH(M, N)                                                                                               \
    {                                                                                                                        \
      hipError_t e = cg_lds_limit((const void*)cg_tail_k3_kernel<M, N>, lds);                                                \
      if (e != hipSuccess) return (int)e;                                                                                    \
      hipLaunchKernelGGL((cg_tail_k3_kernel<M, N>), dim3((unsigned)nwg), dim3(CG_TAIL_THREADS), lds, stream, *t, tps, total, per, CG_TAIL_REPLICAS); \
    }
    if (mt == 4 && nt2 == 8) CG_TAIL_K3_LAUNCH(4, 8)
    else if (mt == 2 && nt2 == 4) CG_TAIL_K3_LAUNCH(2, 4)
    else if (mt == 1 && nt2 == 2) CG_TAIL_K3_LAUNCH(1, 2)
    else if (mt == 1 && nt2 == 1) CG_TAIL_K3_LAUNCH(1, 1)
    else CG_TAIL_K3_LAUNCH(0, 0)
#undef CG_TAIL_K3_LAUNCH
    st = cg_launch_status();
    if (st != CG_OK) return st;
    const int n = C * 2 * C;
    hipLaunchKernelGGL(cg_tail_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, t->dWc_ws, CG_TAIL_REPLICAS, n, t->dWc);
  } else if (phase == 4) {
    if (!t->gp[0] || !t->gp[1] || !t->dr[0] || !t->dr[1] || !t->dw[0] || !t->dw[1] || !t->red_t[0] || !t->red_t[1] || !t->red_p[0]) return CG_EARG;
    hipLaunchKernelGGL(cg_tail_k4_kernel, dim3(rows.x, rows.y, 2), dim3(256), 0, stream, *t, rb);
  } else if (phase == 5) {
    if (!t->dy[0] || !t->dy[1] || !t->dr[0] || !t->dr[1] || !t->red_t[0] || !t->red_t[1] || !t->dgamma_t[0] || !t->dgamma_p[0] || !t->dgamma_c) return CG_EARG;
    hipLaunchKernelGGL(cg_tail_k5_kernel, dim3(rows.x, rows.y, 2), dim3(256), 0, stream, *t, rb);
  } else return CG_EARG;
  return cg_launch_status();
}
