// The interpretability attention map behind its two towers (reference: Map2Adj.forward, CISTGCN.py:183-189 with the `expansor`
// of :165-170): rank-1 seed of the joint / time summaries -> 1x1 conv over the slab axis -> BatchNorm -> Dropout -> PReLU ->
// 1x1 conv = the learned adjacency `Adj`, as phase kernels cut only at the BatchNorm barrier.  The seed
//   space: o[b,v,t,u] = s[b,v,t] q[b,u,v]  (B,V,T,T)      time: o[b,t,v,w] = s[b,v,t] q[b,t,w]  (B,T,V,V)
// is never written: it is a product of two small per-sample tables (S[k][a] Q[k][b'], k = slab, position p = (a, b')), so the
// first conv generates its operand in registers and the backward reduces straight into ds / dq.
//   forward   M1  e = W0 (S x Q) + channel sums of e            (writes e)
//             M2  Adj = W4 PReLU(Dropout(BN(e)))                  (reads e, writes Adj)
//   backward  N1  dh = W4^T dAdj; g = dh PReLU' keep -> HBM, sums of g and g e_hat, d alpha; dW4 += dAdj h^T
//             N2  de = BN'(g); do = W0^T de; dS += do Q, dQ += do S (per-sample LDS accumulators); dW0 += de (S x Q)^T
// One 256-thread workgroup per (sample, tower) walks the 64-position tiles of its slab stack; all products on
// v_mfma_f32_16x16x4_f32 with the weights as A operands from LDS.  Both towers of a block (space and time) share a launch.
#include "cg_common.h"
#include "cg_phase.h"
#include "map2adj_tail.h"
#include <cstdlib>

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

// tile of a workgroup: [KcM channels][PT positions] with KcM * PT = 4096 (16 floats per thread and staged tensor):
// PT = 64 for 33..64 slabs (KcM = 64), 128 for 17..32, 256 up to 16
#define CG_ADJ_THREADS 256
#define CG_ADJ_REPLICAS 32

#define CG_ADJ_TPW_MAX 8                    // 64-position tiles per workgroup

// Tile geometry of a padded slab count KcM (16 / 32 / 64): compile-time in the kernels (one instantiation each), so that every matrix-core
// loop has a constant trip count and constant LDS strides - with run-time geometry the compiler left the k loops rolled: two LDS reads,
// a full wait, four dependent MFMAs per iteration (round 4).  The host's CgAdjGeom carries the same numbers for the launch arithmetic.
template <int KCM> struct CgAdjK {
  static constexpr int KcM = KCM, WS = KCM + 4, PT = 4096 / KCM, PS = PT + 4, NP = PT / 32, MT = KCM / 16;
  static constexpr int lgq = KCM > 32 ? 4 : KCM > 16 ? 5 : 6;          // log2(PT / 4)
};

// geometry of a tower (host side, cg_adj_geometry): handed to the kernels next to the argument block
__device__ __forceinline__ const CgAdjGeom& cg_adj_geom(const CgAdjTailPair& pr, int i) { return pr.g[i]; }
__device__ __forceinline__ unsigned cg_adj_div(unsigned n, unsigned magic) { return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n; }

// Staging loops: the global loads of four strides are issued together, then their LDS stores.
// S[k][a], Q[k][b'] of sample b: [KcM][JS] tables, rows >= Kc zero
template <int KCM>
__device__ __forceinline__ void cg_adj_tables(const CgAdjTail& t, const CgAdjGeom& g, int b, float* sS, float* sQ) {
  using K = CgAdjK<KCM>;
  const int V = t.domain == 0 ? t.Kc : t.J, T = t.domain == 0 ? t.J : t.Kc, KJ = t.Kc * t.J;
  const float* sb = t.s + (long long)b * V * T;      // (V, T)
  const float* qb = t.q + (long long)b * T * V;      // (T, V)
  for (int i0 = threadIdx.x; i0 < KJ; i0 += 4 * CG_ADJ_THREADS) {
    float sv[4], qv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + CG_ADJ_THREADS * j, k = (int)cg_adj_div((unsigned)i, g.magicJ), a = i - k * t.J;
      sv[j] = qv[j] = 0.f;
      if (i < KJ) {
        if (t.domain == 0) { sv[j] = sb[k * T + a]; qv[j] = qb[a * V + k]; }       // k = joint: S = s[v][t], Q = q[tau][v]
        else { sv[j] = sb[a * T + k]; qv[j] = qb[k * V + a]; }                     // k = frame: S = s[v][t], Q = q[t][w]
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + CG_ADJ_THREADS * j, k = (int)cg_adj_div((unsigned)i, g.magicJ), a = i - k * t.J;
      if (i < KJ) { sS[k * g.JS + a] = sv[j]; sQ[k * g.JS + a] = qv[j]; }
    }
  }
  for (int e = t.Kc * g.JS + threadIdx.x; e < K::KcM * g.JS; e += CG_ADJ_THREADS) { sS[e] = 0.f; sQ[e] = 0.f; }
  for (int k = threadIdx.x; k < t.Kc; k += CG_ADJ_THREADS) { sS[k * g.JS + t.J] = 0.f; sQ[k * g.JS + t.J] = 0.f; }     // the pad column: read (and discarded) by clamped indices
}

// W (Kc x Kc) -> sW [KcM][WS], padding zero
template <int KCM>
__device__ __forceinline__ void cg_adj_weight(const float* __restrict__ W, const CgAdjTail& t, const CgAdjGeom& g, float* sW) {
  using K = CgAdjK<KCM>;
  const int n = t.Kc * t.Kc, padw = K::WS - t.Kc;
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * CG_ADJ_THREADS) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int i = i0 + CG_ADJ_THREADS * j; v[j] = i < n ? W[i] : 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + CG_ADJ_THREADS * j, r = (int)cg_adj_div((unsigned)i, g.magicKc);
      if (i < n) sW[r * K::WS + i - r * t.Kc] = v[j];
    }
  }
  for (int i = threadIdx.x; i < t.Kc * padw; i += CG_ADJ_THREADS) {            // columns Kc .. WS - 1 of the rows < Kc
    const int r = (int)cg_adj_div((unsigned)i, g.magicPad);
    sW[r * K::WS + t.Kc + i - r * padw] = 0.f;
  }
  for (int e = t.Kc * K::WS + threadIdx.x; e < K::KcM * K::WS; e += CG_ADJ_THREADS) sW[e] = 0.f;
}

// rows Kc .. KcM - 1 of `count` consecutive [KcM][PS] tile images: never written by the staging, read by the fragments
template <int KCM>
__device__ __forceinline__ void cg_adj_zero_pad_rows(const CgAdjTail& t, const CgAdjGeom& g, float* img, int count) {
  using K = CgAdjK<KCM>;
  const int n = (K::KcM - t.Kc) * K::PS;
  for (int c = 0; c < count; ++c)
    for (int e = threadIdx.x; e < n; e += CG_ADJ_THREADS) img[(c * K::KcM + t.Kc) * K::PS + e] = 0.f;
}

// generated fragment of the seed for 16-wide k chunk k0: v[s] = S[k][a] Q[k][b'], k = k0 + 4*slot + s
__device__ __forceinline__ void cg_adj_seed_frag_k(const float* sS, const float* sQ, int JS, int k0, int slot, int a, int bp, bool ok, float v[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = k0 + 4 * slot + s;
    v[s] = ok ? sS[k * JS + a] * sQ[k * JS + bp] : 0.f;
  }
}

// ======================================================================================================================
// M1: e[u][p] = sum_k W0[u][k] S[k][a] Q[k][b']   + channel sums of e
// ======================================================================================================================
template <int KCM>
__device__ __forceinline__ void cg_adj_m1_body(const CgAdjTail& t, const CgAdjGeom& g, int b, int tile0, int tile1) {
  using K = CgAdjK<KCM>;
  float* sS = reinterpret_cast<float*>(cg_dyn_lds);
  float* sQ = sS + K::KcM * g.JS;
  float* sW = sQ + K::KcM * g.JS;
  double* sStat = reinterpret_cast<double*>(sW + K::KcM * K::WS + ((K::KcM * g.JS * 2 + K::KcM * K::WS) & 1));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  cg_adj_tables<KCM>(t, g, b, sS, sQ);
  cg_adj_weight<KCM>(t.W0, t, g, sW);
  for (int e = tid; e < 2 * K::KcM; e += CG_ADJ_THREADS) sStat[e] = 0.0;
  __syncthreads();
  constexpr int MT = K::MT;
  float* eb = t.e + (long long)b * t.Kc * g.Pn;
  const int P0 = tile0 * K::PT, ngroups = (min(g.Pn, tile1 * K::PT) - P0 + 31) / 32;
  // C[position][channel] tiles: a lane ends up with four consecutive positions of ONE channel (float4 stores, per-lane channel
  // sums); a wave keeps its channel tile (4 % MT == 0), so the sums stay in registers over all its groups
  const bool vec = (g.Pn & 3) == 0;
  const int mt = wave % MT, u = 16 * mt + l15;
  const bool uok = u < t.Kc;
  const float* ap = cg_tfrag_ptr<0>(sW + 16 * mt * K::WS, K::WS, l15, slot);
  float s1 = 0.f, s2 = 0.f;
  for (int w = wave; w < ngroups * MT; w += CG_ADJ_THREADS / 64) {       // (group of 32 positions, u tile)
    const int grp = w / MT, p0 = P0 + 32 * grp;
    const int pa = p0 + l15, pb = p0 + 16 + l15;
    const int aa = (int)cg_adj_div((unsigned)pa, g.magicJ), ab = (int)cg_adj_div((unsigned)pb, g.magicJ);
    const bool oka = pa < g.Pn, okb = pb < g.Pn;
    cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    for (int k0 = 0; k0 < K::KcM; k0 += 16) {
      float av[4], b0v[4], b1v[4];
      cg_tfrag<0>(ap, K::WS, k0, av);
      // (positions beyond the tensor read the tables at (0, 0): their result rows are never stored, so nothing is masked - a condition
      // here turns into a branch and a wait around every one of the LDS reads)
      cg_adj_seed_frag_k(sS, sQ, g.JS, k0, slot, oka ? aa : 0, oka ? pa - aa * t.J : 0, true, b0v);
      cg_adj_seed_frag_k(sS, sQ, g.JS, k0, slot, okb ? ab : 0, okb ? pb - ab * t.J : 0, true, b1v);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0v[s], av[s], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1v[s], av[s], c1, 0, 0, 0);
      }
    }
    if (uok) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pq = p0 + 16 * h + 4 * slot;               // first of this lane's four positions
        const cg_f32x4 c = h ? c1 : c0;
        float* dst = eb + (long long)u * g.Pn + pq;
        if (vec) {
          if (pq < g.Pn) {
            *reinterpret_cast<float4*>(dst) = make_float4(c[0], c[1], c[2], c[3]);
            s1 += (c[0] + c[1]) + (c[2] + c[3]); s2 += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (pq + q < g.Pn) { dst[q] = c[q]; s1 += c[q]; s2 += c[q] * c[q]; }
        }
      }
    }
  }
  if (t.train && uok) { atomicAdd(&sStat[2 * u], (double)s1); atomicAdd(&sStat[2 * u + 1], (double)s2); }
  if (t.train) {
    __syncthreads();
    double* rep = t.bn.stats + (long long)(blockIdx.x % CG_STAT_REPLICAS) * 2 * t.Kc;
    for (int e = tid; e < 2 * t.Kc; e += CG_ADJ_THREADS) atomicAdd(&rep[e], sStat[e]);
  }
}

// the instantiation of a tower's padded slab count (wave-uniform: one tower per blockIdx.y)
#define CG_ADJ_DISPATCH(KcM_, CALL)                  \
  switch (KcM_) {                                     \
    case 16: { constexpr int KCM = 16; CALL; } break; \
    case 32: { constexpr int KCM = 32; CALL; } break; \
    default: { constexpr int KCM = 64; CALL; } break; \
  }

__global__ __launch_bounds__(CG_ADJ_THREADS) void cg_adj_m1_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom& g = pr.g[blockIdx.y];
  const int b = blockIdx.x / pr.nch_max, ch = blockIdx.x - b * pr.nch_max;
  if (ch >= g.nch) return;
  const int tile0 = ch * g.tpw, tile1 = min(g.ntiles, tile0 + g.tpw);
  CG_ADJ_DISPATCH(g.KcM, cg_adj_m1_body<KCM>(t, g, b, tile0, tile1))
}

// `red` ([CG_ADJ_REPLICAS][2 Kc + 1] f64: sums of g and g e_hat per channel, d alpha) summed over its replicas
__device__ __forceinline__ double cg_adj_red(const CgAdjTail& t, int i) {
  double s = 0.0;
  for (int r = 0; r < CG_ADJ_REPLICAS; ++r) s += t.red[(long long)r * (2 * t.Kc + 1) + i];
  return s;
}

// per-channel constants [KcM][8]: mean, rstd, scale = gamma * rstd, beta, m1, m2 (backward), gamma, scale * rstd * m2
__device__ __forceinline__ void cg_adj_consts(const CgAdjTail& t, const CgAdjGeom& g, float* sK, bool backward, bool owner) {
  const double cnt = (double)t.B * g.Pn;
  for (int c = threadIdx.x; c < t.Kc; c += CG_ADJ_THREADS) {
    const CgAff a = cg_tail_aff(t.bn, c, t.Kc, cnt, t.train, backward, owner);
    float* k = sK + 8 * c;
    k[0] = a.mean; k[1] = a.rstd; k[2] = a.gamma * a.rstd; k[3] = a.beta;
    k[4] = (backward && t.train) ? (float)(cg_adj_red(t, 2 * c) / cnt) : 0.f;
    k[5] = (backward && t.train) ? (float)(cg_adj_red(t, 2 * c + 1) / cnt) : 0.f;
    k[6] = a.gamma;
    k[7] = k[2] * k[1] * k[5];                    // N2: de = scale (g - m1) - (e - mean) * [scale rstd m2]
  }
}

__device__ __forceinline__ float cg_adj_keep(const CgAdjTail& t, unsigned long long seed, long long idx) {
  if (!(t.train && t.drop_p > 0.f)) return 1.f;
  return cg_drop_scale(t.drop_p, seed, t.salt, (unsigned long long)idx);
}

// ---- pipelined staging: a thread's share of one [Kc][64-position] tile travels global -> registers (issued a tile ahead, in flight
// during the matrix work) -> transform -> LDS.  VEC (J*J % 4 == 0): float4 number r of the thread is element 4 * (tid + 256 r) of the
// tile image; otherwise scalar number j is element tid + 256 j.
template <int KCM, bool VEC>
__device__ __forceinline__ void cg_adj_fetch(const float* __restrict__ src, int Kc, int Pn, int p0, int np, float buf[16]) {
  using K = CgAdjK<KCM>;
  if (VEC) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = (int)threadIdx.x + CG_ADJ_THREADS * r, c = e >> K::lgq, pp = 4 * (e & ((1 << K::lgq) - 1));
      // no branch around the load (element 0 of the sample stands in for what lies outside): the loads of a thread issue back to back
      const bool in = c < Kc && pp < np;
      const float4 v = *reinterpret_cast<const float4*>(src + (in ? (long long)c * Pn + p0 + pp : 0ll));
      buf[4 * r] = in ? v.x : 0.f; buf[4 * r + 1] = in ? v.y : 0.f; buf[4 * r + 2] = in ? v.z : 0.f; buf[4 * r + 3] = in ? v.w : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = (int)threadIdx.x + CG_ADJ_THREADS * j, c = e >> (K::lgq + 2), pp = e & (K::PT - 1);
      buf[j] = (c < Kc && pp < np) ? src[(long long)c * Pn + p0 + pp] : 0.f;
    }
  }
}
// fn(c, pp, ok, off): channel c < Kc, tile position pp of staging register number `off` (4 consecutive positions with VEC, else 1);
// ok: inside the tensor
template <int KCM, bool VEC, typename F>
__device__ __forceinline__ void cg_adj_commit(int Kc, int np, F fn) {
  using K = CgAdjK<KCM>;
  if (VEC) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = (int)threadIdx.x + CG_ADJ_THREADS * r, c = e >> K::lgq, pp = 4 * (e & ((1 << K::lgq) - 1));
      if (c < Kc) fn(c, pp, pp < np, 4 * r);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = (int)threadIdx.x + CG_ADJ_THREADS * j, c = e >> (K::lgq + 2), pp = e & (K::PT - 1);
      if (c < Kc) fn(c, pp, pp < np, j);
    }
  }
}
template <bool VEC>
__device__ __forceinline__ void cg_adj_keeps(float drop_p, unsigned int salt, bool drop, unsigned long long seed, long long idx, float keep[4]) {
  if (VEC) cg_keep4(drop, drop_p, seed, salt, (unsigned long long)idx, keep);
  else keep[0] = drop ? cg_drop_scale(drop_p, seed, salt, (unsigned long long)idx) : 1.f;
}
template <int KCM, bool VEC>
__device__ __forceinline__ void cg_adj_put(float* img, int c, int pp, const float v[4]) {
  using K = CgAdjK<KCM>;
  if (VEC) *reinterpret_cast<float4*>(img + c * K::PS + pp) = make_float4(v[0], v[1], v[2], v[3]);
  else img[c * K::PS + pp] = v[0];
}

// ======================================================================================================================
// M2: Adj[u'][p] = sum_u W4[u'][u] h[u][p],  h = PReLU(Dropout(BN(e)))
// ======================================================================================================================
template <int KCM, bool VEC>
__device__ __forceinline__ void cg_adj_m2_body(const CgAdjTail& t, const CgAdjGeom& g, int b, int tile0, int tile1, bool owner) {
  using K = CgAdjK<KCM>;
  float* sH = reinterpret_cast<float*>(cg_dyn_lds);              // [KcM][PS]
  float* sW = sH + K::KcM * K::PS;                             // [KcM][WS]
  float* sK = sW + K::KcM * K::WS;                                  // [KcM][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  constexpr int N = VEC ? 4 : 1;
  // read once what the tile loop needs of the argument blocks (kernel-argument memory behind a run-time index)
  const int Kc = t.Kc, Pn = g.Pn;
  const bool train = t.train != 0;
  (void)train;
  const float* eb = t.e + (long long)b * Kc * Pn;
  float ebuf[16];
  cg_adj_fetch<KCM, VEC>(eb, Kc, Pn, tile0 * K::PT, min(K::PT, Pn - tile0 * K::PT), ebuf);
  cg_adj_zero_pad_rows<KCM>(t, g, sH, 1);
  cg_adj_weight<KCM>(t.W4, t, g, sW);
  cg_adj_consts(t, g, sK, false, owner);
  const float drop_p = t.drop_p; const unsigned int salt = t.salt;
  const bool drop = train && drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  const float alpha = t.alpha[0];
  constexpr int MT = K::MT;
  float* ab = t.adj + (long long)b * Kc * Pn;
  float* const tap = t.tap;
  for (int tile = tile0; tile < tile1; ++tile) {
    const int p0 = tile * K::PT, np = min(K::PT, Pn - p0);
    __syncthreads();
    cg_adj_commit<KCM, VEC>(Kc, np, [&](int c, int pp, bool ok, int off) {
      const float4 k4 = *reinterpret_cast<const float4*>(sK + 8 * c);           // mean, rstd, scale, beta
      const long long idx = ((long long)b * Kc + c) * Pn + p0 + pp;
      float keep[4], v[4];
      cg_adj_keeps<VEC>(drop_p, salt, drop && ok, seed, idx, keep);
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = ok ? cg_prelu(((ebuf[off + j] - k4.x) * k4.z + k4.w) * keep[j], alpha) : 0.f;
      cg_adj_put<KCM, VEC>(sH, c, pp, v);
      if (tap && ok) {
        if (VEC) *reinterpret_cast<float4*>(tap + idx) = make_float4(v[0], v[1], v[2], v[3]);
        else tap[idx] = v[0];
      }
    });
    __syncthreads();
    if (tile + 1 < tile1) cg_adj_fetch<KCM, VEC>(eb, Kc, Pn, p0 + K::PT, min(K::PT, Pn - p0 - K::PT), ebuf);
    for (int w = wave; w < MT * K::NP; w += CG_ADJ_THREADS / 64) {
      const int mt = w / K::NP, n0 = 32 * (w - mt * K::NP), n1 = n0 + 16;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      const float* ap = cg_tfrag_ptr<0>(sW + 16 * mt * K::WS, K::WS, l15, slot);
      const float* bp0 = cg_tfrag_ptr<1>(sH + n0, K::PS, l15, slot);
      const float* bp1 = cg_tfrag_ptr<1>(sH + n1, K::PS, l15, slot);
      for (int k0 = 0; k0 < K::KcM; k0 += 16) {
        float av[4], b0v[4], b1v[4];
        cg_tfrag<0>(ap, K::WS, k0, av); cg_tfrag<1>(bp0, K::PS, k0, b0v); cg_tfrag<1>(bp1, K::PS, k0, b1v);
#pragma unroll
        for (int s = 0; s < 4; ++s) {                     // C[position][channel]: four consecutive positions per lane
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0v[s], av[s], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1v[s], av[s], c1, 0, 0, 0);
        }
      }
      const int u = 16 * mt + l15;
      if (u < Kc) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pq = (h ? n1 : n0) + 4 * slot;
          const cg_f32x4 c = h ? c1 : c0;
          float* dst = ab + (long long)u * Pn + p0 + pq;
          if (VEC) { if (pq < np) *reinterpret_cast<float4*>(dst) = make_float4(c[0], c[1], c[2], c[3]); }
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (pq + q < np) dst[q] = c[q];
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(CG_ADJ_THREADS, 2) void cg_adj_m2_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom& g = pr.g[blockIdx.y];
  const int b = blockIdx.x / pr.nch_max, ch = blockIdx.x - b * pr.nch_max;
  if (ch >= g.nch) return;
  const int tile0 = ch * g.tpw, tile1 = min(g.ntiles, tile0 + g.tpw);
  if ((g.Pn & 3) == 0) { CG_ADJ_DISPATCH(g.KcM, (cg_adj_m2_body<KCM, true>(t, g, b, tile0, tile1, blockIdx.x == 0))) }
  else { CG_ADJ_DISPATCH(g.KcM, (cg_adj_m2_body<KCM, false>(t, g, b, tile0, tile1, blockIdx.x == 0))) }
}

// ======================================================================================================================
// N1: dh = W4^T dAdj;  g = dh PReLU'(u) keep -> HBM;  red = { sum g, sum g e_hat }, d alpha;  dW4 += dAdj h^T
// ======================================================================================================================
template <int KCM, bool VEC>
__device__ __forceinline__ void cg_adj_n1_body(const CgAdjTail& t, const CgAdjGeom& g, int b, int tile0, int tile1, int dbg) {
  using K = CgAdjK<KCM>;
  float* sE = reinterpret_cast<float*>(cg_dyn_lds);              // [KcM][PS] e_hat
  float* sP = sE + K::KcM * K::PS;                             // [KcM][PS] dropout keep factors
  float* sD = sP + K::KcM * K::PS;                             // [KcM][PS] dAdj
  float* sW = sD + K::KcM * K::PS;                             // [KcM][WS] W4
  float* sK = sW + K::KcM * K::WS;                                  // [KcM][8]
  double* sRed = reinterpret_cast<double*>(sK + 8 * K::KcM);       // [KcM][2] + [1]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4, nw = CG_ADJ_THREADS / 64;
  constexpr int N = VEC ? 4 : 1;
  // read once what the tile loop needs of the argument blocks (kernel-argument memory behind a run-time index)
  const int Kc = t.Kc, Pn = g.Pn;
  const bool train = t.train != 0;
  (void)train;
  const float* eb = t.e + (long long)b * Kc * Pn;
  const float* db = t.dadj + (long long)b * Kc * Pn;
  float ebuf[16], dbuf[16];
  {
    const int p0 = tile0 * K::PT, np = min(K::PT, Pn - p0);
    cg_adj_fetch<KCM, VEC>(eb, Kc, Pn, p0, np, ebuf);
    cg_adj_fetch<KCM, VEC>(db, Kc, Pn, p0, np, dbuf);
  }
  cg_adj_zero_pad_rows<KCM>(t, g, sE, 3);
  for (int e = tid; e < 2 * K::KcM + 1; e += CG_ADJ_THREADS) sRed[e] = 0.0;
  cg_adj_weight<KCM>(t.W4, t, g, sW);
  cg_adj_consts(t, g, sK, true, false);
  const float drop_p = t.drop_p; const unsigned int salt = t.salt;
  const bool drop = train && drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  const float alpha = t.alpha[0];
  constexpr int MT = K::MT;
  constexpr int NU = (MT * MT + 3) / 4;                  // weight-gradient tiles of a wave
  static_assert(MT * K::NP == 8 && NU <= CG_ADJ_MAXW, "task split of the four waves");
  constexpr int NS = NU == 1 ? 2 : 1;                    // a wave with one tile splits its sum over even / odd steps: two MFMA chains
  cg_f32x4 wacc[NU][NS];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int h = 0; h < NS; ++h) wacc[u][h] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  float* gb = t.g + (long long)b * Kc * Pn;
  float racc[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, sa = 0.f;   // per lane: sums of g and g e_hat (one channel per dh task), d alpha
  if (dbg & 64) tile1 = tile0;
  for (int tile = tile0; tile < tile1; ++tile) {
    const int p0 = tile * K::PT, np = min(K::PT, Pn - p0);
    __syncthreads();
    cg_adj_commit<KCM, VEC>(Kc, np, [&](int c, int pp, bool ok, int off) {
      const float2 k01 = *reinterpret_cast<const float2*>(sK + 8 * c);          // mean, rstd
      float keep[4], v[4], d[4];
      cg_adj_keeps<VEC>(drop_p, salt, drop && ok, seed, ((long long)b * Kc + c) * Pn + p0 + pp, keep);
#pragma unroll
      for (int j = 0; j < N; ++j) { v[j] = ok ? (ebuf[off + j] - k01.x) * k01.y : 0.f; keep[j] = ok ? keep[j] : 0.f; d[j] = dbuf[off + j]; }
      cg_adj_put<KCM, VEC>(sE, c, pp, v);
      cg_adj_put<KCM, VEC>(sP, c, pp, keep);
      cg_adj_put<KCM, VEC>(sD, c, pp, d);
    });
    __syncthreads();
    if (tile + 1 < tile1 && !(dbg & 8)) {
      const int q0 = p0 + K::PT, nq = min(K::PT, Pn - q0);
      cg_adj_fetch<KCM, VEC>(eb, Kc, Pn, q0, nq, ebuf);
      cg_adj_fetch<KCM, VEC>(db, Kc, Pn, q0, nq, dbuf);
    }
    // dW4[u'][u] += sum_p dAdj[u'][p] h[u][p],  h rebuilt from e_hat and the keep factors in the B fragments (lane = channel u)
    if (!(dbg & 1) && (MT * MT >= nw || wave < MT * MT)) {       // 1 tile (wave 0), 4 tiles (one per wave) or 16 (four per wave)
#pragma unroll
      for (int k0 = 0; k0 < K::PT; k0 += 16) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int id = u * nw + wave;
          {
            const int mt = id / MT, n2 = id - mt * MT, cu = 16 * n2 + l15;
            const bool cok = cu < Kc;
            const float gam = cok ? sK[8 * cu + 6] : 0.f, bet = cok ? sK[8 * cu + 3] : 0.f;
            float av[4], bv[4], keep[4];
            cg_tfrag<0>(cg_tfrag_ptr<0>(sD + 16 * mt * K::PS, K::PS, l15, slot), K::PS, k0, av);
            cg_tfrag<0>(cg_tfrag_ptr<0>(sE + 16 * n2 * K::PS, K::PS, l15, slot), K::PS, k0, bv);
            cg_tfrag<0>(cg_tfrag_ptr<0>(sP + 16 * n2 * K::PS, K::PS, l15, slot), K::PS, k0, keep);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              // rows of channels >= Kc and positions beyond the tensor hold keep = 0: h = 0
              const float h = cg_prelu((gam * bv[s] + bet) * keep[s], alpha);
              wacc[u][(k0 / 16) % NS] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], h, wacc[u][(k0 / 16) % NS], 0, 0, 0);
            }
          }
        }
      }
    }
    // dh[u][p] = sum_u' W4[u'][u] dAdj[u'][p]
    cg_f32x4 dacc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) dacc[ti][0] = dacc[ti][1] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(dbg & 2)) {
#pragma unroll
      for (int k0 = 0; k0 < K::KcM; k0 += 16) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          const int w = wave + nw * ti, mt = w / K::NP, n0 = 32 * (w - mt * K::NP), n1 = n0 + 16;
          float av[4], b0v[4], b1v[4];
          cg_tfrag<1>(cg_tfrag_ptr<1>(sW + 16 * mt, K::WS, l15, slot), K::WS, k0, av);
          cg_tfrag<1>(cg_tfrag_ptr<1>(sD + n0, K::PS, l15, slot), K::PS, k0, b0v);
          cg_tfrag<1>(cg_tfrag_ptr<1>(sD + n1, K::PS, l15, slot), K::PS, k0, b1v);
#pragma unroll
          for (int s = 0; s < 4; ++s) {                     // C[position][channel]: four consecutive positions per lane
            dacc[ti][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0v[s], av[s], dacc[ti][0], 0, 0, 0);
            dacc[ti][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1v[s], av[s], dacc[ti][1], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      if (dbg & 2) break;
      const int w = wave + nw * ti, mt = w / K::NP, n0 = 32 * (w - mt * K::NP), n1 = n0 + 16;
      const cg_f32x4 c0 = dacc[ti][0], c1 = dacc[ti][1];
      const int u = 16 * mt + l15;
      if (u < Kc) {
        const float gam = sK[8 * u + 6], bet = sK[8 * u + 3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pq = (h ? n1 : n0) + 4 * slot;
          if (pq >= np) continue;
          const cg_f32x4 c = h ? c1 : c0;
          const float4 e4 = *reinterpret_cast<const float4*>(sE + u * K::PS + pq), k4 = *reinterpret_cast<const float4*>(sP + u * K::PS + pq);
          const float ev[4] = {e4.x, e4.y, e4.z, e4.w}, kv[4] = {k4.x, k4.y, k4.z, k4.w};
          float gv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool in = VEC || pq + q < np;
            const float upre = (gam * ev[q] + bet) * kv[q];
            gv[q] = in ? (upre > 0.f ? c[q] : alpha * c[q]) * kv[q] : 0.f;
            racc[ti][0] += gv[q]; racc[ti][1] += gv[q] * ev[q];
            if (in && !(upre > 0.f)) sa += c[q] * upre;
          }
          float* dst = gb + (long long)u * Pn + p0 + pq;
          if (VEC) *reinterpret_cast<float4*>(dst) = make_float4(gv[0], gv[1], gv[2], gv[3]);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (pq + q < np) dst[q] = gv[q];
          }
        }
      }
    }
  }
  // the four lanes l15, l15 + 16, .. of a wave hold partial sums of one channel: straight into the workgroup's f64 words
  sa = cg_row16_sum(sa);
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int w = wave + nw * ti;
    const int u = 16 * (w / K::NP) + l15;
    if (u < Kc) { atomicAdd(&sRed[2 * u], (double)racc[ti][0]); atomicAdd(&sRed[2 * u + 1], (double)racc[ti][1]); }
  }
  if (l15 == 0) atomicAdd(&sRed[2 * K::KcM], (double)sa);
  __syncthreads();
  if (dbg & 32) return;
  double* red = t.red + (long long)(blockIdx.x % CG_ADJ_REPLICAS) * (2 * Kc + 1);
  for (int e = tid; e < 2 * Kc; e += CG_ADJ_THREADS) atomicAdd(&red[e], sRed[e]);
  if (tid == 0) atomicAdd(&red[2 * Kc], sRed[2 * K::KcM]);
  float* dW = t.dW4_ws + (long long)(blockIdx.x % CG_ADJ_REPLICAS) * Kc * Kc;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int id = u * nw + wave;
    if (id < MT * MT) {
      const int mt = id / MT, n2 = id - mt * MT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * mt + 4 * slot + q, c = 16 * n2 + l15;
        if (r < Kc && c < Kc) atomicAdd(&dW[r * Kc + c], NS == 2 ? wacc[u][0][q] + wacc[u][NS - 1][q] : wacc[u][0][q]);
      }
    }
  }
}

__global__ __launch_bounds__(CG_ADJ_THREADS, 2) void cg_adj_n1_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom& g = pr.g[blockIdx.y];
  const int b = blockIdx.x / pr.nch_max, ch = blockIdx.x - b * pr.nch_max;
  if (ch >= g.nch) return;
  const int tile0 = ch * g.tpw, tile1 = min(g.ntiles, tile0 + g.tpw);
  if ((g.Pn & 3) == 0) { CG_ADJ_DISPATCH(g.KcM, (cg_adj_n1_body<KCM, true>(t, g, b, tile0, tile1, pr.dbg))) }
  else { CG_ADJ_DISPATCH(g.KcM, (cg_adj_n1_body<KCM, false>(t, g, b, tile0, tile1, pr.dbg))) }
}

// Diagnostic build only (tools/stamps_adj.py compiles a private copy of the library with -DCG_ADJ_STAMPS): thread 0 of every workgroup
// of N2 stores the shader clock at its phase boundaries; nothing depends on it, the shipped library has no stamp.
#ifdef CG_ADJ_STAMPS
__device__ unsigned long long* cg_adj_stamp_buf = nullptr;
extern "C" int cg_adj_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(cg_adj_stamp_buf), &p, sizeof(p)); }
#define CG_ASTAMP()                                                                                                          \
  do {                                                                                                                       \
    if (threadIdx.x == 0 && cg_adj_stamp_buf && nst < 250)                                                                    \
      cg_adj_stamp_buf[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + (++nst)] = __builtin_amdgcn_s_memtime();      \
  } while (0)
#define CG_ASTAMP_END() do { if (threadIdx.x == 0 && cg_adj_stamp_buf) cg_adj_stamp_buf[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256] = nst; } while (0)
#else
#define CG_ASTAMP() do { } while (0)
#define CG_ASTAMP_END() do { } while (0)
#endif

// ======================================================================================================================
// N2: de = BN'(g);  do = W0^T de;  dS[k][a] += do Q[k][b'],  dQ[k][b'] += do S[k][a];  dW0 += de (S x Q)^T
// ======================================================================================================================
template <int KCM, bool VEC>
__device__ __forceinline__ void cg_adj_n2_body(const CgAdjTail& t, const CgAdjGeom& g, int b, int ch, int tile0, int tile1, int dbg) {
  using K = CgAdjK<KCM>;
  float* sS = reinterpret_cast<float*>(cg_dyn_lds);
  float* sQ = sS + K::KcM * g.JS;
  float* sDS = sQ + K::KcM * g.JS;                                 // dS of this chunk: cell (k, a) has one owner thread per tile, plain updates
  float* sDQ = sDS + K::KcM * g.JS;                                // dQ of this chunk: cell (k, b') belongs to one thread, plain updates
  float* sDE = sDQ + K::KcM * g.JS;                                // [KcM][PS] de
  float* sDO = sDE + K::KcM * K::PS;                           // [KcM][PS] do
  float* sW = sDO + K::KcM * K::PS;                            // [KcM][WS] W0
  float* sK = sW + K::KcM * K::WS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4, nw = CG_ADJ_THREADS / 64;
  constexpr int N = VEC ? 4 : 1;
  // what the tile loop reads of the argument blocks, once (they live in kernel-argument memory behind a run-time index: the compiler
  // re-read t.train in front of every element of the staging code)
  const int Kc = t.Kc, J = t.J, Pn = g.Pn, JS = g.JS;
  const unsigned magicJ = g.magicJ;
  const bool train = t.train != 0;
  const float* gsrc = t.g + (long long)b * Kc * Pn;
  const float* eb = t.e + (long long)b * Kc * Pn;
  float gbuf[16], ebuf[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ebuf[i] = 0.f;             // eval: e is not needed (its coefficient is zero)
  int nst = 0; (void)nst;
  CG_ASTAMP();
  {
    const int p0 = tile0 * K::PT, np = min(K::PT, Pn - p0);
    cg_adj_fetch<KCM, VEC>(gsrc, Kc, Pn, p0, np, gbuf);
    if (train) cg_adj_fetch<KCM, VEC>(eb, Kc, Pn, p0, np, ebuf);
  }
  cg_adj_tables<KCM>(t, g, b, sS, sQ);
  for (int e = tid; e < 2 * K::KcM * g.JS; e += CG_ADJ_THREADS) sDS[e] = 0.f;
  cg_adj_zero_pad_rows<KCM>(t, g, sDE, 2);
  cg_adj_weight<KCM>(t.W0, t, g, sW);
  cg_adj_consts(t, g, sK, true, false);
  constexpr int MT = K::MT; const int KJ = Kc * J;
  constexpr int NU = (MT * MT + 3) / 4;                  // weight-gradient tiles of a wave
  static_assert(MT * K::NP == 8 && NU <= CG_ADJ_MAXW, "task split of the four waves");
  constexpr int NS = NU == 1 ? 2 : 1;                    // a wave with one tile splits its sum over even / odd steps: two MFMA chains
  cg_f32x4 wacc[NU][NS];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int h = 0; h < NS; ++h) wacc[u][h] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  CG_ASTAMP();
  for (int tile = tile0; tile < tile1; ++tile) {
    const int p0 = tile * K::PT, np = min(K::PT, Pn - p0);
    __syncthreads();
    CG_ASTAMP();                                        // A: top barrier passed
    // de = scale (g - m1) - (e - mean) [scale rstd m2]  (eval: m1 = m2 = 0): the constants of a channel in two 16-byte LDS reads
    cg_adj_commit<KCM, VEC>(Kc, np, [&](int c, int pp, bool ok, int off) {
      const float4 ka = *reinterpret_cast<const float4*>(sK + 8 * c), kb = *reinterpret_cast<const float4*>(sK + 8 * c + 4);
      float v[4];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float x = ka.z * (gbuf[off + j] - kb.x) - (ebuf[off + j] - ka.x) * kb.w;
        v[j] = ok ? x : 0.f;
      }
      cg_adj_put<KCM, VEC>(sDE, c, pp, v);
    });
    CG_ASTAMP();                                        // A1: de committed
    // The seed tile o[k][p] = S[k][a] Q[k][b'] of this tile, ONCE, into the image that `do` takes later (it is free until then):
    // KcM * PT = 4096 = 16 consecutive positions of one slab per thread.  Rows k >= Kc of the tables are zero.
    {
      constexpr int per_row = K::PT >> 4;
      const int k = tid / per_row, pp0 = 16 * (tid - k * per_row), p = p0 + pp0;
      const int a = (int)cg_adj_div((unsigned)p, magicJ), bp = p - a * J;
      const float* srow = sS + k * JS; const float* qrow = sQ + k * JS;
      float* dst = sDO + k * K::PS + pp0;
      if (J >= 16) {
        // at most one step of a inside the 16 positions: no branch, no division per element, and no condition in front of an LDS read
        // (positions >= Pn get finite values: de is zero there, and `do` overwrites the image)
        const float s0 = srow[min(a, J)], s1 = srow[min(a + 1, J)];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int bj = bp + 4 * j4 + j;
            const bool wrap = bj >= J;
            v[j] = (wrap ? s1 : s0) * qrow[wrap ? bj - J : bj];
          }
          *reinterpret_cast<float4*>(dst + 4 * j4) = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
        int aa = a, bb = bp;
        float sv = aa < J ? srow[aa] : 0.f;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = p + 4 * j4 + j < Pn ? sv * qrow[bb] : 0.f;
            if (++bb == J) { bb = 0; ++aa; sv = aa < J ? srow[aa] : 0.f; }
          }
          *reinterpret_cast<float4*>(dst + 4 * j4) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    CG_ASTAMP();                                        // A2: seed image written
    __syncthreads();
    CG_ASTAMP();                                        // A3: barrier
    if (tile + 1 < tile1 && !(dbg & 8)) {
      const int q0 = p0 + K::PT, nq = min(K::PT, Pn - q0);
      cg_adj_fetch<KCM, VEC>(gsrc, Kc, Pn, q0, nq, gbuf);
      if (train) cg_adj_fetch<KCM, VEC>(eb, Kc, Pn, q0, nq, ebuf);
    }
    CG_ASTAMP();                                        // B: images committed, prefetch issued
    // dW0[u][k] += sum_p de[u][p] o[k][p]: the (up to four) output tiles of a wave advance together through the tile's positions, so
    // their MFMA chains are independent and the LDS reads of a step are issued in front of all of them
    if (!(dbg & 1) && (MT * MT >= nw || wave < MT * MT)) {       // 1 tile (wave 0), 4 tiles (one per wave) or 16 (four per wave)
#pragma unroll
      for (int k0 = 0; k0 < K::PT; k0 += 16) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int id = u * nw + wave;
          {
            const int mt = id / MT, n2 = id - mt * MT;
            float av[4], ov[4];
            cg_tfrag<0>(cg_tfrag_ptr<0>(sDE + 16 * mt * K::PS, K::PS, l15, slot), K::PS, k0, av);
            cg_tfrag<0>(cg_tfrag_ptr<0>(sDO + 16 * n2 * K::PS, K::PS, l15, slot), K::PS, k0, ov);
#pragma unroll
            for (int s = 0; s < 4; ++s) wacc[u][(k0 / 16) % NS] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], ov[s], wacc[u][(k0 / 16) % NS], 0, 0, 0);
          }
        }
      }
    }
    CG_ASTAMP();                                        // C: dW0 done (this wave)
    __syncthreads();                                    // the seed image has been read: `do` overwrites it
    CG_ASTAMP();                                        // D: barrier
    // do[k][p] = sum_u W0[u][k] de[u][p]  ->  image for the dS / dQ cells
    // (always eight tasks = two per wave: MT * NP == 8; both advance through k together: four independent MFMA chains)
    cg_f32x4 dacc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) dacc[ti][0] = dacc[ti][1] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(dbg & 2)) {
#pragma unroll
      for (int k0 = 0; k0 < K::KcM; k0 += 16) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          const int w = wave + nw * ti, mt = w / K::NP, n0 = 32 * (w - mt * K::NP), n1 = n0 + 16;
          float av[4], b0v[4], b1v[4];
          cg_tfrag<1>(cg_tfrag_ptr<1>(sW + 16 * mt, K::WS, l15, slot), K::WS, k0, av);
          cg_tfrag<1>(cg_tfrag_ptr<1>(sDE + n0, K::PS, l15, slot), K::PS, k0, b0v);
          cg_tfrag<1>(cg_tfrag_ptr<1>(sDE + n1, K::PS, l15, slot), K::PS, k0, b1v);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            dacc[ti][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0v[s], av[s], dacc[ti][0], 0, 0, 0);
            dacc[ti][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1v[s], av[s], dacc[ti][1], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int ti = 0; ti < 2 && !(dbg & 2); ++ti) {
      const int w = wave + nw * ti, mt = w / K::NP, n0 = 32 * (w - mt * K::NP), n1 = n0 + 16;
      // C[position][slab]: lane = slab k, four consecutive positions: the do image in one float4
      const int k = 16 * mt + l15;
      const bool kok = k < Kc;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pq = (h ? n1 : n0) + 4 * slot;
        const cg_f32x4 c = h ? dacc[ti][1] : dacc[ti][0];
        float d[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = (kok && pq + q < np) ? c[q] : 0.f;
        *reinterpret_cast<float4*>(sDO + k * K::PS + pq) = make_float4(d[0], d[1], d[2], d[3]);
      }
    }
    __syncthreads();
    CG_ASTAMP();                                        // E: do done, barrier passed
    // The tile covers the positions p0 .. p0 + np - 1 = (a0, r0) .. in row-major (a, b') order.  Both reductions of the `do` image are
    // owned cells (no atomics): dQ[k][b'] += sum_m do[k][b' - r0 + J m] S[k][a0 + m]  (one cell per (k, b'), a few terms each) and
    // dS[k][a0 + m] += sum_b' do[k][J m - r0 + b'] Q[k][b']  (one cell per (k, m): a contiguous run of the image against a row of Q).
    // Measured alternatives (profiles/r04_stamps.txt): LDS float atomics straight from the result registers (24 ds_add_f32 per wave and
    // tile: 25 000 cycles, against 6 000 for this pass); four lanes per dS cell with a DPP sum, register-resident dQ cells: no gain -
    // the pass is bound by the index arithmetic in front of each term, not by its LDS reads.
    const int a0 = (int)cg_adj_div((unsigned)p0, magicJ), r0 = p0 - a0 * J;
    if (!(dbg & 16)) {
      for (int cell = tid; cell < KJ; cell += CG_ADJ_THREADS) {
        const int k = (int)cg_adj_div((unsigned)cell, magicJ), bq = cell - k * J;
        const float* img = sDO + k * K::PS; const float* srow = sS + k * JS + a0;
        float acc = 0.f;
        int m = bq < r0 ? 1 : 0;
        for (int pp = bq - r0 + J * m; pp < np; pp += J, ++m) acc += img[pp] * srow[m];
        sDQ[k * JS + bq] += acc;
      }
      if (J >= 16) {
        // dS by (slab, 16 consecutive positions) like the seed image above: at most one step of a inside the run, so a thread ends with
        // two partial sums, for rows a and a + 1 (clamped to the pad column behind the last row).  The PT / 16 threads of a slab are
        // adjacent lanes of one wave: they add their sums to the slab's cells one after the other - the LDS operations of a wave are
        // executed in order, the wave barrier keeps the compiler from merging the phases
        constexpr int per_row = K::PT >> 4;
        const int k = tid / per_row, sub = tid - k * per_row, pp0 = 16 * sub, p = p0 + pp0;
        const int a = (int)cg_adj_div((unsigned)p, magicJ), bp = p - a * J;
        const float* img = sDO + k * K::PS + pp0; const float* qrow = sQ + k * JS;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const float4 d4 = *reinterpret_cast<const float4*>(img + 4 * j4);
          const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int bj = bp + 4 * j4 + j;
            const bool wrap = bj >= J;
            const float pr = dv[j] * qrow[wrap ? bj - J : bj];        // the image is zero beyond np and in the rows k >= Kc
            s0 += wrap ? 0.f : pr; s1 += wrap ? pr : 0.f;
          }
        }
        float* c0 = sDS + k * JS + min(a, J); float* c1 = sDS + k * JS + min(a + 1, J);
#pragma unroll
        for (int j = 0; j < per_row; ++j) {
          if (sub == j) { *c0 += s0; *c1 += s1; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      } else {
      const int M = (r0 + np - 1) / J + 1;               // rows a of the position grid that the tile touches
      for (int cell = tid; cell < Kc * M; cell += CG_ADJ_THREADS) {
        const int k = cell / M, m = cell - k * M;
        const int lo = max(0, J * m - r0), hi = min(np, J * (m + 1) - r0);
        const float* img = sDO + k * K::PS; const float* qrow = sQ + k * JS + (r0 - J * m);
        float acc0 = 0.f, acc1 = 0.f;
        int pp = lo;
        for (; pp + 1 < hi; pp += 2) { acc0 += img[pp] * qrow[pp]; acc1 += img[pp + 1] * qrow[pp + 1]; }
        if (pp < hi) acc0 += img[pp] * qrow[pp];
        if (a0 + m < J) sDS[k * JS + a0 + m] += acc0 + acc1;
      }
      }
    }
  }
  __syncthreads();
  CG_ASTAMP();
  // this chunk's share of dS / dQ: [b][chunk][2][Kc * J], summed over the chunks by cg_adj_finish_kernel
  if (dbg & 32) return;
  float* part = t.part + ((long long)b * g.nch + ch) * 2 * KJ;
  for (int cell = tid; cell < KJ; cell += CG_ADJ_THREADS) {
    const int k = (int)cg_adj_div((unsigned)cell, g.magicJ), a = cell - k * t.J;
    part[cell] = sDS[k * g.JS + a];
    part[KJ + cell] = sDQ[k * g.JS + a];
  }
  float* dW = t.dW0_ws + (long long)(blockIdx.x % CG_ADJ_REPLICAS) * t.Kc * t.Kc;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int id = u * nw + wave;
    if (id < MT * MT) {
      const int mt = id / MT, n2 = id - mt * MT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * mt + 4 * slot + q, c = 16 * n2 + l15;
        if (r < t.Kc && c < t.Kc) atomicAdd(&dW[r * t.Kc + c], NS == 2 ? wacc[u][0][q] + wacc[u][NS - 1][q] : wacc[u][0][q]);
      }
    }
  }
  CG_ASTAMP();
  CG_ASTAMP_END();
}

__global__ __launch_bounds__(CG_ADJ_THREADS, 2) void cg_adj_n2_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom& g = pr.g[blockIdx.y];
  const int b = blockIdx.x / pr.nch_max, ch = blockIdx.x - b * pr.nch_max;
  if (ch >= g.nch) return;
  const int tile0 = ch * g.tpw, tile1 = min(g.ntiles, tile0 + g.tpw);
  if ((g.Pn & 3) == 0) { CG_ADJ_DISPATCH(g.KcM, (cg_adj_n2_body<KCM, true>(t, g, b, ch, tile0, tile1, pr.dbg))) }
  else { CG_ADJ_DISPATCH(g.KcM, (cg_adj_n2_body<KCM, false>(t, g, b, ch, tile0, tile1, pr.dbg))) }
}

// per-channel parameter gradients, fold of the replicated weight gradients and of the per-chunk dS / dQ (both towers)
__global__ void cg_adj_finish_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom& g = pr.g[blockIdx.y];
  const int n = t.Kc * t.Kc, KJ = t.Kc * t.J;
  const int V = t.domain == 0 ? t.Kc : t.J, T = t.domain == 0 ? t.J : t.Kc;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)t.B * KJ; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / KJ), e = (int)(i - (long long)b * KJ), k = e / t.J, a = e - k * t.J;
    const float* part = t.part + (long long)b * g.nch * 2 * KJ + e;
    float sd = 0.f, sq = 0.f;
    for (int c = 0; c < g.nch; ++c) { sd += part[(long long)c * 2 * KJ]; sq += part[(long long)c * 2 * KJ + KJ]; }
    // layouts of s (V,T) and q (T,V)
    float* dsb = t.ds + (long long)b * V * T;
    float* dqb = t.dq + (long long)b * T * V;
    if (t.domain == 0) { dsb[k * T + a] = sd; dqb[a * V + k] = sq; }
    else { dsb[a * T + k] = sd; dqb[k * V + a] = sq; }
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s0 = 0.f, s4 = 0.f;
    for (int r = 0; r < CG_ADJ_REPLICAS; ++r) { s0 += t.dW0_ws[(long long)r * n + i]; s4 += t.dW4_ws[(long long)r * n + i]; }
    t.dW0[i] = s0; t.dW4[i] = s4;
  }
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < t.Kc; c += blockDim.x) {
      t.dgamma[c] = (float)cg_adj_red(t, 2 * c + 1); t.dbeta[c] = (float)cg_adj_red(t, 2 * c);
      if (c == 0) t.dalpha[0] = (float)cg_adj_red(t, 2 * t.Kc);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static unsigned cg_adj_magic(int d) { return d > 1 ? (unsigned)((0x100000000ULL + d - 1) / d) : 0u; }
// tiles per workgroup: long workgroups amortise their prologue (weights, tables) when the batch alone fills the chip
static int cg_adj_tpw(int B) { return B >= 128 ? CG_ADJ_TPW_MAX : B >= 64 ? 4 : B >= 32 ? 2 : 1; }
// padded slab count of a tile: 16 / 32 / 64.  The staging code hands every thread 16 floats of a [KcM][PT] tile
// (KcM * PT == 4096), which has no 48-row form: 33..48 slabs take the 64-row tile with zero rows behind Kc
static int cg_adj_kcm(int Kc) { return Kc > 32 ? 64 : (Kc + 15) & ~15; }
static CgAdjGeom cg_adj_geometry(int B, int Kc, int J) {
  CgAdjGeom g;
  g.KcM = cg_adj_kcm(Kc);
  g.WS = g.KcM + 4;
  g.JS = J + 1;
  g.Pn = J * J;
  g.PT = g.KcM > 32 ? 64 : g.KcM > 16 ? 128 : 256;       // KcM * PT = 4096
  g.PS = g.PT + 4;
  g.NP = g.PT / 32;                                      // pairs of 16-position column groups per tile
  g.lgq = g.KcM > 32 ? 4 : g.KcM > 16 ? 5 : 6;           // log2(PT / 4)
  g.ntiles = (g.Pn + g.PT - 1) / g.PT;
  g.tpw = cg_adj_tpw(B);
  g.nch = (g.ntiles + g.tpw - 1) / g.tpw;
  g.magicJ = cg_adj_magic(J); g.magicKc = cg_adj_magic(Kc); g.magicPad = cg_adj_magic(g.WS - Kc);
  return g;
}
// ablation mask of the backward kernels (a private library built with -DCG_ABLATION; profiles/r02_ablations.txt): results are WRONG
// with it set; the shipped library is compiled without the flag and cannot skip a phase
#ifdef CG_ABLATION
static int cg_adj_dbg() { static const int v = getenv("CG_ADJ_DBG") ? atoi(getenv("CG_ADJ_DBG")) : 0; return v; }
#else
static int cg_adj_dbg() { return 0; }
#endif
static int cg_adj_tile(int Kc) { const int KcM = cg_adj_kcm(Kc); return KcM > 32 ? 64 : KcM > 16 ? 128 : 256; }

static int cg_adj_check(const CgAdjTail* it, int n) {
  if (!it || n <= 0 || n > 2) return CG_EARG;
  for (int i = 0; i < n; ++i) {
    const CgAdjTail& t = it[i];
    if (t.B <= 0 || t.Kc <= 0 || t.J <= 0 || t.Kc > 64 || t.J > 64 || (t.domain != 0 && t.domain != 1) || t.B != it[0].B) return CG_ESHAPE;
    if (!t.s || !t.q || !t.W0 || !t.W4 || !t.alpha || !t.bn.gamma || !t.bn.beta || !t.bn.save || !t.e) return CG_EARG;
    if (t.train && t.drop_p > 0.f && !t.seed) return CG_EARG;
  }
  return CG_OK;
}

static size_t cg_adj_lds(const CgAdjTail* it, int n, int phase, bool bwd) {
  size_t best = 0;
  for (int i = 0; i < n; ++i) {
    const int KcM = cg_adj_kcm(it[i].Kc), WS = KcM + 4, JS = it[i].J + 1, PS = cg_adj_tile(it[i].Kc) + 4;
    size_t f;
    if (!bwd && phase == 1) f = (size_t)2 * KcM * JS + (size_t)KcM * WS + 2 + (size_t)4 * KcM;                 // tables, W0, f64 sums
    else if (!bwd) f = (size_t)KcM * PS + (size_t)KcM * WS + (size_t)8 * KcM;
    else if (phase == 1) f = (size_t)3 * KcM * PS + (size_t)KcM * WS + (size_t)8 * KcM + (size_t)2 * (2 * KcM + 1) + 2;
    else f = (size_t)4 * KcM * JS + (size_t)2 * KcM * PS + (size_t)KcM * WS + (size_t)8 * KcM;
    best = f > best ? f : best;
  }
  return best * sizeof(float) + 16;
}

static int cg_adj_chunks(int B, int Kc, int J) { const int PT = cg_adj_tile(Kc), tpw = cg_adj_tpw(B); return ((J * J + PT - 1) / PT + tpw - 1) / tpw; }
static int cg_adj_max_chunks(const CgAdjTail* it, int n) {
  int m = 1;
  for (int i = 0; i < n; ++i) m = cg_adj_chunks(it[i].B, it[i].Kc, it[i].J) > m ? cg_adj_chunks(it[i].B, it[i].Kc, it[i].J) : m;
  return m;
}

// zeroed scratch of one tower: the replicated weight gradients (two halves: dW0_ws, dW4_ws) / f64 words of `red`
extern "C" long long cg_map2adj_tail_ws_floats(int Kc) { return (long long)2 * CG_ADJ_REPLICAS * Kc * Kc; }
extern "C" long long cg_map2adj_tail_red_doubles(int Kc) { return (long long)CG_ADJ_REPLICAS * (2 * Kc + 1); }
// floats of the per-chunk dS / dQ scratch `part` (no zeroing needed)
extern "C" long long cg_map2adj_tail_part_floats(int B, int Kc, int J) { return (long long)B * cg_adj_chunks(B, Kc, J) * 2 * Kc * J; }

// include/cistgcn_hip.h : cg_map2adj_tail_fwd (phases 1, 2) / cg_map2adj_tail_bwd (phases 1, 2)
extern "C" int cg_map2adj_tail_fwd(const CgAdjTail* items, int n, int phase, void* stream_) {
  int st = cg_adj_check(items, n);
  if (st != CG_OK) return st;
  CgAdjTailPair pr;
  pr.n = n; pr.nch_max = cg_adj_max_chunks(items, n); pr.dbg = cg_adj_dbg(); pr.pad = 0;
  for (int i = 0; i < n; ++i) pr.g[i] = cg_adj_geometry(items[i].B, items[i].Kc, items[i].J);
  for (int i = 0; i < n; ++i) {
    pr.t[i] = items[i];
    if (phase == 1 && items[i].train && !items[i].bn.stats) return CG_EARG;
    if (phase == 2 && !items[i].adj) return CG_EARG;
  }
  const size_t lds = cg_adj_lds(items, n, phase, false);
  dim3 grid((unsigned)(items[0].B * pr.nch_max), (unsigned)n), block(CG_ADJ_THREADS);
  hipStream_t stream = (hipStream_t)stream_;
  if (phase == 1) {
    hipError_t e = cg_lds_limit((const void*)cg_adj_m1_kernel, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_adj_m1_kernel, grid, block, lds, stream, pr);
  } else if (phase == 2) {
    hipError_t e = cg_lds_limit((const void*)cg_adj_m2_kernel, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_adj_m2_kernel, grid, block, lds, stream, pr);
  } else return CG_EARG;
  return cg_launch_status();
}

extern "C" int cg_map2adj_tail_bwd(const CgAdjTail* items, int n, int phase, void* stream_) {
  int st = cg_adj_check(items, n);
  if (st != CG_OK) return st;
  CgAdjTailPair pr;
  pr.n = n; pr.nch_max = cg_adj_max_chunks(items, n); pr.dbg = cg_adj_dbg(); pr.pad = 0;
  for (int i = 0; i < n; ++i) pr.g[i] = cg_adj_geometry(items[i].B, items[i].Kc, items[i].J);
  for (int i = 0; i < n; ++i) {
    const CgAdjTail& t = items[i];
    pr.t[i] = t;
    if (!t.dadj || !t.g || !t.red || !t.dW0_ws || !t.dW4_ws) return CG_EARG;
    if (phase == 2 && (!t.ds || !t.dq || !t.part || !t.dW0 || !t.dW4 || !t.dgamma || !t.dbeta || !t.dalpha)) return CG_EARG;
  }
  const size_t lds = cg_adj_lds(items, n, phase, true);
  dim3 grid((unsigned)(items[0].B * pr.nch_max), (unsigned)n), block(CG_ADJ_THREADS);
  hipStream_t stream = (hipStream_t)stream_;
  if (phase == 1) {
    hipError_t e = cg_lds_limit((const void*)cg_adj_n1_kernel, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_adj_n1_kernel, grid, block, lds, stream, pr);
  } else if (phase == 2) {
    hipError_t e = cg_lds_limit((const void*)cg_adj_n2_kernel, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_adj_n2_kernel, grid, block, lds, stream, pr);
    st = cg_launch_status();
    if (st != CG_OK) return st;
    hipLaunchKernelGGL(cg_adj_finish_kernel, dim3(128, (unsigned)n), dim3(256), 0, stream, pr);
  } else return CG_EARG;
  return cg_launch_status();
}
