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

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

#define CG_ADJ_PT 64
#define CG_ADJ_PS (CG_ADJ_PT + 4)
#define CG_ADJ_THREADS 256
#define CG_ADJ_REPLICAS 16

struct CgAdjGeom { int KcM, WS, JS, Pn; unsigned magicJ; };

__device__ __forceinline__ CgAdjGeom cg_adj_geom(const CgAdjTail& t) {
  CgAdjGeom g;
  g.KcM = (t.Kc + 15) & ~15;
  g.WS = g.KcM + 4;
  g.JS = t.J + 1;                      // odd-ish row stride of the S / Q tables
  g.Pn = t.J * t.J;
  g.magicJ = t.J > 1 ? (unsigned)((0x100000000ULL + t.J - 1) / t.J) : 0u;
  return g;
}
__device__ __forceinline__ unsigned cg_adj_div(unsigned n, unsigned magic) { return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n; }

// S[k][a], Q[k][b'] of sample b: [KcM][JS] tables, rows >= Kc zero
__device__ __forceinline__ void cg_adj_tables(const CgAdjTail& t, const CgAdjGeom& g, int b, float* sS, float* sQ) {
  const int V = t.domain == 0 ? t.Kc : t.J, T = t.domain == 0 ? t.J : t.Kc;
  const float* sb = t.s + (long long)b * V * T;      // (V, T)
  const float* qb = t.q + (long long)b * T * V;      // (T, V)
  for (int e = threadIdx.x; e < g.KcM * g.JS; e += CG_ADJ_THREADS) {
    const int k = e / g.JS, a = e - k * g.JS;
    float sv = 0.f, qv = 0.f;
    if (k < t.Kc && a < t.J) {
      if (t.domain == 0) { sv = sb[k * T + a]; qv = qb[a * V + k]; }       // k = joint: S = s[v][t], Q = q[tau][v]
      else { sv = sb[a * T + k]; qv = qb[k * V + a]; }                     // k = frame: S = s[v][t], Q = q[t][w]
    }
    sS[e] = sv; sQ[e] = qv;
  }
}

__device__ __forceinline__ void cg_adj_weight(const float* __restrict__ W, int Kc, int KcM, int WS, float* sW) {
  for (int e = threadIdx.x; e < KcM * WS; e += CG_ADJ_THREADS) {
    const int r = e / WS, c = e - r * WS;
    sW[e] = (r < Kc && c < Kc) ? W[r * Kc + c] : 0.f;
  }
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
__global__ __launch_bounds__(CG_ADJ_THREADS) void cg_adj_m1_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom g = cg_adj_geom(t);
  const int b = blockIdx.x;
  float* sS = reinterpret_cast<float*>(cg_dyn_lds);
  float* sQ = sS + g.KcM * g.JS;
  float* sW = sQ + g.KcM * g.JS;
  double* sStat = reinterpret_cast<double*>(sW + g.KcM * g.WS + ((g.KcM * g.JS * 2 + g.KcM * g.WS) & 1));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, slot = lane >> 4;
  cg_adj_tables(t, g, b, sS, sQ);
  cg_adj_weight(t.W0, t.Kc, g.KcM, g.WS, sW);
  for (int e = tid; e < 2 * g.KcM; e += CG_ADJ_THREADS) sStat[e] = 0.0;
  __syncthreads();
  const int MT = g.KcM / 16, ntiles = (g.Pn + CG_ADJ_PT - 1) / CG_ADJ_PT;
  float* eb = t.e + (long long)b * t.Kc * g.Pn;
  for (int w = wave; w < ntiles * MT * 2; w += CG_ADJ_THREADS / 64) {       // (tile, u tile, pair of position tiles)
    const int tile = w / (MT * 2), r = w - tile * (MT * 2), mt = r >> 1, p0 = tile * CG_ADJ_PT + 32 * (r & 1);
    const int pa = p0 + l15, pb = p0 + 16 + l15;
    const int aa = (int)cg_adj_div((unsigned)pa, g.magicJ), ab = (int)cg_adj_div((unsigned)pb, g.magicJ);
    const bool oka = pa < g.Pn, okb = pb < g.Pn;
    cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    const float* ap = cg_tfrag_ptr<0>(sW + 16 * mt * g.WS, g.WS, l15, slot);
    for (int k0 = 0; k0 < g.KcM; k0 += 16) {
      float av[4], b0v[4], b1v[4];
      cg_tfrag<0>(ap, g.WS, k0, av);
      cg_adj_seed_frag_k(sS, sQ, g.JS, k0, slot, oka ? aa : 0, oka ? pa - aa * t.J : 0, oka, b0v);
      cg_adj_seed_frag_k(sS, sQ, g.JS, k0, slot, okb ? ab : 0, okb ? pb - ab * t.J : 0, okb, b1v);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b0v[s], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b1v[s], c1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = 16 * mt + 4 * slot + q;
      const bool uok = u < t.Kc;
      const float v0 = (uok && oka) ? c0[q] : 0.f, v1 = (uok && okb) ? c1[q] : 0.f;
      if (uok && oka) eb[(long long)u * g.Pn + pa] = v0;
      if (uok && okb) eb[(long long)u * g.Pn + pb] = v1;
      if (t.train) {
        float s1 = v0 + v1, s2 = v0 * v0 + v1 * v1;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
        if (l15 == 0 && uok) { atomicAdd(&sStat[2 * u], (double)s1); atomicAdd(&sStat[2 * u + 1], (double)s2); }
      }
    }
  }
  if (t.train) {
    __syncthreads();
    double* rep = t.bn.stats + (long long)(blockIdx.x % CG_STAT_REPLICAS) * 2 * t.Kc;
    for (int e = tid; e < 2 * t.Kc; e += CG_ADJ_THREADS) atomicAdd(&rep[e], sStat[e]);
  }
}

// per-channel constants [KcM][8]: mean, rstd, scale = gamma * rstd, beta, m1, m2 (backward), -, -
__device__ __forceinline__ void cg_adj_consts(const CgAdjTail& t, const CgAdjGeom& g, float* sK, bool backward, bool owner) {
  const double cnt = (double)t.B * g.Pn;
  for (int c = threadIdx.x; c < t.Kc; c += CG_ADJ_THREADS) {
    const CgAff a = cg_tail_aff(t.bn, c, t.Kc, cnt, t.train, backward, owner);
    float* k = sK + 8 * c;
    k[0] = a.mean; k[1] = a.rstd; k[2] = a.gamma * a.rstd; k[3] = a.beta;
    k[4] = (backward && t.train) ? (float)(t.red[2 * c] / cnt) : 0.f;
    k[5] = (backward && t.train) ? (float)(t.red[2 * c + 1] / cnt) : 0.f;
  }
}

__device__ __forceinline__ float cg_adj_keep(const CgAdjTail& t, unsigned long long seed, long long idx) {
  if (!(t.train && t.drop_p > 0.f)) return 1.f;
  return cg_drop_scale(t.drop_p, seed, t.salt, (unsigned long long)idx);
}

// stage one [Kc][64-position] tile of a (B,Kc,J,J) tensor through `fn(value, channel, flat index)` into an LDS image
template <typename F>
__device__ __forceinline__ void cg_adj_stage(const CgAdjTail& t, const CgAdjGeom& g, const float* __restrict__ src, int b, int p0, float* img, F fn) {
  const int np = min(CG_ADJ_PT, g.Pn - p0);
  if ((g.Pn & 3) == 0) {
#pragma unroll 2
    for (int e = threadIdx.x; e < t.Kc * (CG_ADJ_PT / 4); e += CG_ADJ_THREADS) {
      const int c = e / (CG_ADJ_PT / 4), pp = 4 * (e - c * (CG_ADJ_PT / 4));
      float val[4] = {0.f, 0.f, 0.f, 0.f};
      if (pp < np) {
        const long long off = ((long long)b * t.Kc + c) * g.Pn + p0 + pp;
        const float4 x4 = *reinterpret_cast<const float4*>(src + off);
        fn(x4, c, off, val);
      }
      *reinterpret_cast<float4*>(img + c * CG_ADJ_PS + pp) = make_float4(val[0], val[1], val[2], val[3]);
    }
  } else {
    for (int e = threadIdx.x; e < t.Kc * CG_ADJ_PT; e += CG_ADJ_THREADS) {
      const int c = e / CG_ADJ_PT, pp = e - c * CG_ADJ_PT;
      float val[4] = {0.f, 0.f, 0.f, 0.f};
      if (pp < np) {
        const long long off = ((long long)b * t.Kc + c) * g.Pn + p0 + pp;
        fn(make_float4(src[off], 0.f, 0.f, 0.f), c, -off - 1, val);        // negative index: scalar element `-(idx) - 1`
      }
      img[c * CG_ADJ_PS + pp] = val[0];
    }
  }
}

// ======================================================================================================================
// M2: Adj[u'][p] = sum_u W4[u'][u] h[u][p],  h = PReLU(Dropout(BN(e)))
// ======================================================================================================================
__global__ __launch_bounds__(CG_ADJ_THREADS, 2) void cg_adj_m2_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom g = cg_adj_geom(t);
  const int b = blockIdx.x;
  float* sH = reinterpret_cast<float*>(cg_dyn_lds);              // [KcM][PS]
  float* sW = sH + g.KcM * CG_ADJ_PS;                             // [KcM][WS]
  float* sK = sW + g.KcM * g.WS;                                  // [KcM][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, slot = lane >> 4;
  for (int e = tid; e < g.KcM * CG_ADJ_PS; e += CG_ADJ_THREADS) sH[e] = 0.f;
  cg_adj_weight(t.W4, t.Kc, g.KcM, g.WS, sW);
  cg_adj_consts(t, g, sK, false, b == 0);
  const unsigned long long seed = (t.train && t.drop_p > 0.f) ? *t.seed : 0ull;
  const float alpha = t.alpha[0];
  const bool drop = t.train && t.drop_p > 0.f;
  const int MT = g.KcM / 16, ntiles = (g.Pn + CG_ADJ_PT - 1) / CG_ADJ_PT;
  float* ab = t.adj + (long long)b * t.Kc * g.Pn;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int p0 = tile * CG_ADJ_PT, np = min(CG_ADJ_PT, g.Pn - p0);
    __syncthreads();
    cg_adj_stage(t, g, t.e, b, p0, sH, [&](float4 x4, int c, long long idx, float val[4]) {
      const float* k = sK + 8 * c;
      if (idx >= 0) {
        float keep[4];
        cg_keep4(drop, t.drop_p, seed, t.salt, (unsigned long long)idx, keep);
        const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = cg_prelu(((xv[j] - k[0]) * k[2] + k[3]) * keep[j], alpha);
      } else val[0] = cg_prelu(((x4.x - k[0]) * k[2] + k[3]) * cg_adj_keep(t, seed, -idx - 1), alpha);
    });
    __syncthreads();
    if (t.tap)
      for (int e = tid; e < t.Kc * CG_ADJ_PT; e += CG_ADJ_THREADS) {
        const int c = e / CG_ADJ_PT, pp = e - c * CG_ADJ_PT;
        if (pp < np) t.tap[((long long)b * t.Kc + c) * g.Pn + p0 + pp] = sH[c * CG_ADJ_PS + pp];
      }
    for (int w = wave; w < MT * 2; w += CG_ADJ_THREADS / 64) {
      const int mt = w >> 1, n0 = 32 * (w & 1), n1 = n0 + 16;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      const float* ap = cg_tfrag_ptr<0>(sW + 16 * mt * g.WS, g.WS, l15, slot);
      const float* bp0 = cg_tfrag_ptr<1>(sH + n0, CG_ADJ_PS, l15, slot);
      const float* bp1 = cg_tfrag_ptr<1>(sH + n1, CG_ADJ_PS, l15, slot);
      for (int k0 = 0; k0 < g.KcM; k0 += 16) {
        float av[4], b0v[4], b1v[4];
        cg_tfrag<0>(ap, g.WS, k0, av); cg_tfrag<1>(bp0, CG_ADJ_PS, k0, b0v); cg_tfrag<1>(bp1, CG_ADJ_PS, k0, b1v);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b0v[s], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b1v[s], c1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u = 16 * mt + 4 * slot + q;
        if (u >= t.Kc) continue;
        if (n0 + l15 < np) ab[(long long)u * g.Pn + p0 + n0 + l15] = c0[q];
        if (n1 + l15 < np) ab[(long long)u * g.Pn + p0 + n1 + l15] = c1[q];
      }
    }
  }
}

// ======================================================================================================================
// N1: dh = W4^T dAdj;  g = dh PReLU'(u) keep -> HBM;  red = { sum g, sum g e_hat }, d alpha;  dW4 += dAdj h^T
// ======================================================================================================================
__global__ __launch_bounds__(CG_ADJ_THREADS, 2) void cg_adj_n1_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom g = cg_adj_geom(t);
  const int b = blockIdx.x;
  float* sE = reinterpret_cast<float*>(cg_dyn_lds);              // [KcM][PS] e_hat
  float* sD = sE + g.KcM * CG_ADJ_PS;                             // [KcM][PS] dAdj
  float* sW = sD + g.KcM * CG_ADJ_PS;                             // [KcM][WS] W4
  float* sK = sW + g.KcM * g.WS;                                  // [KcM][8]
  double* sRed = reinterpret_cast<double*>(sK + 8 * g.KcM);       // [KcM][2] + [1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, slot = lane >> 4, nw = CG_ADJ_THREADS / 64;
  for (int e = tid; e < 2 * g.KcM * CG_ADJ_PS; e += CG_ADJ_THREADS) sE[e] = 0.f;
  for (int e = tid; e < 2 * g.KcM + 1; e += CG_ADJ_THREADS) sRed[e] = 0.0;
  cg_adj_weight(t.W4, t.Kc, g.KcM, g.WS, sW);
  cg_adj_consts(t, g, sK, true, false);
  const unsigned long long seed = (t.train && t.drop_p > 0.f) ? *t.seed : 0ull;
  const float alpha = t.alpha[0];
  const bool drop = t.train && t.drop_p > 0.f, vec = (g.Pn & 3) == 0;
  const int MT = g.KcM / 16, ntiles = (g.Pn + CG_ADJ_PT - 1) / CG_ADJ_PT;
  cg_f32x4 wacc[CG_ADJ_MAXW];
#pragma unroll
  for (int u = 0; u < CG_ADJ_MAXW; ++u) wacc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  float* gb = t.g + (long long)b * t.Kc * g.Pn;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int p0 = tile * CG_ADJ_PT, np = min(CG_ADJ_PT, g.Pn - p0);
    __syncthreads();
    cg_adj_stage(t, g, t.e, b, p0, sE, [&](float4 x4, int c, long long idx, float val[4]) {
      const float* k = sK + 8 * c;
      const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) val[j] = (xv[j] - k[0]) * k[1];
    });
    cg_adj_stage(t, g, t.dadj, b, p0, sD, [&](float4 x4, int c, long long idx, float val[4]) { val[0] = x4.x; val[1] = x4.y; val[2] = x4.z; val[3] = x4.w; });
    __syncthreads();
    // dW4[u'][u] += sum_p dAdj[u'][p] h[u][p],  h rebuilt from e_hat in the B fragments (lane = channel u)
#pragma unroll
    for (int u = 0; u < CG_ADJ_MAXW; ++u) {
      const int id = u * nw + wave;
      if (id < MT * MT) {
        const int mt = id / MT, n2 = id - mt * MT, cu = 16 * n2 + l15;
        const bool cok = cu < t.Kc;
        const float gam = cok ? t.bn.gamma[cu] : 0.f, bet = cok ? t.bn.beta[cu] : 0.f;
        const float* ap = cg_tfrag_ptr<0>(sD + 16 * mt * CG_ADJ_PS, CG_ADJ_PS, l15, slot);
        const float* bp = cg_tfrag_ptr<0>(sE + 16 * n2 * CG_ADJ_PS, CG_ADJ_PS, l15, slot);
        for (int k0 = 0; k0 < CG_ADJ_PT; k0 += 16) {
          float av[4], bv[4], keep[4];
          cg_tfrag<0>(ap, CG_ADJ_PS, k0, av); cg_tfrag<0>(bp, CG_ADJ_PS, k0, bv);
          const long long idx = ((long long)b * t.Kc + (cok ? cu : 0)) * g.Pn + p0 + k0 + 4 * slot;
          if (vec) cg_keep4(drop, t.drop_p, seed, t.salt, (unsigned long long)idx, keep);
          else {
#pragma unroll
            for (int s = 0; s < 4; ++s) keep[s] = cg_adj_keep(t, seed, idx + s);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            // positions beyond the tile: dAdj is zero there, the product vanishes
            const float h = cok ? cg_prelu((gam * bv[s] + bet) * keep[s], alpha) : 0.f;
            wacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], h, wacc[u], 0, 0, 0);
          }
        }
      }
    }
    // dh[u][p] = sum_u' W4[u'][u] dAdj[u'][p]
    for (int w = wave; w < MT * 2; w += nw) {
      const int mt = w >> 1, n0 = 32 * (w & 1), n1 = n0 + 16;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      const float* ap = cg_tfrag_ptr<1>(sW + 16 * mt, g.WS, l15, slot);
      const float* bp0 = cg_tfrag_ptr<1>(sD + n0, CG_ADJ_PS, l15, slot);
      const float* bp1 = cg_tfrag_ptr<1>(sD + n1, CG_ADJ_PS, l15, slot);
      for (int k0 = 0; k0 < g.KcM; k0 += 16) {
        float av[4], b0v[4], b1v[4];
        cg_tfrag<1>(ap, g.WS, k0, av); cg_tfrag<1>(bp0, CG_ADJ_PS, k0, b0v); cg_tfrag<1>(bp1, CG_ADJ_PS, k0, b1v);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b0v[s], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b1v[s], c1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u = 16 * mt + 4 * slot + q;
        const bool uok = u < t.Kc;
        const float gam = uok ? t.bn.gamma[u] : 0.f, bet = uok ? t.bn.beta[u] : 0.f;
        float s1 = 0.f, s2 = 0.f, sa = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pp = (h ? n1 : n0) + l15;
          const float dh = h ? c1[q] : c0[q];
          if (uok && pp < np) {
            const long long idx = ((long long)b * t.Kc + u) * g.Pn + p0 + pp;
            const float eh = sE[u * CG_ADJ_PS + pp], keep = cg_adj_keep(t, seed, idx);
            const float upre = (gam * eh + bet) * keep;
            const float gg = (upre > 0.f ? dh : alpha * dh) * keep;
            gb[(long long)u * g.Pn + p0 + pp] = gg;
            s1 += gg; s2 += gg * eh;
            if (!(upre > 0.f)) sa += dh * upre;
          }
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); sa += __shfl_xor(sa, off, 64); }
        if (l15 == 0 && uok) { atomicAdd(&sRed[2 * u], (double)s1); atomicAdd(&sRed[2 * u + 1], (double)s2); atomicAdd(&sRed[2 * g.KcM], (double)sa); }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < 2 * t.Kc; e += CG_ADJ_THREADS) atomicAdd(&t.red[e], sRed[e]);
  if (tid == 0) atomicAdd(&t.red[2 * t.Kc], sRed[2 * g.KcM]);
  float* dW = t.dW4_ws + (long long)(blockIdx.x % CG_ADJ_REPLICAS) * t.Kc * t.Kc;
#pragma unroll
  for (int u = 0; u < CG_ADJ_MAXW; ++u) {
    const int id = u * nw + wave;
    if (id < MT * MT) {
      const int mt = id / MT, n2 = id - mt * MT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * mt + 4 * slot + q, c = 16 * n2 + l15;
        if (r < t.Kc && c < t.Kc) atomicAdd(&dW[r * t.Kc + c], wacc[u][q]);
      }
    }
  }
}

// ======================================================================================================================
// N2: de = BN'(g);  do = W0^T de;  dS[k][a] += do Q[k][b'],  dQ[k][b'] += do S[k][a];  dW0 += de (S x Q)^T
// ======================================================================================================================
__global__ __launch_bounds__(CG_ADJ_THREADS, 2) void cg_adj_n2_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const CgAdjGeom g = cg_adj_geom(t);
  const int b = blockIdx.x;
  float* sS = reinterpret_cast<float*>(cg_dyn_lds);
  float* sQ = sS + g.KcM * g.JS;
  float* sDS = sQ + g.KcM * g.JS;                                 // accumulators
  float* sDQ = sDS + g.KcM * g.JS;
  float* sDE = sDQ + g.KcM * g.JS;                                // [KcM][PS]
  float* sW = sDE + g.KcM * CG_ADJ_PS;                            // [KcM][WS] W0
  float* sK = sW + g.KcM * g.WS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, slot = lane >> 4, nw = CG_ADJ_THREADS / 64;
  cg_adj_tables(t, g, b, sS, sQ);
  for (int e = tid; e < 2 * g.KcM * g.JS + g.KcM * CG_ADJ_PS; e += CG_ADJ_THREADS) sDS[e] = 0.f;
  cg_adj_weight(t.W0, t.Kc, g.KcM, g.WS, sW);
  cg_adj_consts(t, g, sK, true, false);
  const int MT = g.KcM / 16, ntiles = (g.Pn + CG_ADJ_PT - 1) / CG_ADJ_PT;
  cg_f32x4 wacc[CG_ADJ_MAXW];
#pragma unroll
  for (int u = 0; u < CG_ADJ_MAXW; ++u) wacc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ebase = t.e + (long long)b * t.Kc * g.Pn;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int p0 = tile * CG_ADJ_PT, np = min(CG_ADJ_PT, g.Pn - p0);
    __syncthreads();
    cg_adj_stage(t, g, t.g, b, p0, sDE, [&](float4 x4, int c, long long idx, float val[4]) {
      const float* k = sK + 8 * c;
      const float gv[4] = {x4.x, x4.y, x4.z, x4.w};
      if (!t.train) {
#pragma unroll
        for (int j = 0; j < 4; ++j) val[j] = gv[j] * k[2];
        return;
      }
      const long long off = idx >= 0 ? idx : -idx - 1;
      const float* ep = t.e + off;
      const int n = idx >= 0 ? 4 : 1;
      for (int j = 0; j < n; ++j) val[j] = k[2] * (gv[j] - k[4] - (ep[j] - k[0]) * k[1] * k[5]);
    });
    __syncthreads();
    // dW0[u][k] += sum_p de[u][p] o[k][p]: the seed generated along p in the B fragments (lane = slab k)
#pragma unroll
    for (int u = 0; u < CG_ADJ_MAXW; ++u) {
      const int id = u * nw + wave;
      if (id < MT * MT) {
        const int mt = id / MT, n2 = id - mt * MT, kk = 16 * n2 + l15;
        const float* ap = cg_tfrag_ptr<0>(sDE + 16 * mt * CG_ADJ_PS, CG_ADJ_PS, l15, slot);
        for (int k0 = 0; k0 < CG_ADJ_PT; k0 += 16) {
          float av[4];
          cg_tfrag<0>(ap, CG_ADJ_PS, k0, av);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int p = p0 + k0 + 4 * slot + s;
            float o = 0.f;
            if (p < g.Pn) { const int a = (int)cg_adj_div((unsigned)p, g.magicJ); o = sS[kk * g.JS + a] * sQ[kk * g.JS + p - a * t.J]; }
            wacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], o, wacc[u], 0, 0, 0);
          }
        }
      }
    }
    // do[k][p] = sum_u W0[u][k] de[u][p]  ->  dS, dQ
    for (int w = wave; w < MT * 2; w += nw) {
      const int mt = w >> 1, n0 = 32 * (w & 1), n1 = n0 + 16;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      const float* ap = cg_tfrag_ptr<1>(sW + 16 * mt, g.WS, l15, slot);
      const float* bp0 = cg_tfrag_ptr<1>(sDE + n0, CG_ADJ_PS, l15, slot);
      const float* bp1 = cg_tfrag_ptr<1>(sDE + n1, CG_ADJ_PS, l15, slot);
      for (int k0 = 0; k0 < g.KcM; k0 += 16) {
        float av[4], b0v[4], b1v[4];
        cg_tfrag<1>(ap, g.WS, k0, av); cg_tfrag<1>(bp0, CG_ADJ_PS, k0, b0v); cg_tfrag<1>(bp1, CG_ADJ_PS, k0, b1v);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b0v[s], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], b1v[s], c1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pp = (h ? n1 : n0) + l15, p = p0 + pp;
        const bool pok = pp < np;
        const int a = pok ? (int)cg_adj_div((unsigned)p, g.magicJ) : -1, bp = pok ? p - a * t.J : 0;
        // the 16 lanes of a column group hold consecutive positions: distinct b' (no same-address LDS atomics for dQ when
        // J >= 16), runs of equal a (segmented sum, one atomic per run for dS)
        int an[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) an[o] = __shfl_down(a, 1 << o, 16);
        const int aprev = __shfl_up(a, 1, 16);
        const bool head = pok && (l15 == 0 || aprev != a);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = 16 * mt + 4 * slot + q;
          const bool kok = k < t.Kc;                         // uniform over the 16 lanes of a group
          const float d = (pok && kok) ? (h ? c1[q] : c0[q]) : 0.f;
          float v = (pok && kok) ? d * sQ[k * g.JS + bp] : 0.f;
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            const float vn = __shfl_down(v, 1 << o, 16);
            if (l15 + (1 << o) < 16 && an[o] == a) v += vn;
          }
          if (head && kok) atomicAdd(&sDS[k * g.JS + a], v);
          if (pok && kok) atomicAdd(&sDQ[k * g.JS + bp], d * sS[k * g.JS + a]);
        }
      }
    }
  }
  __syncthreads();
  // write dS / dQ back in the layouts of s (V,T) and q (T,V)
  const int V = t.domain == 0 ? t.Kc : t.J, T = t.domain == 0 ? t.J : t.Kc;
  float* dsb = t.ds + (long long)b * V * T;
  float* dqb = t.dq + (long long)b * T * V;
  for (int e = tid; e < t.Kc * t.J; e += CG_ADJ_THREADS) {
    const int k = e / t.J, a = e - k * t.J;
    if (t.domain == 0) { dsb[k * T + a] = sDS[k * g.JS + a]; dqb[a * V + k] = sDQ[k * g.JS + a]; }
    else { dsb[a * T + k] = sDS[k * g.JS + a]; dqb[k * V + a] = sDQ[k * g.JS + a]; }
  }
  float* dW = t.dW0_ws + (long long)(blockIdx.x % CG_ADJ_REPLICAS) * t.Kc * t.Kc;
#pragma unroll
  for (int u = 0; u < CG_ADJ_MAXW; ++u) {
    const int id = u * nw + wave;
    if (id < MT * MT) {
      const int mt = id / MT, n2 = id - mt * MT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * mt + 4 * slot + q, c = 16 * n2 + l15;
        if (r < t.Kc && c < t.Kc) atomicAdd(&dW[r * t.Kc + c], wacc[u][q]);
      }
    }
  }
  (void)ebase;
}

// per-channel parameter gradients + fold of the replicated weight gradients (both towers)
__global__ void cg_adj_finish_kernel(CgAdjTailPair pr) {
  const CgAdjTail& t = pr.t[blockIdx.y];
  const int n = t.Kc * t.Kc;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s0 = 0.f, s4 = 0.f;
    for (int r = 0; r < CG_ADJ_REPLICAS; ++r) { s0 += t.dW0_ws[(long long)r * n + i]; s4 += t.dW4_ws[(long long)r * n + i]; }
    t.dW0[i] = s0; t.dW4[i] = s4;
  }
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < t.Kc; c += blockDim.x) {
      t.dgamma[c] = (float)t.red[2 * c + 1]; t.dbeta[c] = (float)t.red[2 * c];
      if (c == 0) t.dalpha[0] = (float)t.red[2 * t.Kc];
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
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
    const int KcM = (it[i].Kc + 15) & ~15, WS = KcM + 4, JS = it[i].J + 1;
    size_t f;
    if (!bwd && phase == 1) f = (size_t)2 * KcM * JS + (size_t)KcM * WS + 2 + (size_t)4 * KcM;                 // tables, W0, f64 sums
    else if (!bwd) f = (size_t)KcM * CG_ADJ_PS + (size_t)KcM * WS + (size_t)8 * KcM;
    else if (phase == 1) f = (size_t)2 * KcM * CG_ADJ_PS + (size_t)KcM * WS + (size_t)8 * KcM + (size_t)2 * (2 * KcM + 1) + 2;
    else f = (size_t)4 * KcM * JS + (size_t)KcM * CG_ADJ_PS + (size_t)KcM * WS + (size_t)8 * KcM;
    best = f > best ? f : best;
  }
  return best * sizeof(float) + 16;
}

extern "C" long long cg_map2adj_tail_ws_floats(int Kc) { return (long long)2 * CG_ADJ_REPLICAS * Kc * Kc; }

// include/cistgcn_hip.h : cg_map2adj_tail_fwd (phases 1, 2) / cg_map2adj_tail_bwd (phases 1, 2)
extern "C" int cg_map2adj_tail_fwd(const CgAdjTail* items, int n, int phase, void* stream_) {
  int st = cg_adj_check(items, n);
  if (st != CG_OK) return st;
  CgAdjTailPair pr;
  pr.n = n; pr.pad = 0;
  for (int i = 0; i < n; ++i) {
    pr.t[i] = items[i];
    if (phase == 1 && items[i].train && !items[i].bn.stats) return CG_EARG;
    if (phase == 2 && !items[i].adj) return CG_EARG;
  }
  const size_t lds = cg_adj_lds(items, n, phase, false);
  dim3 grid((unsigned)items[0].B, (unsigned)n), block(CG_ADJ_THREADS);
  hipStream_t stream = (hipStream_t)stream_;
  if (phase == 1) {
    hipError_t e = hipFuncSetAttribute((const void*)cg_adj_m1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_adj_m1_kernel, grid, block, lds, stream, pr);
  } else if (phase == 2) {
    hipError_t e = hipFuncSetAttribute((const void*)cg_adj_m2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_adj_m2_kernel, grid, block, lds, stream, pr);
  } else return CG_EARG;
  return cg_launch_status();
}

extern "C" int cg_map2adj_tail_bwd(const CgAdjTail* items, int n, int phase, void* stream_) {
  int st = cg_adj_check(items, n);
  if (st != CG_OK) return st;
  CgAdjTailPair pr;
  pr.n = n; pr.pad = 0;
  for (int i = 0; i < n; ++i) {
    const CgAdjTail& t = items[i];
    pr.t[i] = t;
    if (!t.dadj || !t.g || !t.red || !t.dW0_ws || !t.dW4_ws) return CG_EARG;
    if (phase == 2 && (!t.ds || !t.dq || !t.dW0 || !t.dW4 || !t.dgamma || !t.dbeta || !t.dalpha)) return CG_EARG;
  }
  const size_t lds = cg_adj_lds(items, n, phase, true);
  dim3 grid((unsigned)items[0].B, (unsigned)n), block(CG_ADJ_THREADS);
  hipStream_t stream = (hipStream_t)stream_;
  if (phase == 1) {
    hipError_t e = hipFuncSetAttribute((const void*)cg_adj_n1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_adj_n1_kernel, grid, block, lds, stream, pr);
  } else if (phase == 2) {
    hipError_t e = hipFuncSetAttribute((const void*)cg_adj_n2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_adj_n2_kernel, grid, block, lds, stream, pr);
    st = cg_launch_status();
    if (st != CG_OK) return st;
    hipLaunchKernelGGL(cg_adj_finish_kernel, dim3(8, (unsigned)n), dim3(256), 0, stream, pr);
  } else return CG_EARG;
  return cg_launch_status();
}
