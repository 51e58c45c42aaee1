// Matrix-core kernels of the fused ST-GCN stage (reference: Domain_GCNN_layer.forward, CISTGCN.py:265-266, with
// ConvTemporalGraphical.forward :122-124 and the 1x1 `tcn` convolution :229-234) for wide layers: every product of the
// stage runs on v_mfma_f32_16x16x4_f32 (exact f32 fma chains).  Persistent 256-thread workgroups, two per CU (LDS image
// <= 80 KiB), each walking a contiguous range of tiles; the next tile's operands are fetched into registers while the
// current tile's last phase runs, and the second workgroup of the CU computes while the first one waits for memory
// (measured with the phases switched off, round-2 ablation builds, profiles/r02_ablations.txt: loads, compute and stores of a tile cost about the same,
// serialised they were 3x the compute time).
//
// Backward, per tile = (sample b, GT groups; a group is one joint in the "space" domain, one frame in the "time" domain):
//   P1  G  = X . A                 per group   (Cin x J)(J x J)          graph product, recomputed
//   P5  dW += dY . G^T             whole tile  (Cout x P)(P x Cin)       accumulators live in registers across all tiles
//   P4  dG = W^T . dY              whole tile  (Cin x Cout)(Cout x P)
//   P2  dX = dG . A^T              per group   (Cin x J)(J x J)
//   P3  dA = X^T . dG              per group   (J x Cin)(Cin x J)
// LDS images (floats), zero-filled once; pads are never written with anything but exact zeros, so that 16-wide fragment
// reads beyond the data contribute nothing:
//   sX, sDY, sBuf [channel][RS]    position p = grp*Js + j inside a row (Js = J up to 4); sBuf holds G, then dG
//   sA [GT][Jr][Jsa]               adjacency slab, Jr = J up to 16 rows, Jsa = Jr + 4 columns
//   sW [CoM][WS]                   mixing weights (when the image fits; otherwise the dG product reads W through L1/L2)
// Row strides are == 4 (mod 8): the k-strided fragment reads (one float per lane per MFMA, `cg_frag<1>`) are bank-conflict
// free and the k-contiguous ones (one ds_read_b128 per four MFMAs, `cg_frag<0>`) see a 2-way conflict (tools: bank check in
// DESIGN.md).  Inside a 16-wide k chunk the four MFMA steps take k = 4*slot + step (slot = lane / 16): the same permutation
// on both operands, so a lane's four values are contiguous.
#include "cg_common.h"
#include "stgcn_domain.h"
#include <stdlib.h>

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

typedef float cg_f32x4 __attribute__((vector_size(16)));

__device__ __forceinline__ unsigned cg_fastdiv(unsigned n, unsigned magic) {    // n / d for 2 <= d <= 64, n < 2^24; magic = ceil(2^32 / d), 0 for d = 1
  return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n;
}

// One operand fragment set for a 16-wide k chunk: v[s] is the lane's value for MFMA step s.
//   KIND 0: image is [index][k] (k contiguous): p points at [index of this lane][4*slot]; one 16-byte read
//   KIND 1: image is [k][index] (k strided by rs): p points at [4*slot][index of this lane]; four scalar reads
template <int KIND>
__device__ __forceinline__ void cg_frag(const float* __restrict__ p, int rs, int k0, float v[4]) {
  if (KIND == 0) {
    const float4 t = *reinterpret_cast<const float4*>(p + k0);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    const float* q = p + (long long)k0 * rs;
    v[0] = q[0]; v[1] = q[rs]; v[2] = q[2 * rs]; v[3] = q[3 * rs];
  }
}

template <int KIND>
__device__ __forceinline__ const float* cg_frag_ptr(const float* base, int rs, int l15, int slot) {
  return KIND == 0 ? base + l15 * rs + 4 * slot : base + l15 + 4 * slot * rs;
}

// C0 += A . B0, C1 += A . B1 over k in [0, K): two 16x16 output tiles that share the A fragments (two independent
// accumulators also cover the 40-cycle dependent latency of the 32-cycle MFMA).
template <int AK, int BK>
__device__ __forceinline__ void cg_mma_pair(const float* a, int a_rs, const float* b0, const float* b1, int b_rs, int K,
                                            cg_f32x4& c0, cg_f32x4& c1) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, slot = lane >> 4;
  const float* ap = cg_frag_ptr<AK>(a, a_rs, l15, slot);
  const float* bp0 = cg_frag_ptr<BK>(b0, b_rs, l15, slot);
  const float* bp1 = cg_frag_ptr<BK>(b1, b_rs, l15, slot);
  int k0 = 0;
  for (; k0 + 16 <= K; k0 += 16) {
    float av[4], bv0[4], bv1[4];
    cg_frag<AK>(ap, a_rs, k0, av); cg_frag<BK>(bp0, b_rs, k0, bv0); cg_frag<BK>(bp1, b_rs, k0, bv1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv0[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv1[s], c1, 0, 0, 0);
    }
  }
  const int rem = K - k0;                      // step s of the last chunk covers k0 + 4*slot + s: needed while s < rem
  if (rem > 0) {
    float av[4], bv0[4], bv1[4];
    cg_frag<AK>(ap, a_rs, k0, av); cg_frag<BK>(bp0, b_rs, k0, bv0); cg_frag<BK>(bp1, b_rs, k0, bv1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (s < rem) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv0[s], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv1[s], c1, 0, 0, 0);
      }
  }
}

// magic-number division helpers live in the geometry
__device__ __forceinline__ int cg_domm_off(const CgDomM& g, int domain, int grp_abs, int j) {
  return domain == 1 ? grp_abs * g.V + j : j * g.V + grp_abs;
}
template <int DOMAIN>
__device__ __forceinline__ int cg_domm_tile_base(const CgDomM& g, int g0) { return DOMAIN == 1 ? g0 * g.V : g0; }

#define CG_DOMM_THREADS 256
#define CG_DOMM_PF 16          // register slots per thread for one prefetched image (PF * 256 elements)

// A [C][GT][J] tile of a contiguous (B,C,T,V) tensor: element e = tid + i*256 of the tile is (c, grp, j).  Its offset in HBM
// relative to the tile's first element and its offset in the [C][RS] LDS image do not depend on the tile, so every thread
// derives them ONCE (the index divisions cost more than the loads) and keeps them in registers:
//   gof[i]  offset from (sample base + tile base), lof[i] = LDS offset | grp << 24, or -1 past the end of the tile
template <int DOMAIN>
__device__ __forceinline__ void cg_domm_slots(const CgDomM& g, int C, int gof[CG_DOMM_PF], int lof[CG_DOMM_PF]) {
  const int TV = g.T * g.V, run = g.GT * g.J, n = C * run;
#pragma unroll
  for (int i = 0; i < CG_DOMM_PF; ++i) {
    const int e = threadIdx.x + i * CG_DOMM_THREADS;
    gof[i] = 0; lof[i] = 0;
    if (e < n) {
      const int c = (int)cg_fastdiv((unsigned)e, g.magicRun), r = e - c * run;
      int grp, j;
      if (DOMAIN == 1) { grp = (int)cg_fastdiv((unsigned)r, g.magicJ); j = r - grp * g.J; }        // j fastest: contiguous in HBM
      else { j = (int)cg_fastdiv((unsigned)r, g.magicGT); grp = r - j * g.GT; }                      // joint fastest
      gof[i] = c * TV + (DOMAIN == 1 ? grp * g.V + j : j * g.V + grp);
      lof[i] = (c * g.RS + grp * g.Js + j) | (grp << 24);
    }
  }
}

// load: HBM -> registers (all loads of a thread are issued before anything waits for them); `tile` points at the first
// element of the tile; groups >= ng (last tile of a sample) read as zero
// (the tables are built for max(Cin, Cout) channels; `n` = elements of THIS image)
__device__ __forceinline__ void cg_domm_tile_load(const float* __restrict__ tile, int n, int ng, const int gof[CG_DOMM_PF], const int lof[CG_DOMM_PF],
                                                  float v[CG_DOMM_PF]) {
#pragma unroll
  for (int i = 0; i < CG_DOMM_PF; ++i)
    v[i] = ((int)threadIdx.x + i * CG_DOMM_THREADS < n && (lof[i] >> 24) < ng) ? tile[gof[i]] : 0.f;
}

__device__ __forceinline__ void cg_domm_tile_store(int n, const int lof[CG_DOMM_PF], const float v[CG_DOMM_PF], float* dst) {
#pragma unroll
  for (int i = 0; i < CG_DOMM_PF; ++i)
    if ((int)threadIdx.x + i * CG_DOMM_THREADS < n) dst[lof[i] & 0xFFFFFF] = v[i];
}

// adjacency slabs of a tile: contiguous (ng, J, J) in HBM -> sA[grp][j][o] (rows of Jsa floats)
__device__ __forceinline__ void cg_domm_adj_store(const CgDomM& g, const float v[CG_DOMM_PF], float* sA) {
  const int n = g.GT * g.J * g.J;
#pragma unroll
  for (int i = 0; i < CG_DOMM_PF; ++i) {
    const int e = threadIdx.x + i * CG_DOMM_THREADS;
    if (e < n) {
      const int r = (int)cg_fastdiv((unsigned)e, g.magicJ), o = e - r * g.J;      // r = grp*J + j
      const int grp = (int)cg_fastdiv((unsigned)r, g.magicJ), j = r - grp * g.J;
      sA[(grp * g.Jr + j) * g.Jsa + o] = v[i];
    }
  }
}

__device__ __forceinline__ void cg_domm_adj_load(const CgDomM& g, const float* __restrict__ ab, int ng, float v[CG_DOMM_PF]) {
  const int n = ng * g.J * g.J;
#pragma unroll
  for (int i = 0; i < CG_DOMM_PF; ++i) {
    const int e = threadIdx.x + i * CG_DOMM_THREADS;
    v[i] = e < n ? ab[e] : 0.f;
  }
}

// XCD-aware order (placement affects speed only): hardware blocks b and b+8 share an L2; every XCD walks a contiguous
// range of workgroups and every workgroup a contiguous range of tiles, so the tiles of one sample - which touch the same
// cache lines of x, dy and dx in the space domain - meet in one L2, close in time
__device__ __forceinline__ int cg_domm_wg(const CgDomM& g) {
  const int chunk = gridDim.x / 8;
  const int wg = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  return wg < (g.total + g.per - 1) / g.per ? wg : -1;
}

// P1 of both directions: G[ci][grp, o] = sum_j X[ci][grp, j] A[grp][j][o]  -> sBuf
__device__ __forceinline__ void cg_domm_graph_product(const CgDomM& g, const float* sX, const float* sA, float* sBuf) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = CG_DOMM_THREADS / 64, l15 = lane & 15, slot = lane >> 4;
  const int MTi = g.CiM / 16, NT = g.Jr / 16, NP = (NT + 1) / 2;
  for (int w = wave; w < g.GT * MTi * NP; w += nw) {
    const int r = (int)cg_fastdiv((unsigned)w, g.magicNP), np = w - r * NP;
    const int grp = (int)cg_fastdiv((unsigned)r, g.magicMTi), mt = r - grp * MTi;
    const int n0 = 32 * np, n1 = (n0 + 16 < g.Jr) ? n0 + 16 : n0;
    cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    const float* as = sA + grp * g.Jr * g.Jsa;
    cg_mma_pair<0, 1>(sX + 16 * mt * g.RS + grp * g.Js, g.RS, as + n0, as + n1, g.Jsa, g.J, c0, c1);
    float* out = sBuf + (16 * mt + 4 * slot) * g.RS + grp * g.Js + l15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (n0 + l15 < g.J) out[q * g.RS + n0] = c0[q];
      if (n1 != n0 && n1 + l15 < g.J) out[q * g.RS + n1] = c1[q];
    }
  }
}

template <int DOMAIN, int MAXW>
__global__ __launch_bounds__(CG_DOMM_THREADS, 2) void cg_stgcn_domain_bwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                       const float* __restrict__ W, const float* __restrict__ dy,
                                                                       float* __restrict__ dx, float* __restrict__ dadj,
                                                                       float* __restrict__ ws, int replicas, CgDomM g) {
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);
  float* sDY = sX + g.CiM * g.RS;
  float* sBuf = sDY + g.CoM * g.RS;
  float* sA = sBuf + g.CiM * g.RS;
  float* sW = sA + g.GT * g.Jr * g.Jsa;                   // only when !g.w_global
  const int lds_floats = (2 * g.CiM + g.CoM) * g.RS + g.GT * g.Jr * g.Jsa + (g.w_global ? 0 : g.CoM * g.WS);

  const int tid = threadIdx.x, nt = CG_DOMM_THREADS;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6, l15 = lane & 15, slot = lane >> 4;
  const int wg = cg_domm_wg(g);
  if (wg < 0) return;

  for (int e = tid; e < lds_floats; e += nt) sX[e] = 0.f;
  __syncthreads();
  if (!g.w_global)
    for (int e = tid; e < g.Cout * g.Cin; e += nt) {
      const int co = e / g.Cin, ci = e - co * g.Cin;
      sW[co * g.WS + ci] = W[e];
    }
  const float* wimg = g.w_global ? W : sW;                // [co][ci], row stride:
  const int wrs = g.w_global ? g.Cin : g.WS;

  const int MTi = g.CiM / 16, MTo = g.CoM / 16, NT = g.Jr / 16, NP = (NT + 1) / 2;
  const int NTp = (g.GT * g.Js + 15) / 16, NPp = (NTp + 1) / 2;
  const long long TV = (long long)g.T * g.V;

  cg_f32x4 wacc[MAXW];
#pragma unroll
  for (int u = 0; u < MAXW; ++u) wacc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  float bacc[MAXW];
#pragma unroll
  for (int u = 0; u < MAXW; ++u) bacc[u] = 0.f;

  float px[CG_DOMM_PF], pdy[CG_DOMM_PF], pa[CG_DOMM_PF];  // the next tile, in flight
  int gof[CG_DOMM_PF], lof[CG_DOMM_PF];
  cg_domm_slots<DOMAIN>(g, max(g.Cin, g.Cout), gof, lof);
  const int nx = g.Cin * g.GT * g.J, ny = g.Cout * g.GT * g.J;
  const int lid0 = wg * g.per;
  {
    const int b = lid0 / g.ntiles, tile = lid0 - b * g.ntiles, g0 = tile * g.GT, ng = min(g.GT, g.NG - g0);
    cg_domm_tile_load(x + (long long)b * g.Cin * TV + cg_domm_tile_base<DOMAIN>(g, g0), nx, ng, gof, lof, px);
    cg_domm_tile_load(dy + (long long)b * g.Cout * TV + cg_domm_tile_base<DOMAIN>(g, g0), ny, ng, gof, lof, pdy);
    cg_domm_adj_load(g, adj + ((long long)b * g.NG + g0) * g.J * g.J, ng, pa);
  }

  for (int it = 0; it < g.per; ++it) {
    const int lid = lid0 + it;
    if (lid >= g.total) break;                   // uniform across the workgroup
    const int b = lid / g.ntiles, tile = lid - b * g.ntiles;
    const int g0 = tile * g.GT, ng = min(g.GT, g.NG - g0);
    __syncthreads();                             // previous tile fully consumed
    cg_domm_tile_store(nx, lof, px, sX);
    cg_domm_tile_store(ny, lof, pdy, sDY);
    cg_domm_adj_store(g, pa, sA);
    __syncthreads();

    cg_domm_graph_product(g, sX, sA, sBuf);      // P1
    __syncthreads();

    // P5: dW[co][ci] += sum_p dY[co][p] G[ci][p]   (+ db[co] += sum_p dY[co][p] from the A fragments of the ci-tile-0 owners)
#pragma unroll
    for (int u = 0; u < MAXW; ++u) {
      const int id = u * nw + wave;
      if (id < MTo * MTi) {
        const int mt = id / MTi, ntl = id - mt * MTi;
        const float* ap = cg_frag_ptr<0>(sDY + 16 * mt * g.RS, g.RS, l15, slot);
        const float* bp = cg_frag_ptr<0>(sBuf + 16 * ntl * g.RS, g.RS, l15, slot);
        const int K = g.GT * g.Js;               // multiple of 4: every chunk runs its four steps
        float bs = 0.f;
        for (int k0 = 0; k0 < K; k0 += 16) {
          float av[4], bv[4];
          cg_frag<0>(ap, g.RS, k0, av); cg_frag<0>(bp, g.RS, k0, bv);
          bs += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
          for (int s = 0; s < 4; ++s) wacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], wacc[u], 0, 0, 0);
        }
        if (ntl == 0) bacc[u] += bs;
      }
    }
    __syncthreads();

    // P4: dG[ci][p] = sum_co W[co][ci] dY[co][p]  -> sBuf (G is dead)
    for (int w = wave; w < MTi * NPp; w += nw) {
      const int mt = (int)cg_fastdiv((unsigned)w, g.magicNPp), np = w - mt * NPp;
      const int n0 = 32 * np, n1 = (n0 + 16 < 16 * NTp) ? n0 + 16 : n0;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      cg_mma_pair<1, 1>(wimg + 16 * mt, wrs, sDY + n0, sDY + n1, g.RS, g.Cout, c0, c1);
      float* out = sBuf + (16 * mt + 4 * slot) * g.RS + l15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        out[q * g.RS + n0] = c0[q];
        if (n1 != n0) out[q * g.RS + n1] = c1[q];
      }
    }
    __syncthreads();

    // the next tile's operands start their way from HBM now and land while P2 / P3 run
    if (it + 1 < g.per && lid + 1 < g.total) {
      const int l2 = lid + 1, b2 = l2 / g.ntiles, t2 = l2 - b2 * g.ntiles, h0 = t2 * g.GT, nh = min(g.GT, g.NG - h0);
      cg_domm_tile_load(x + (long long)b2 * g.Cin * TV + cg_domm_tile_base<DOMAIN>(g, h0), nx, nh, gof, lof, px);
      cg_domm_tile_load(dy + (long long)b2 * g.Cout * TV + cg_domm_tile_base<DOMAIN>(g, h0), ny, nh, gof, lof, pdy);
      cg_domm_adj_load(g, adj + ((long long)b2 * g.NG + h0) * g.J * g.J, nh, pa);
    }

    // P2: dX[ci][grp, j] = sum_o dG[ci][grp, o] A[grp][j][o]      P3: dA[grp][j][o] = sum_ci X[ci][grp, j] dG[ci][grp, o]
    const int n2 = g.GT * MTi * NP, n3 = g.GT * NT * NP;
    float* dxb = dx + (long long)b * g.Cin * TV;
    float* dab = dadj + ((long long)b * g.NG + g0) * g.J * g.J;
    for (int w = wave; w < n2 + n3; w += nw) {
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      if (w < n2) {
        const int r = (int)cg_fastdiv((unsigned)w, g.magicNP), np = w - r * NP;
        const int grp = (int)cg_fastdiv((unsigned)r, g.magicMTi), mt = r - grp * MTi;
        const int n0 = 32 * np, n1 = (n0 + 16 < g.Jr) ? n0 + 16 : n0;
        if (grp >= ng) continue;
        const float* as = sA + grp * g.Jr * g.Jsa;
        cg_mma_pair<0, 0>(sBuf + 16 * mt * g.RS + grp * g.Js, g.RS, as + n0 * g.Jsa, as + n1 * g.Jsa, g.Jsa, g.J, c0, c1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ci = 16 * mt + 4 * slot + q;
          if (ci >= g.Cin) continue;
          const int j0 = n0 + l15, j1 = n1 + l15;
          if (j0 < g.J) dxb[ci * TV + cg_domm_off(g, DOMAIN, g0 + grp, j0)] = c0[q];
          if (n1 != n0 && j1 < g.J) dxb[ci * TV + cg_domm_off(g, DOMAIN, g0 + grp, j1)] = c1[q];
        }
      } else {
        const int v = w - n2;
        const int r = (int)cg_fastdiv((unsigned)v, g.magicNP), np = v - r * NP;
        const int grp = (int)cg_fastdiv((unsigned)r, g.magicNT), mt = r - grp * NT;
        const int n0 = 32 * np, n1 = (n0 + 16 < g.Jr) ? n0 + 16 : n0;
        if (grp >= ng) continue;
        cg_mma_pair<1, 1>(sX + grp * g.Js + 16 * mt, g.RS, sBuf + grp * g.Js + n0, sBuf + grp * g.Js + n1, g.RS, g.Cin, c0, c1);
        float* row = dab + ((long long)grp * g.J) * g.J;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = 16 * mt + 4 * slot + q;
          if (j >= g.J) continue;
          const int o0 = n0 + l15, o1 = n1 + l15;
          if (o0 < g.J) row[(long long)j * g.J + o0] = c0[q];
          if (n1 != n0 && o1 < g.J) row[(long long)j * g.J + o1] = c1[q];
        }
      }
    }
  }

  // weight / bias gradients: one fp32 atomic per entry and workgroup into one of `replicas` copies (folded afterwards)
  float* dW = ws + (long long)(blockIdx.x % replicas) * (g.Cout * g.Cin + g.Cout);
  float* db = dW + g.Cout * g.Cin;
#pragma unroll
  for (int u = 0; u < MAXW; ++u) {
    const int id = u * nw + wave;
    if (id < MTo * MTi) {
      const int mt = id / MTi, ntl = id - mt * MTi;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = 16 * mt + 4 * slot + q, ci = 16 * ntl + l15;
        if (co < g.Cout && ci < g.Cin) atomicAdd(&dW[co * g.Cin + ci], wacc[u][q]);
      }
      if (ntl == 0) {
        float s = bacc[u];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (slot == 0 && 16 * mt + l15 < g.Cout) atomicAdd(&db[16 * mt + l15], s);
      }
    }
  }
}

// Forward on the matrix cores, same images and staging as the backward:
//   P1  G = X . A  per group -> sBuf          PY  Y = W . G + bias  whole tile -> HBM  (+ f64 channel sums of Y for the
// train-mode BatchNorm that follows, CISTGCN.py:235)
template <int DOMAIN>
__global__ __launch_bounds__(CG_DOMM_THREADS, 2) void cg_stgcn_domain_fwd_mfma2_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                                        float* __restrict__ y, double* __restrict__ ystats, CgDomM g) {
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);
  float* sBuf = sX + g.CiM * g.RS;
  float* sA = sBuf + g.CiM * g.RS;
  float* sW = sA + g.GT * g.Jr * g.Jsa;
  double* sStat = reinterpret_cast<double*>(sW + g.CoM * g.WS);          // [CoM][2], 8-byte aligned: every term is a multiple of 4 floats
  const int lds_floats = 2 * g.CiM * g.RS + g.GT * g.Jr * g.Jsa + g.CoM * g.WS + 4 * g.CoM;

  const int tid = threadIdx.x, nt = CG_DOMM_THREADS;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6, l15 = lane & 15, slot = lane >> 4;
  const int wg = cg_domm_wg(g);
  if (wg < 0) return;

  for (int e = tid; e < lds_floats; e += nt) sX[e] = 0.f;
  __syncthreads();
  for (int e = tid; e < g.Cout * g.Cin; e += nt) {
    const int co = e / g.Cin, ci = e - co * g.Cin;
    sW[co * g.WS + ci] = W[e];
  }

  const int MTo = g.CoM / 16;
  const int NTp = (g.GT * g.Js + 15) / 16, NPp = (NTp + 1) / 2;
  const long long TV = (long long)g.T * g.V;

  float px[CG_DOMM_PF], pa[CG_DOMM_PF];
  int gof[CG_DOMM_PF], lof[CG_DOMM_PF];
  cg_domm_slots<DOMAIN>(g, g.Cin, gof, lof);
  const int nx = g.Cin * g.GT * g.J;
  const int lid0 = wg * g.per;
  {
    const int b = lid0 / g.ntiles, tile = lid0 - b * g.ntiles, g0 = tile * g.GT, ng = min(g.GT, g.NG - g0);
    cg_domm_tile_load(x + (long long)b * g.Cin * TV + cg_domm_tile_base<DOMAIN>(g, g0), nx, ng, gof, lof, px);
    cg_domm_adj_load(g, adj + ((long long)b * g.NG + g0) * g.J * g.J, ng, pa);
  }
  for (int it = 0; it < g.per; ++it) {
    const int lid = lid0 + it;
    if (lid >= g.total) break;                   // uniform across the workgroup
    const int b = lid / g.ntiles, tile = lid - b * g.ntiles;
    const int g0 = tile * g.GT, ng = min(g.GT, g.NG - g0);
    __syncthreads();                             // previous tile fully consumed
    cg_domm_tile_store(nx, lof, px, sX);
    cg_domm_adj_store(g, pa, sA);
    __syncthreads();
    cg_domm_graph_product(g, sX, sA, sBuf);      // P1
    __syncthreads();
    if (it + 1 < g.per && lid + 1 < g.total) {   // next tile on its way while PY runs
      const int l2 = lid + 1, b2 = l2 / g.ntiles, t2 = l2 - b2 * g.ntiles, h0 = t2 * g.GT, nh = min(g.GT, g.NG - h0);
      cg_domm_tile_load(x + (long long)b2 * g.Cin * TV + cg_domm_tile_base<DOMAIN>(g, h0), nx, nh, gof, lof, px);
      cg_domm_adj_load(g, adj + ((long long)b2 * g.NG + h0) * g.J * g.J, nh, pa);
    }
    float* yb = y + (long long)b * g.Cout * TV;
    for (int w = wave; w < MTo * NPp; w += nw) {                           // PY
      const int mt = (int)cg_fastdiv((unsigned)w, g.magicNPp), np = w - mt * NPp;
      const int n0 = 32 * np, n1 = (n0 + 16 < 16 * NTp) ? n0 + 16 : n0;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      cg_mma_pair<0, 1>(sW + 16 * mt * g.WS, g.WS, sBuf + n0, sBuf + n1, g.RS, g.Cin, c0, c1);
      const int p0 = n0 + l15, p1 = n1 + l15;
      const int gr0 = (int)cg_fastdiv((unsigned)p0, g.magicJs), o0 = p0 - gr0 * g.Js;
      const int gr1 = (int)cg_fastdiv((unsigned)p1, g.magicJs), o1 = p1 - gr1 * g.Js;
      const bool ok0 = gr0 < ng && o0 < g.J, ok1 = n1 != n0 && gr1 < ng && o1 < g.J;
      const int a0 = ok0 ? cg_domm_off(g, DOMAIN, g0 + gr0, o0) : 0, a1 = ok1 ? cg_domm_off(g, DOMAIN, g0 + gr1, o1) : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = 16 * mt + 4 * slot + q;
        const bool cok = co < g.Cout;
        const float bv = (cok && bias) ? bias[co] : 0.f;
        const float v0 = (ok0 && cok) ? c0[q] + bv : 0.f, v1 = (ok1 && cok) ? c1[q] + bv : 0.f;
        if (ok0 && cok) yb[co * TV + a0] = v0;
        if (ok1 && cok) yb[co * TV + a1] = v1;
        if (ystats) {          // the 16 lanes of a slot hold the same channel: reduce them before touching LDS
          float s1 = v0 + v1, s2 = v0 * v0 + v1 * v1;
          s1 = cg_row16_sum(s1); s2 = cg_row16_sum(s2);
          if (l15 == 0 && cok) { atomicAdd(&sStat[2 * co], (double)s1); atomicAdd(&sStat[2 * co + 1], (double)s2); }
        }
      }
    }
  }
  if (ystats) {
    __syncthreads();
    double* rep = ystats + (long long)(blockIdx.x % CG_STAT_REPLICAS) * 2 * g.Cout;
    for (int e = tid; e < 2 * g.Cout; e += nt) atomicAdd(&rep[e], sStat[e]);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_up(int v, int m) { return (v + m - 1) / m * m; }
static int cg_up_4mod8(int v) { int r = cg_up(v, 4); return (r % 8 == 4) ? r : r + 4; }
static unsigned cg_magic(int d) { return d > 1 ? (unsigned)((0x100000000ULL + d - 1) / d) : 0u; }

size_t cg_domm_lds_bytes(const CgDomM& g, bool bwd) {
  if (bwd) return ((size_t)(2 * g.CiM + g.CoM) * g.RS + (size_t)g.GT * g.Jr * g.Jsa + (g.w_global ? 0 : (size_t)g.CoM * g.WS)) * sizeof(float);
  return ((size_t)2 * g.CiM * g.RS + (size_t)g.GT * g.Jr * g.Jsa + (size_t)g.CoM * g.WS + (size_t)4 * g.CoM) * sizeof(float);
}

static int cg_domm_rs(const CgDomM& g, int gt) {
  const int a = (gt - 1) * g.Js + g.Jr, b = cg_up(gt * g.Js, 16);
  return cg_up_4mod8(a > b ? a : b);
}

int cg_domm_geom(CgDomM& g, int B, int Cin, int Cout, int T, int V, int domain, bool bwd) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || (domain != 0 && domain != 1)) return CG_ESHAPE;
  g.B = B; g.Cin = Cin; g.Cout = Cout; g.T = T; g.V = V;
  g.NG = domain == 1 ? T : V;
  g.J = domain == 1 ? V : T;
  if (g.J > 64) return CG_ESHAPE;
  g.Js = cg_up(g.J, 4);
  g.Jr = cg_up(g.J, 16);
  g.Jsa = g.Jr + 4;
  g.CiM = cg_up(Cin, 16); g.CoM = cg_up(Cout, 16);
  g.WS = g.CiM + 4;
  g.magicJ = cg_magic(g.J);
  g.magicJs = cg_magic(g.Js);
  g.dbg = 0;
  // two workgroups per CU: the image must fit 80 KiB.  The backward drops the weight image first (the dG product then
  // reads W through L1/L2, which needs whole 16-wide fragments: channel counts that are multiples of 16)
  const size_t limit = 80 * 1024;
  const int cap = CG_DOMM_PF * CG_DOMM_THREADS;          // elements of one prefetched image
  const int cmax = Cin > Cout ? Cin : Cout;
  int best = 0;
  for (int pass = 0; pass < 2 && !best; ++pass) {
    g.w_global = (bwd && pass == 1) ? 1 : 0;
    if (g.w_global && ((Cin & 15) || (Cout & 15))) break;
    if (!bwd && pass == 1) break;
    for (int gt = g.NG; gt >= 1; --gt) {
      g.GT = gt;
      g.RS = cg_domm_rs(g, gt);
      if (cg_domm_lds_bytes(g, bwd) > limit) continue;
      if ((long long)cmax * gt * g.J > cap || gt * g.J * g.J > cap) continue;
      const int ntiles = (g.NG + gt - 1) / gt;
      const double waste = (double)ntiles * gt / g.NG;
      const long long total = (long long)B * ntiles, want = (long long)B * g.NG < 1024 ? (long long)B * g.NG : 1024;
      if (gt > 1 && (waste > 1.15 || total < want)) continue;
      best = gt;
      break;
    }
  }
  if (best == 0) return CG_ESHAPE;
  g.GT = best;
  g.RS = cg_domm_rs(g, best);
  g.magicGT = cg_magic(g.GT);
  g.magicRun = cg_magic(g.GT * g.J);
  g.magicNP = cg_magic((g.Jr / 16 + 1) / 2);
  g.magicMTi = cg_magic(g.CiM / 16);
  g.magicNT = cg_magic(g.Jr / 16);
  g.magicNPp = cg_magic(((g.GT * g.Js + 15) / 16 + 1) / 2);
  g.ntiles = (g.NG + best - 1) / best;
  const long long total = (long long)B * g.ntiles;
  if (total > 2147483647LL) return CG_ESHAPE;
  g.total = (int)total;
  g.per = (int)((total + 511) / 512);            // two workgroups per CU, a contiguous range of tiles each
  return CG_OK;
}

// launches the backward kernel; `ws` holds `replicas` zeroed copies of (dW, db)
int cg_domm_bwd_launch(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj, float* ws,
                       int replicas, int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream) {
  CgDomM g;
  int st = cg_domm_geom(g, B, Cin, Cout, T, V, domain, true);
  if (st != CG_OK) return st;
  const int tilesW = (g.CoM / 16) * (g.CiM / 16);
  if (tilesW > 4 * 4) return CG_ESHAPE;          // 4 waves x 4 register tiles (Cin, Cout <= 64); wider layers: VALU kernel
  const size_t lds = cg_domm_lds_bytes(g, true);
  const long long nwg = ((long long)g.total + g.per - 1) / g.per;
  dim3 grid((unsigned)(((nwg + 7) / 8) * 8)), block(CG_DOMM_THREADS);
  const int maxw = (tilesW + 3) / 4;
#define CG_DOMM_LAUNCH(D, M)                                                                                              \
  do {                                                                                                                    \
    hipError_t e = cg_lds_limit((const void*)cg_stgcn_domain_bwd_mfma_kernel<D, M>,                                \
                                       lds);                             \
    if (e != hipSuccess) return (int)e;                                                                                   \
    hipLaunchKernelGGL((cg_stgcn_domain_bwd_mfma_kernel<D, M>), grid, block, lds, stream, x, adj, W, dy, dx, dadj, ws,    \
                       replicas, g);                                                                                      \
  } while (0)
  if (domain == 0) {
    if (maxw <= 1) CG_DOMM_LAUNCH(0, 1); else CG_DOMM_LAUNCH(0, 4);
  } else {
    if (maxw <= 1) CG_DOMM_LAUNCH(1, 1); else CG_DOMM_LAUNCH(1, 4);
  }
#undef CG_DOMM_LAUNCH
  return cg_launch_status();
}

int cg_domm_fwd_launch(const float* x, const float* adj, const float* W, const float* bias, float* y, double* ystats,
                       int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream) {
  CgDomM g;
  int st = cg_domm_geom(g, B, Cin, Cout, T, V, domain, false);
  if (st != CG_OK) return st;
  const size_t lds = cg_domm_lds_bytes(g, false);
  const long long nwg = ((long long)g.total + g.per - 1) / g.per;
  dim3 grid((unsigned)(((nwg + 7) / 8) * 8)), block(CG_DOMM_THREADS);
  const void* fn = domain == 0 ? (const void*)cg_stgcn_domain_fwd_mfma2_kernel<0> : (const void*)cg_stgcn_domain_fwd_mfma2_kernel<1>;
  hipError_t e = cg_lds_limit(fn, lds);
  if (e != hipSuccess) return (int)e;
  if (domain == 0) hipLaunchKernelGGL(cg_stgcn_domain_fwd_mfma2_kernel<0>, grid, block, lds, stream, x, adj, W, bias, y, ystats, g);
  else hipLaunchKernelGGL(cg_stgcn_domain_fwd_mfma2_kernel<1>, grid, block, lds, stream, x, adj, W, bias, y, ystats, g);
  return cg_launch_status();
}
