// Fused ST-GCN stage of CIST-GCN (reference: Domain_GCNN_layer.forward, CISTGCN.py:265-266, with
// ConvTemporalGraphical.forward :122-124 and the 1x1 `tcn` convolution :229-234).
//
//   G[b,ci,.,.] = graph product of x with the per-sample learned adjacency
//        domain 0 ("space", CISTGCN.py:117):  G[ci,q,v] = sum_t x[ci,t,v] * Adj[b,v,t,q]   (T x T per joint)
//        domain 1 ("time",  CISTGCN.py:110):  G[ci,t,w] = sum_v x[ci,t,v] * Adj[b,t,v,w]   (V x V per frame)
//   y[b,co,.,.] = bias[co] + sum_ci W[co,ci] * G[b,ci,.,.]                                  (channel mix)
//
// One workgroup owns (sample, tile of GT "groups"), a group being one joint (domain 0) or one frame
// (domain 1): its adjacency slabs, its x slice and the mixing weights are staged in LDS, the graph
// product is written to LDS only, and the channel mix reads it back as float4 — G never reaches HBM.
// Per-channel f64 sums of y for the following train-mode BatchNorm are an optional epilogue.
//
// LDS images (floats; Jp = J rounded up to 4 so that rows can be read as ds_read_b128):
//   sA [GT][J][Jp]   adjacency, o padded with zeros         J = contraction/output length (T or V)
//   sX [Cin][GT][J]  input slice
//   sG [Cinp][PP]    graph product, PP = GT*Jp positions
//   sWt[Cin][Coutp]  W transposed (fwd) / sW [Cout][Cinp] (bwd)
#include "cg_common.h"
#include "stgcn_domain.h"
#include <atomic>
#include <cstdlib>
#include <stdlib.h>

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

// Kernel-generation switches for A/B runs (tools/bench_planes.py).  They are honoured only in a process started with
// CISTGCN_ABLATION=1 (read once); a production process never calls getenv() per launch and a stray variable cannot change
// which kernel runs.  INTEGRATION.md lists them.
static const char* cg_dom_env(const char* name) {
  static const bool ablation = getenv("CISTGCN_ABLATION") != nullptr;
  return ablation ? getenv(name) : nullptr;
}
// grid size from which the plane kernels fill the chip
// A process-wide switch read by every forward and backward launch: atomic, and changes are refused (the current value is returned, nothing
// moves) unless the process runs with CISTGCN_ABLATION=1 like the other kernel-generation switches - a test or tool that dies between
// setting and restoring it cannot silently change what production launches and captured HIP graphs run.
static std::atomic<long long> cg_dom_planes_min{256};
static long long cg_dom_planes_min_wgs() { return cg_dom_planes_min.load(std::memory_order_relaxed); }
extern "C" long long cg_stgcn_domain_planes_min_workgroups(long long n) {
  static const bool allowed = [] { const char* e = getenv("CISTGCN_ABLATION"); return e && e[0] == '1'; }();
  if (n >= 0 && allowed) return cg_dom_planes_min.exchange(n);
  return cg_dom_planes_min_wgs();
}

struct CgDomainGeom {
  int B, Cin, Cout, T, V;
  int GT, ntiles;      // groups per tile, tiles per sample
  int NG, J, Jp, PP;   // groups per sample, contraction length, padded, positions per tile
  int Cinp, Coutp;
  int per;             // tiles handled by one workgroup (weights / gradient accumulators persist across them)
  int xt, XS;          // xt = 1: x slice stored channel-fastest [GT][J][XS] (XS = Cinp + 4) so that the graph
                       // product reads four input channels per ds_read_b128 (4x4 register tile, Cin >= 16)
};

__device__ __forceinline__ int cg_dom_x_index(const CgDomainGeom& g, int ci, int grp, int j) {
  return g.xt ? (grp * g.J + j) * g.XS + ci : (ci * g.GT + grp) * g.J + j;
}
__device__ __forceinline__ int cg_dom_x_floats(const CgDomainGeom& g) {
  return g.xt ? g.GT * g.J * g.XS : ((g.Cin * g.GT * g.J + 3) & ~3);
}

// XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so
// hardware block b and b+8 share an L2.  Logical tiles are renumbered so that each XCD walks a contiguous
// range of (sample, tile) pairs: the tiles of one sample, which touch the same cache lines of x and y
// (4-byte columns of a 64-byte line in the space domain), then hit in ONE L2 instead of eight.
// Placement only affects speed, never results.  The grid is padded to a multiple of 8.
__device__ __forceinline__ int cg_dom_logical_block(const CgDomainGeom& g) {
  const int chunk = gridDim.x / 8;
  const int lid = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  const int nwg = (g.B * g.ntiles + g.per - 1) / g.per;
  return lid < nwg ? lid : -1;
}

template <int DOMAIN>
__device__ __forceinline__ long long cg_dom_off(const CgDomainGeom& g, int grp, int j) {
  // offset of element (group grp, index j) inside one channel plane of a (T,V) tensor
  return DOMAIN == 1 ? (long long)grp * g.V + j : (long long)j * g.V + grp;
}

template <int DOMAIN>
__device__ __forceinline__ void cg_dom_stage_inputs(const CgDomainGeom& g, const float* __restrict__ xb, const float* __restrict__ ab,
                                                    int g0, int ng, float* sA, float* sX) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int J = g.J, Jp = g.Jp, GT = g.GT;
  // adjacency slabs are contiguous in HBM: (ng, J, J)
  for (int e = tid; e < GT * J * Jp; e += nt) sA[e] = 0.f;
  for (int e = tid; e < cg_dom_x_floats(g); e += nt) sX[e] = 0.f;
  __syncthreads();
  for (int e = tid; e < ng * J * J; e += nt) {
    const int o = e % J, r = e / J;            // r = grp*J + j
    sA[r * Jp + o] = ab[e];
  }
  if (DOMAIN == 1) {
    const int run = ng * J;                    // contiguous (grp, j) run per channel
    for (int e = tid; e < g.Cin * run; e += nt) {
      const int ci = e / run, r = e - ci * run;
      sX[cg_dom_x_index(g, ci, r / J, r % J)] = xb[(long long)ci * g.T * g.V + (long long)g0 * g.V + r];
    }
  } else {
    for (int e = tid; e < g.Cin * J * ng; e += nt) {
      const int grp = e % ng, r = e / ng, j = r % J, ci = r / J;
      sX[cg_dom_x_index(g, ci, grp, j)] = xb[(long long)ci * g.T * g.V + (long long)j * g.V + g0 + grp];
    }
  }
}

// sG[ci][grp*Jp + o] = sum_j sX[ci][grp][j] * sA[grp][j][o]   (rows ci >= Cin of sG are zeroed)
__device__ __forceinline__ void cg_dom_graph_product(const CgDomainGeom& g, const float* sA, const float* sX, float* sG) {
  const int J = g.J, Jp = g.Jp, GT = g.GT, oq = Jp / 4;
  if (g.xt) {   // 4 input channels x 4 outputs per work item, both operands as ds_read_b128
    for (int idx = threadIdx.x; idx < (g.Cinp / 4) * GT * oq; idx += blockDim.x) {
      const int oc = idx % oq, r = idx / oq, grp = r % GT, iq = r / GT;
      float acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][q] = 0.f;
      const float* xr = sX + (grp * J) * g.XS + 4 * iq;
      const float* ar = sA + (grp * J) * Jp + 4 * oc;
      for (int j = 0; j < J; ++j) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + j * g.XS);
        const float4 av = *reinterpret_cast<const float4*>(ar + j * Jp);
        const float x4[4] = {xv.x, xv.y, xv.z, xv.w};
        const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[a][q] = fmaf(x4[a], a4[q], acc[a][q]);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
        *reinterpret_cast<float4*>(sG + (4 * iq + a) * g.PP + grp * Jp + 4 * oc) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
    }
    return;
  }
  for (int idx = threadIdx.x; idx < g.Cinp * GT * oq; idx += blockDim.x) {
    const int oc = idx % oq, r = idx / oq;     // r = ci*GT + grp
    const int grp = r % GT, ci = r / GT;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ci < g.Cin) {
      const float* xr = sX + r * J;
      const float* ar = sA + (grp * J) * Jp + 4 * oc;
      for (int j = 0; j < J; ++j) {
        const float xv = xr[j];
        const float4 a = *reinterpret_cast<const float4*>(ar + j * Jp);
        acc.x = fmaf(xv, a.x, acc.x); acc.y = fmaf(xv, a.y, acc.y);
        acc.z = fmaf(xv, a.z, acc.z); acc.w = fmaf(xv, a.w, acc.w);
      }
    }
    *reinterpret_cast<float4*>(sG + ci * g.PP + grp * Jp + 4 * oc) = acc;
  }
}

template <int DOMAIN>
__global__ __launch_bounds__(256) void cg_stgcn_domain_fwd_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                  const float* __restrict__ W, const float* __restrict__ bias,
                                                                  float* __restrict__ y, double* __restrict__ ystats, CgDomainGeom g) {
  float* sA = reinterpret_cast<float*>(cg_dyn_lds);
  float* sX = sA + g.GT * g.J * g.Jp;
  float* sG = sX + cg_dom_x_floats(g);
  float* sWt = sG + g.Cinp * g.PP;
  double* sStat = reinterpret_cast<double*>(sWt + g.Cin * g.Coutp);

  const int wg = cg_dom_logical_block(g);
  if (wg < 0) return;
  const long long TV = (long long)g.T * g.V;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int total = g.B * g.ntiles;

  // the mixing weights and the statistics accumulators are staged once per workgroup and reused for all its tiles
  for (int e = tid; e < g.Cin * g.Coutp; e += nt) {
    const int co = e % g.Coutp, ci = e / g.Coutp;
    sWt[e] = co < g.Cout ? W[co * g.Cin + ci] : 0.f;
  }
  if (ystats) for (int e = tid; e < 2 * g.Coutp; e += nt) sStat[e] = 0.0;

  for (int it = 0; it < g.per; ++it) {
    const int lid = wg * g.per + it;
    if (lid >= total) break;                     // uniform across the workgroup
    const int b = lid / g.ntiles, tile = lid % g.ntiles;
    const int g0 = tile * g.GT, ng = min(g.GT, g.NG - g0);
    const float* xb = x + (long long)b * g.Cin * TV;
    const float* ab = adj + ((long long)b * g.NG + g0) * g.J * g.J;
    __syncthreads();                             // previous tile fully consumed before its LDS images are overwritten
    cg_dom_stage_inputs<DOMAIN>(g, xb, ab, g0, ng, sA, sX);
    __syncthreads();
    cg_dom_graph_product(g, sA, sX, sG);
    __syncthreads();

    // channel mix: 4 output channels x 4 positions per work item
    const int pq = g.PP / 4, cq = g.Coutp / 4;
    float* yb = y + (long long)b * g.Cout * TV;
    for (int idx = tid; idx < cq * pq; idx += nt) {
      const int pc = idx % pq, cc = idx / pq;
      float acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][q] = 0.f;
      for (int ci = 0; ci < g.Cin; ++ci) {
        const float4 w = *reinterpret_cast<const float4*>(sWt + ci * g.Coutp + 4 * cc);
        const float4 v = *reinterpret_cast<const float4*>(sG + ci * g.PP + 4 * pc);
        const float wv[4] = {w.x, w.y, w.z, w.w};
        const float gv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[a][q] = fmaf(wv[a], gv[q], acc[a][q]);
      }
      const int pos0 = 4 * pc, grp = pos0 / g.Jp, o0 = pos0 % g.Jp;   // 4 positions never straddle a group (Jp % 4 == 0)
      if (grp >= ng) continue;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int co = 4 * cc + a;
        if (co >= g.Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
        double s = 0.0, sq = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int o = o0 + q;
          if (o >= g.J) continue;
          const float v = acc[a][q] + bv;
          yb[co * TV + cg_dom_off<DOMAIN>(g, g0 + grp, o)] = v;
          s += (double)v; sq += (double)v * (double)v;
        }
        if (ystats) { atomicAdd(&sStat[2 * co], s); atomicAdd(&sStat[2 * co + 1], sq); }
      }
    }
  }
  if (ystats) {
    __syncthreads();
    double* rep = ystats + (long long)(blockIdx.x % CG_STAT_REPLICAS) * 2 * g.Cout;
    for (int e = tid; e < 2 * g.Cout; e += nt) atomicAdd(&rep[e], sStat[e]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Matrix-core forward for wide layers (Cin, Cout >= 16): both products of a tile run on v_mfma_f32_16x16x4_f32
// (exact f32 fma chains).  Per tile and group:   G (Cin x J) = X (Cin x J) . A (J x J)   then, over the whole tile,
// Y (Cout x positions) = W (Cout x Cin) . G.   LDS images are padded so that every 16x16x4 fragment read is in
// bounds and reads zeros outside the data:  sA [GT][Jk][Jg], sX [GT][Jk][XS] (channel fastest), sG [Cink][GT*Jg],
// sWt [Cink][Coutm]   with Jk = J up to 4, Jg = J up to 16, Cink = Cin up to 4 (>= 16), Coutm = Cout up to 16.
// ---------------------------------------------------------------------------------------------------------
typedef float cg_dom_f32x4 __attribute__((vector_size(16)));

struct CgDomMfma {
  int Jk, Jg, PPg, Cink, Cinm, Coutm, XS;
};

__host__ __device__ static inline CgDomMfma cg_dom_mfma_geom(const CgDomainGeom& g) {
  CgDomMfma m;
  m.Jk = (g.J + 3) & ~3;
  m.Jg = (g.J + 15) & ~15;
  m.PPg = g.GT * m.Jg;
  m.Cink = (g.Cin + 3) & ~3;
  m.Cinm = (g.Cin + 15) & ~15;
  m.Coutm = (g.Cout + 15) & ~15;
  m.XS = m.Cinm + 4;
  return m;
}

static size_t cg_dom_mfma_lds_bytes(const CgDomainGeom& g) {
  const CgDomMfma m = cg_dom_mfma_geom(g);
  const size_t f = (size_t)g.GT * m.Jk * m.Jg + (size_t)g.GT * m.Jk * m.XS + (size_t)m.Cinm * m.PPg + (size_t)m.Cink * m.Coutm;
  return f * sizeof(float) + (size_t)2 * m.Coutm * sizeof(double) + 16;
}

template <int DOMAIN>
__global__ __launch_bounds__(256) void cg_stgcn_domain_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                       const float* __restrict__ W, const float* __restrict__ bias,
                                                                       float* __restrict__ y, double* __restrict__ ystats, CgDomainGeom g) {
  const CgDomMfma m = cg_dom_mfma_geom(g);
  float* sA = reinterpret_cast<float*>(cg_dyn_lds);
  float* sX = sA + g.GT * m.Jk * m.Jg;
  float* sG = sX + g.GT * m.Jk * m.XS;
  float* sWt = sG + m.Cinm * m.PPg;
  double* sStat = reinterpret_cast<double*>(sWt + m.Cink * m.Coutm);

  const int wg = cg_dom_logical_block(g);
  if (wg < 0) return;
  const long long TV = (long long)g.T * g.V;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int total = g.B * g.ntiles;
  const int J = g.J, GT = g.GT;

  for (int e = tid; e < m.Cink * m.Coutm; e += nt) {
    const int co = e % m.Coutm, ci = e / m.Coutm;
    sWt[e] = (co < g.Cout && ci < g.Cin) ? W[co * g.Cin + ci] : 0.f;
  }
  if (ystats) for (int e = tid; e < 2 * m.Coutm; e += nt) sStat[e] = 0.0;

  for (int it = 0; it < g.per; ++it) {
    const int lid = wg * g.per + it;
    if (lid >= total) break;                     // uniform across the workgroup
    const int b = lid / g.ntiles, tile = lid % g.ntiles;
    const int g0 = tile * GT, ng = min(GT, g.NG - g0);
    const float* xb = x + (long long)b * g.Cin * TV;
    const float* ab = adj + ((long long)b * g.NG + g0) * J * J;
    __syncthreads();                             // previous tile fully consumed
    for (int e = tid; e < GT * m.Jk * m.Jg; e += nt) sA[e] = 0.f;
    for (int e = tid; e < GT * m.Jk * m.XS; e += nt) sX[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < ng * J * J; e += nt) {            // adjacency slabs: contiguous (ng, J, J)
      const int o = e % J, r = e / J, j = r % J, grp = r / J;
      sA[(grp * m.Jk + j) * m.Jg + o] = ab[e];
    }
    if (DOMAIN == 1) {
      const int run = ng * J;
      for (int e = tid; e < g.Cin * run; e += nt) {
        const int ci = e / run, r = e - ci * run, grp = r / J, j = r % J;
        sX[(grp * m.Jk + j) * m.XS + ci] = xb[(long long)ci * TV + (long long)g0 * g.V + r];
      }
    } else {
      for (int e = tid; e < g.Cin * J * ng; e += nt) {
        const int grp = e % ng, r = e / ng, j = r % J, ci = r / J;
        sX[(grp * m.Jk + j) * m.XS + ci] = xb[(long long)ci * TV + (long long)j * g.V + g0 + grp];
      }
    }
    __syncthreads();

    // graph product on the matrix cores: tiles (group, 16 channels, 16 outputs) dealt round-robin to the 4 waves
    const int mt = m.Cinm / 16, ntl = m.Jg / 16;
    for (int t = wv; t < GT * mt * ntl; t += 4) {
      const int n0 = (t % ntl) * 16, r = t / ntl, m0 = (r % mt) * 16, grp = r / mt;
      cg_dom_f32x4 acc = cg_dom_f32x4{0.f, 0.f, 0.f, 0.f};
      const float* xa = sX + (grp * m.Jk + l4) * m.XS + m0 + l15;
      const float* aa = sA + (grp * m.Jk + l4) * m.Jg + n0 + l15;
      for (int k0 = 0; k0 < m.Jk; k0 += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[k0 * m.XS], aa[k0 * m.Jg], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) sG[(m0 + 4 * l4 + q) * m.PPg + grp * m.Jg + n0 + l15] = acc[q];
    }
    __syncthreads();

    // channel mix on the matrix cores: tiles (16 output channels, 16 positions)
    float* yb = y + (long long)b * g.Cout * TV;
    const int ct = m.Coutm / 16, pt = m.PPg / 16;
    for (int t = wv; t < ct * pt; t += 4) {
      const int n0 = (t % pt) * 16, m0 = (t / pt) * 16;
      cg_dom_f32x4 acc = cg_dom_f32x4{0.f, 0.f, 0.f, 0.f};
      const float* wa = sWt + l4 * m.Coutm + m0 + l15;
      const float* ga = sG + l4 * m.PPg + n0 + l15;
      for (int k0 = 0; k0 < m.Cink; k0 += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k0 * m.Coutm], ga[k0 * m.PPg], acc, 0, 0, 0);
      const int pos = n0 + l15, grp = pos / m.Jg, o = pos % m.Jg;
      const bool pos_ok = grp < ng && o < J;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = m0 + 4 * l4 + q;
        const bool ok = co < g.Cout && pos_ok;
        const float v = ok ? acc[q] + (bias ? bias[co] : 0.f) : 0.f;
        if (ok) yb[co * TV + cg_dom_off<DOMAIN>(g, g0 + grp, o)] = v;
        if (ystats) {          // the 16 lanes of a row hold the same channel: reduce them before touching LDS
          float s1 = v, s2 = v * v;
          s1 = cg_row16_sum(s1); s2 = cg_row16_sum(s2);
          if (l15 == 0 && co < g.Cout) { atomicAdd(&sStat[2 * co], (double)s1); atomicAdd(&sStat[2 * co + 1], (double)s2); }
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

// Backward of the fused stage.  Recomputes G from x and Adj, then
//   dG = W^T dy ; dx = dG A^T ; dAdj = x^T dG ; dW += dy G^T ; db += sum dy.
// dW / db partial sums stay in registers across the tiles of a workgroup and are added to HBM once at its end.
template <int DOMAIN>
__global__ __launch_bounds__(256) void cg_stgcn_domain_bwd_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                  const float* __restrict__ W, const float* __restrict__ dy,
                                                                  float* __restrict__ dx, float* __restrict__ dadj,
                                                                  float* __restrict__ dW, float* __restrict__ dbias, int replicas,
                                                                  CgDomainGeom g) {
  // weight/bias gradients are accumulated with fp32 atomics; spreading the workgroups over `replicas`
  // copies keeps same-address contention low (summed by cg_dom_fold_replicas_kernel afterwards)
  dW += (long long)(blockIdx.x % replicas) * (g.Cout * g.Cin + g.Cout);
  if (dbias) dbias = dW + g.Cout * g.Cin;
  float* sA = reinterpret_cast<float*>(cg_dyn_lds);
  float* sX = sA + g.GT * g.J * g.Jp;
  float* sG = sX + cg_dom_x_floats(g);
  float* sDG = sG + g.Cinp * g.PP;
  float* sDY = sDG + g.Cinp * g.PP;
  float* sW = sDY + g.Coutp * g.PP;        // [Coutp][Cinp]

  const int wg = cg_dom_logical_block(g);
  if (wg < 0) return;
  const long long TV = (long long)g.T * g.V;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int J = g.J, Jp = g.Jp, GT = g.GT, PP = g.PP;
  const int total = g.B * g.ntiles;
  const int pq = PP / 4, iq = g.Cinp / 4, cq = g.Coutp / 4, oq = Jp / 4;

  for (int e = tid; e < g.Coutp * g.Cinp; e += nt) {
    const int ci = e % g.Cinp, co = e / g.Cinp;
    sW[e] = (co < g.Cout && ci < g.Cin) ? W[co * g.Cin + ci] : 0.f;
  }
  // register accumulators: dW tile (4 co x 4 ci) of work item idx = tid (+ nt ...), one db partial per thread
  constexpr int kMaxWItems = 4;            // ceil(cq*iq / 256) <= 4 for Cout, Cin <= 128
  float wacc[kMaxWItems][4][4];
#pragma unroll
  for (int u = 0; u < kMaxWItems; ++u)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) wacc[u][a][q] = 0.f;
  float bacc = 0.f;

  for (int it = 0; it < g.per; ++it) {
    const int lid = wg * g.per + it;
    if (lid >= total) break;                     // uniform across the workgroup
    const int b = lid / g.ntiles, tile = lid % g.ntiles;
    const int g0 = tile * g.GT, ng = min(g.GT, g.NG - g0);
    const float* xb = x + (long long)b * g.Cin * TV;
    const float* ab = adj + ((long long)b * g.NG + g0) * g.J * g.J;
    const float* dyb = dy + (long long)b * g.Cout * TV;
    __syncthreads();                             // previous tile fully consumed
    cg_dom_stage_inputs<DOMAIN>(g, xb, ab, g0, ng, sA, sX);
    for (int e = tid; e < g.Coutp * PP; e += nt) {
      const int pos = e % PP, co = e / PP, grp = pos / Jp, o = pos % Jp;
      float v = 0.f;
      if (co < g.Cout && grp < ng && o < J) v = dyb[co * TV + cg_dom_off<DOMAIN>(g, g0 + grp, o)];
      sDY[e] = v;
    }
    __syncthreads();
    cg_dom_graph_product(g, sA, sX, sG);
    // dG[ci][pos] = sum_co W[co][ci] dy[co][pos]
    for (int idx = tid; idx < iq * pq; idx += nt) {
      const int pc = idx % pq, ic = idx / pq;
      float acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][q] = 0.f;
      for (int co = 0; co < g.Cout; ++co) {
        const float4 w = *reinterpret_cast<const float4*>(sW + co * g.Cinp + 4 * ic);
        const float4 v = *reinterpret_cast<const float4*>(sDY + co * PP + 4 * pc);
        const float wv[4] = {w.x, w.y, w.z, w.w};
        const float gv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[a][q] = fmaf(wv[a], gv[q], acc[a][q]);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
        *reinterpret_cast<float4*>(sDG + (4 * ic + a) * PP + 4 * pc) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
    }
    __syncthreads();

    // dx[ci][grp][j] = sum_o dG[ci][grp][o] * A[grp][j][o]
    float* dxb = dx + (long long)b * g.Cin * TV;
    for (int idx = tid; idx < g.Cin * ng * J; idx += nt) {
      int ci, grp, j;
      if (DOMAIN == 1) { j = idx % J; const int r = idx / J; grp = r % ng; ci = r / ng; }
      else { grp = idx % ng; const int r = idx / ng; j = r % J; ci = r / J; }
      const float* dg = sDG + ci * PP + grp * Jp;
      const float* ar = sA + (grp * J + j) * Jp;
      float sacc = 0.f;
      for (int oc = 0; oc < oq; ++oc) {
        const float4 d = *reinterpret_cast<const float4*>(dg + 4 * oc);
        const float4 a = *reinterpret_cast<const float4*>(ar + 4 * oc);
        sacc = fmaf(d.x, a.x, sacc); sacc = fmaf(d.y, a.y, sacc); sacc = fmaf(d.z, a.z, sacc); sacc = fmaf(d.w, a.w, sacc);
      }
      dxb[ci * TV + cg_dom_off<DOMAIN>(g, g0 + grp, j)] = sacc;
    }
    // dAdj[grp][j][o] = sum_ci x[ci][grp][j] * dG[ci][grp][o]
    float* dab = dadj + ((long long)b * g.NG + g0) * J * J;
    for (int idx = tid; idx < ng * J * oq; idx += nt) {
      const int oc = idx % oq, r = idx / oq, j = r % J, grp = r / J;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int ci = 0; ci < g.Cin; ++ci) {
        const float xv = sX[cg_dom_x_index(g, ci, grp, j)];
        const float4 d = *reinterpret_cast<const float4*>(sDG + ci * PP + grp * Jp + 4 * oc);
        acc.x = fmaf(xv, d.x, acc.x); acc.y = fmaf(xv, d.y, acc.y);
        acc.z = fmaf(xv, d.z, acc.z); acc.w = fmaf(xv, d.w, acc.w);
      }
      const float av[4] = {acc.x, acc.y, acc.z, acc.w};
      float* row = dab + ((long long)grp * J + j) * J;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * oc + q < J) row[4 * oc + q] = av[q];
    }
    // dW[co][ci] += sum_pos dy[co][pos] * G[ci][pos]   (4x4 register tile per work item, float4 along positions)
#pragma unroll
    for (int u = 0; u < kMaxWItems; ++u) {
      const int idx = tid + u * 256;
      if (idx < cq * iq) {
        const int ic = idx % iq, cc = idx / iq;
        for (int pc = 0; pc < pq; ++pc) {
          float dv[4][4], gv[4][4];
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const float4 d = *reinterpret_cast<const float4*>(sDY + (4 * cc + a) * PP + 4 * pc);
            dv[a][0] = d.x; dv[a][1] = d.y; dv[a][2] = d.z; dv[a][3] = d.w;
            const float4 v = *reinterpret_cast<const float4*>(sG + (4 * ic + a) * PP + 4 * pc);
            gv[a][0] = v.x; gv[a][1] = v.y; gv[a][2] = v.z; gv[a][3] = v.w;
          }
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) wacc[u][a][q] = fmaf(dv[a][e], gv[q][e], wacc[u][a][q]);
        }
      }
    }
    if (dbias && tid < g.Cout) {
      float sacc = 0.f;
      for (int p = 0; p < PP; ++p) sacc += sDY[tid * PP + p];
      bacc += sacc;
    }
  }

  // one atomic per weight-gradient entry and workgroup
#pragma unroll
  for (int u = 0; u < kMaxWItems; ++u) {
    const int idx = tid + u * 256;
    if (idx < cq * iq) {
      const int ic = idx % iq, cc = idx / iq;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = 4 * cc + a, ci = 4 * ic + q;
          if (co < g.Cout && ci < g.Cin) atomicAdd(&dW[co * g.Cin + ci], wacc[u][a][q]);
        }
    }
  }
  if (dbias && tid < g.Cout) atomicAdd(&dbias[tid], bacc);
}

__global__ void cg_dom_fold_replicas_kernel(const float* __restrict__ ws, int replicas, int n_w, int n_b,
                                            float* __restrict__ dW, float* __restrict__ dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_w + n_b) return;
  float s = 0.f;
  for (int r = 0; r < replicas; ++r) s += ws[(long long)r * (n_w + n_b) + i];
  if (i < n_w) dW[i] = s;
  else if (dbias) dbias[i - n_w] = s;
}

#define CG_DOM_REPLICAS 32

// ---- host side -------------------------------------------------------------------------------------
static size_t cg_dom_lds_bytes(const CgDomainGeom& g, bool bwd) {
  const size_t xf = g.xt ? (size_t)g.GT * g.J * g.XS : (((size_t)g.Cin * g.GT * g.J + 3) & ~(size_t)3);
  size_t f = (size_t)g.GT * g.J * g.Jp + xf + (size_t)g.Cinp * g.PP;
  if (bwd) f += (size_t)g.Cinp * g.PP + (size_t)g.Coutp * g.PP + (size_t)g.Coutp * g.Cinp;
  else f += (size_t)g.Cin * g.Coutp;
  size_t bytes = f * sizeof(float);
  if (!bwd) bytes += (size_t)2 * g.Coutp * sizeof(double) + 8;
  return bytes;
}

static int cg_dom_geom(CgDomainGeom& g, int B, int Cin, int Cout, int T, int V, int domain, bool bwd) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || (domain != 0 && domain != 1)) return CG_ESHAPE;
  g.B = B; g.Cin = Cin; g.Cout = Cout; g.T = T; g.V = V;
  g.NG = domain == 1 ? T : V;
  g.J = domain == 1 ? V : T;
  g.Jp = (g.J + 3) & ~3;
  g.Cinp = (Cin + 3) & ~3;
  g.Coutp = (Cout + 3) & ~3;
  g.xt = Cin >= 16 ? 1 : 0;
  g.XS = g.Cinp + 4;
  // groups per tile: the largest tile that keeps the LDS image <= 64 KiB (two workgroups per CU) and
  // the grid >= 1024 workgroups; a single group is accepted up to the full 160 KiB.  Small problems
  // (fewer than 1024 single-group tiles) instead aim at one wave of ~256 workgroups: one round on the 256 CUs.
  const long long tiles1 = (long long)B * g.NG;
  const int want_wgs = tiles1 < 1024 ? 256 : 1024;
  const char* env_gt = cg_dom_env("CG_DOM_GT");          // tuning aid (tools/bench_planes.py); unset in production
  int best = 0;
  for (int gt = 1; gt <= g.NG; ++gt) {
    g.GT = gt; g.PP = gt * g.Jp;
    const size_t bytes = cg_dom_lds_bytes(g, bwd);
    if (gt == 1) {
      if (bytes > 160 * 1024 - 256) return CG_ESHAPE;
      best = 1;
      continue;
    }
    if (env_gt) { if (gt > atoi(env_gt) || bytes > 150 * 1024) break; best = gt; continue; }
    if (bytes > 64 * 1024) break;
    if ((long long)B * ((g.NG + gt - 1) / gt) < want_wgs) break;
    best = gt;
  }
  if (best == 0) return CG_ESHAPE;
  g.GT = best; g.PP = best * g.Jp;
  g.ntiles = (g.NG + best - 1) / best;
  if ((long long)B * g.ntiles > 2147483647LL) return CG_ESHAPE;
  if (Cin > 128 || Cout > 128 || Cout > 256) return CG_ESHAPE;       // register / thread budget of the backward accumulators
  // tiles per workgroup: amortise the weight staging (and, in backward, the dW/db atomics) while keeping ~2048 workgroups
  const long long total = (long long)B * g.ntiles;
  long long per = total / 2048;
  if (per < 1) per = 1;
  if (per > 16) per = 16;
  // space-domain forward: more tiles in flight per XCD than its 4 MiB L2 can hold let partially written y lines
  // escape to HBM (PMC: 91 MB written per launch instead of 53 MB) -> one tile per workgroup there
  if (domain == 0 && !bwd) per = 1;
  const char* env_per = cg_dom_env("CG_DOM_PER");        // tuning aid
  if (env_per) per = atoi(env_per) > 0 ? atoi(env_per) : per;
  g.per = (int)per;
  return CG_OK;
}

extern "C" int cg_stgcn_domain_fwd(const float* x, const float* adj, const float* W, const float* bias, float* y, double* ystats,
                                   int B, int Cin, int Cout, int T, int V, int domain, void* stream_) {
  if (!x || !adj || !W || !y) return CG_EARG;
  CgDomainGeom g;
  int st = cg_dom_geom(g, B, Cin, Cout, T, V, domain, false);
  if (st != CG_OK) return st;
  // wide layers, third generation: plane kernels (stgcn_domain_planes.hip) - whole plane rows in HBM, LDS as the transposer
  // (a grid of fewer than 256 workgroups - small batches - leaves most CUs idle: the tile kernels split a sample finer; the
  // time-domain variant for odd V reads its slabs one float per lane and measured no faster than the tile kernel)
  if ((Cin >= 16 || Cout >= 16) && (long long)B * ((Cout + 15) / 16) >= cg_dom_planes_min_wgs() && !(domain == 1 && (V & 1)) &&
      cg_dom_env("CG_DOM_NO_PLANES") == nullptr) {
    st = cg_domp_fwd_launch(x, adj, W, bias, y, ystats, B, Cin, Cout, T, V, domain, (hipStream_t)stream_);
    if (st != CG_ESHAPE) return st;
  }
  // wide layers: both products on the matrix cores with the shared staging of stgcn_domain_mfma.hip (CG_DOM_FWD_OLD=1: the
  // first-generation kernels below, kept for A/B runs)
  // (measured at 64->64, B=256, T=50, V=22: time domain 146 vs 173 us; in the space domain both generations sit at ~280 us,
  // bound by the 4-byte-column accesses of x and y - DESIGN.md section 4 - so the first-generation kernel stays there)
  if ((Cin >= 16 || Cout >= 16) && domain == 1 && V <= 64 && cg_dom_env("CG_DOM_FWD_OLD") == nullptr) {
    st = cg_domm_fwd_launch(x, adj, W, bias, y, ystats, B, Cin, Cout, T, V, domain, (hipStream_t)stream_);
    if (st != CG_ESHAPE) return st;
  }
  // matrix cores pay off in the time domain (164 vs 243 us at C=64, B=256); in the space domain the kernel is bound by
  // its 4-byte-column accesses, the MFMA variant is no faster there and measured 1.7x the HBM write traffic (PMC)
  const bool mfma = Cin >= 16 && Cout >= 16 && domain == 1 && cg_dom_env("CG_DOM_NO_MFMA") == nullptr;
  if (mfma) {
    // matrix-core path: its own LDS images; fit the tile to 64 KiB where possible
    while (g.GT > 1 && cg_dom_mfma_lds_bytes(g) > 64 * 1024) { --g.GT; g.PP = g.GT * g.Jp; }
    g.ntiles = (g.NG + g.GT - 1) / g.GT;
    if (cg_dom_mfma_lds_bytes(g) > 160 * 1024 - 256) return CG_ESHAPE;
    const long long total = (long long)B * g.ntiles;
    if (cg_dom_env("CG_DOM_PER") == nullptr) { long long per = total / 2048; g.per = (int)(per < 1 ? 1 : (per > 16 ? 16 : per)); }
    const size_t lds = cg_dom_mfma_lds_bytes(g);
    const long long nwg = (total + g.per - 1) / g.per;
    dim3 grid((unsigned)(((nwg + 7) / 8) * 8)), block(256);
    if (lds > 48 * 1024) {
      const void* fn = domain == 0 ? (const void*)cg_stgcn_domain_fwd_mfma_kernel<0> : (const void*)cg_stgcn_domain_fwd_mfma_kernel<1>;
      hipError_t e = cg_lds_limit(fn, lds);
      if (e != hipSuccess) return (int)e;
    }
    if (domain == 0) hipLaunchKernelGGL(cg_stgcn_domain_fwd_mfma_kernel<0>, grid, block, lds, (hipStream_t)stream_, x, adj, W, bias, y, ystats, g);
    else hipLaunchKernelGGL(cg_stgcn_domain_fwd_mfma_kernel<1>, grid, block, lds, (hipStream_t)stream_, x, adj, W, bias, y, ystats, g);
    return cg_launch_status();
  }
  const size_t lds = cg_dom_lds_bytes(g, false);
  const long long nwg = ((long long)B * g.ntiles + g.per - 1) / g.per;
  dim3 grid((unsigned)(((nwg + 7) / 8) * 8)), block(256);
  if (lds > 48 * 1024) {
    const void* fn = domain == 0 ? (const void*)cg_stgcn_domain_fwd_kernel<0> : (const void*)cg_stgcn_domain_fwd_kernel<1>;
    hipError_t e = cg_lds_limit(fn, lds);
    if (e != hipSuccess) return (int)e;
  }
  if (domain == 0) hipLaunchKernelGGL(cg_stgcn_domain_fwd_kernel<0>, grid, block, lds, (hipStream_t)stream_, x, adj, W, bias, y, ystats, g);
  else hipLaunchKernelGGL(cg_stgcn_domain_fwd_kernel<1>, grid, block, lds, (hipStream_t)stream_, x, adj, W, bias, y, ystats, g);
  return cg_launch_status();
}

extern "C" long long cg_stgcn_domain_bwd_ws_floats(int Cin, int Cout) {
  return (long long)CG_DOM_REPLICAS * ((long long)Cout * Cin + Cout);
}

extern "C" int cg_stgcn_domain_bwd(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj,
                                   float* dW, float* dbias, float* ws, int B, int Cin, int Cout, int T, int V, int domain,
                                   int ws_prezeroed, void* stream_) {
  if (!x || !adj || !W || !dy || !dx || !dadj || !dW || !ws) return CG_EARG;
  CgDomainGeom g;
  int st = cg_dom_geom(g, B, Cin, Cout, T, V, domain, true);
  if (st != CG_OK) return st;
  hipStream_t stream = (hipStream_t)stream_;
  const int n_w = Cout * Cin, n_b = Cout;
  hipError_t e = hipSuccess;
  if (!ws_prezeroed) {
    const int zs = cg_zero_fill(ws, (long long)CG_DOM_REPLICAS * (n_w + n_b) * (long long)sizeof(float), stream);
    if (zs != CG_OK) return zs;
  }
  // wide layers, space domain: plane backward (stgcn_domain_planes.hip)
  // (narrow inputs: the channel-mix-first order makes the graph products as wide as the OUTPUT, the tile kernels keep them
  // as wide as the input - 229 vs 279 us at 10 -> 64; small batches: as in the forward)
  if (Cin >= 16 && domain == 0 && (long long)B * ((T + 15) / 16) >= cg_dom_planes_min_wgs() && cg_dom_env("CG_DOM_NO_PLANES") == nullptr) {
    st = cg_domp_bwd_launch(x, adj, W, dy, dx, dadj, ws, CG_DOM_REPLICAS, B, Cin, Cout, T, V, domain, stream);
    if (st != CG_ESHAPE) {
      if (st != CG_OK) return st;
      hipLaunchKernelGGL(cg_dom_fold_replicas_kernel, dim3((unsigned)((n_w + n_b + 255) / 256)), dim3(256), 0, stream, ws, CG_DOM_REPLICAS,
                         n_w, n_b, dW, dbias);
      return cg_launch_status();
    }
  }
  // wide layers, time domain: local plane backward (a chunk of frames per workgroup)
  if (Cin >= 16 && domain == 1 && (long long)B * ((T + 7) / 8) >= cg_dom_planes_min_wgs() && cg_dom_env("CG_DOM_NO_PLANES") == nullptr) {
    st = cg_domp_bwd_time_launch(x, adj, W, dy, dx, dadj, ws, CG_DOM_REPLICAS, B, Cin, Cout, T, V, stream);
    if (st != CG_ESHAPE) {
      if (st != CG_OK) return st;
      hipLaunchKernelGGL(cg_dom_fold_replicas_kernel, dim3((unsigned)((n_w + n_b + 255) / 256)), dim3(256), 0, stream, ws, CG_DOM_REPLICAS,
                         n_w, n_b, dW, dbias);
      return cg_launch_status();
    }
  }
  // wide layers: every product on the matrix cores (stgcn_domain_mfma.hip); narrow ones (C <= 10 on both sides: CISTGCN-8,
  // the output block) stay on the VALU kernel below, where a 16-wide MFMA tile would be mostly padding
  if ((Cin >= 16 || Cout >= 16) && (domain == 1 ? V : T) <= 64 && cg_dom_env("CG_DOM_BWD_VALU") == nullptr) {
    st = cg_domm_bwd_launch(x, adj, W, dy, dx, dadj, ws, CG_DOM_REPLICAS, B, Cin, Cout, T, V, domain, stream);
    if (st != CG_ESHAPE) {
      if (st != CG_OK) return st;
      hipLaunchKernelGGL(cg_dom_fold_replicas_kernel, dim3((unsigned)((n_w + n_b + 255) / 256)), dim3(256), 0, stream, ws, CG_DOM_REPLICAS,
                         n_w, n_b, dW, dbias);
      return cg_launch_status();
    }
  }
  const size_t lds = cg_dom_lds_bytes(g, true);
  const long long nwg = ((long long)B * g.ntiles + g.per - 1) / g.per;
  dim3 grid((unsigned)(((nwg + 7) / 8) * 8)), block(256);
  if (lds > 48 * 1024) {
    const void* fn = domain == 0 ? (const void*)cg_stgcn_domain_bwd_kernel<0> : (const void*)cg_stgcn_domain_bwd_kernel<1>;
    e = cg_lds_limit(fn, lds);
    if (e != hipSuccess) return (int)e;
  }
  float* wsb = ws + n_w;   // non-null marker: bias partials live behind the weight partials of each replica
  if (domain == 0) hipLaunchKernelGGL(cg_stgcn_domain_bwd_kernel<0>, grid, block, lds, stream, x, adj, W, dy, dx, dadj, ws, wsb, CG_DOM_REPLICAS, g);
  else hipLaunchKernelGGL(cg_stgcn_domain_bwd_kernel<1>, grid, block, lds, stream, x, adj, W, dy, dx, dadj, ws, wsb, CG_DOM_REPLICAS, g);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  hipLaunchKernelGGL(cg_dom_fold_replicas_kernel, dim3((unsigned)((n_w + n_b + 255) / 256)), dim3(256), 0, stream, ws, CG_DOM_REPLICAS,
                     n_w, n_b, dW, dbias);
  return cg_launch_status();
}
