// "Plane" kernels of the fused ST-GCN stage (reference: Domain_GCNN_layer.forward, CISTGCN.py:265-266, with
// ConvTemporalGraphical.forward :122-124 and the 1x1 `tcn` convolution :229-234) for wide layers.
//
// The stage is y = W . (x (*) Adj) + bias, with (*) the graph product of the domain:
//   space (CISTGCN.py:117)  (x (*) A)[c,q,v] = sum_t x[c,t,v] A[b,v,t,q]      one T x T slab per joint v
//   time  (CISTGCN.py:110)  (x (*) A)[c,t,w] = sum_v x[c,t,v] A[b,t,v,w]      one V x V slab per frame t
// The channel mix commutes with the graph product (both are linear, they act on different axes), so the kernels here
// evaluate  y = (W . x) (*) Adj + bias : Z = W . x is a pointwise product over whole (T,V) planes - every HBM access of
// x is a contiguous run along the plane - and the graph product then runs on 16 channels of Z at a time out of an LDS
// image that is laid out group-major.  LDS is the transposer: HBM only ever sees whole plane rows.
// (The first two generations, stgcn_domain.hip / stgcn_domain_mfma.hip, tile by joint and touch x, y, dy and dx as 4-byte
// columns of 88-byte rows in the space domain: 5.7x the algorithmic HBM traffic, profiles/r02_stgcn_domain_pmc.txt.)
//
// Forward, one 256-thread workgroup per (sample b, chunk oc of 16 output channels), two workgroups per CU:
//   P1  Z[o,p] = sum_c W[oc+o,c] x[b,c,p]        v_mfma_f32_16x16x4_f32; the x operand straight from HBM/L2 as 16-byte
//                                                 pieces of plane rows; written to LDS as sZ[group][o][j]
//   P2  Y_g = Z_g . A_g  (+ bias, f64 sums)      per group (joint | frame), one wave per group, A_g straight from HBM/L2
//                                                 (rows are contiguous), result in place over Z_g
//   P3  y[b,oc+o,:,:] <- sZ                       whole plane rows, 16-byte stores
// The NOC workgroups of a sample read the same x[b] and Adj[b]; they are placed on one XCD, next to each other in
// dispatch order, so that HBM delivers those bytes once (placement affects speed only).
//
// Backward of the space domain: see the comment in front of cg_stgcn_planes_bwd_kernel.
#include "cg_common.h"
#include "stgcn_domain.h"
#include <stdlib.h>

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

typedef float cg_f32x4 __attribute__((vector_size(16)));
typedef float cg_f32x2 __attribute__((vector_size(8)));

#define CG_DOMP_THREADS 256
#define CG_DOMP_NW (CG_DOMP_THREADS / 64)

__device__ __forceinline__ unsigned cg_domp_div(unsigned n, unsigned magic) {     // n / d, magic = ceil(2^32 / d), 0 for d = 1; n < 2^20
  return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n;
}

// VW consecutive floats (VW * 4-byte aligned) -> registers; zeros when !ok
template <int VW>
__device__ __forceinline__ void cg_domp_ld(const float* __restrict__ p, bool ok, float (&v)[VW]) {
  if constexpr (VW == 4) {
    cg_f32x4 t = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) t = *reinterpret_cast<const cg_f32x4*>(p);
#pragma unroll
    for (int i = 0; i < VW; ++i) v[i] = t[i];
  } else if constexpr (VW == 2) {
    cg_f32x2 t = cg_f32x2{0.f, 0.f};
    if (ok) t = *reinterpret_cast<const cg_f32x2*>(p);
#pragma unroll
    for (int i = 0; i < VW; ++i) v[i] = t[i];
  } else {
    v[0] = ok ? p[0] : 0.f;
  }
}
template <int VW>
__device__ __forceinline__ void cg_domp_st(float* __restrict__ p, const float (&v)[VW]) {
  if constexpr (VW == 4) *reinterpret_cast<cg_f32x4*>(p) = cg_f32x4{v[0], v[1], v[2], v[3]};
  else if constexpr (VW == 2) *reinterpret_cast<cg_f32x2*>(p) = cg_f32x2{v[0], v[1]};
  else p[0] = v[0];
}

// position p of a (T,V) plane -> (group, index along the contraction axis)
template <int DOMAIN>
__device__ __forceinline__ void cg_domp_split(const CgDomP& g, int p, int& grp, int& j) {
  const int t = (int)cg_domp_div((unsigned)p, g.magicV), v = p - t * g.V;
  if (DOMAIN == 0) { grp = v; j = t; } else { grp = t; j = v; }
}

// ---------------------------------------------------------------------------------------------------------------------
// P1 of the forward (and the Z recomputation of the backward): Z[o][p] = sum_c W[o][c] X[c][p] for the positions
// [pbeg, pend) of the planes at `xb` (channel stride cs), W as the LDS image sW[16][WS] (zero padded), written to
// sZ[grp * GSTR + o * JS + j].  Units of 16 * VW positions are dealt to the waves; a lane holds VW consecutive positions.
// ---------------------------------------------------------------------------------------------------------------------
template <int DOMAIN, int VW>
__device__ __forceinline__ void cg_domp_mix_planes(const CgDomP& g, const float* __restrict__ xb, long long cs, int Cin, int KS,
                                                   const float* sW, int WS, float* sZ, int pbeg, int pend, int jshift) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, slot = lane >> 4;
  const int nunits = (pend - pbeg + 16 * VW - 1) / (16 * VW);
  const float* wrow = sW + l15 * WS + slot;
  for (int u = wave; u < nunits; u += CG_DOMP_NW) {
    const int p0 = pbeg + u * 16 * VW + VW * l15;
    const bool pok = p0 < pend;
    const float* xp = xb + p0 + (long long)slot * cs;
    cg_f32x4 acc[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[i] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    float xa[4][VW], xc[4][VW];
    auto ld = [&](int k0, float (&xv)[4][VW]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) cg_domp_ld<VW>(xp + (long long)(4 * (k0 + s)) * cs, pok && 4 * (k0 + s) + slot < Cin, xv[s]);
    };
    auto mm = [&](int k0, const float (&xv)[4][VW]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (k0 + s >= KS) break;                       // uniform
        const float a = wrow[4 * (k0 + s)];
#pragma unroll
        for (int i = 0; i < VW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xv[s][i], acc[i], 0, 0, 0);
      }
    };
    ld(0, xa);
    for (int k0 = 0; k0 < KS; k0 += 8) {
      if (k0 + 4 < KS) ld(k0 + 4, xc);
      mm(k0, xa);
      if (k0 + 4 < KS) {
        if (k0 + 8 < KS) ld(k0 + 8, xa);
        mm(k0 + 4, xc);
      }
    }
    if (pok) {
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        int grp, j;
        cg_domp_split<DOMAIN>(g, p0 + i, grp, j);
        float* dst = sZ + grp * g.GSTR + (j - jshift) + 4 * slot * g.JS;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r * g.JS] = acc[i][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
template <int DOMAIN, int VW, int VWB, int NL>
__global__ __launch_bounds__(CG_DOMP_THREADS, 2) void cg_stgcn_planes_fwd_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                 const float* __restrict__ W, const float* __restrict__ bias,
                                                                 float* __restrict__ y, double* __restrict__ ystats, CgDomP g) {
  float* sZ = reinterpret_cast<float*>(cg_dyn_lds);
  float* sW = sZ + g.zfloats;
  double* sStat = reinterpret_cast<double*>(sW + 16 * g.WS);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, slot = lane >> 4;
  // placement: the NOC chunks of a sample sit next to each other in the dispatch order of one XCD
  const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
  const int sb = s / g.NOC, oc = s - sb * g.NOC, b = sb * 8 + xcd;
  if (b >= g.B) return;
  const int o0 = oc * 16;

  for (int e = tid; e < 16 * g.WS; e += CG_DOMP_THREADS) {
    const int o = e / g.WS, c = e - o * g.WS;
    sW[e] = (o0 + o < g.Cout && c < g.Cin) ? W[(long long)(o0 + o) * g.Cin + c] : 0.f;
  }
  if (tid < 32) sStat[tid] = 0.0;
  __syncthreads();

  cg_domp_mix_planes<DOMAIN, VW>(g, x + (long long)b * g.Cin * g.TV, g.TV, g.Cin, g.KS, sW, g.WS, sZ, 0, g.TV, 0);
  __syncthreads();

  // P2: one wave per group
  const int J = g.J;
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int co = o0 + 4 * slot + r; bv[r] = (bias && co < g.Cout) ? bias[co] : 0.f; }
  float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int gi = wave; gi < g.NG; gi += CG_DOMP_NW) {
    const float* ag = adj + ((long long)b * g.NG + gi) * J * J;
    float* zs = sZ + gi * g.GSTR;
    const float* za = zs + l15 * g.JS + slot;
    cg_f32x4 acc[NL * VWB];
#pragma unroll
    for (int i = 0; i < NL * VWB; ++i) acc[i] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    float ba[4][NL][VWB], bc[4][NL][VWB];
    auto ld = [&](int k0, float (&v)[4][NL][VWB]) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int jr = 4 * (k0 + s2) + slot;
#pragma unroll
        for (int m = 0; m < NL; ++m) {
          const int q0 = (m * 16 + l15) * VWB;
          cg_domp_ld<VWB>(ag + (long long)jr * J + q0, jr < J && q0 < J, v[s2][m]);
        }
      }
    };
    auto mm = [&](int k0, const float (&v)[4][NL][VWB]) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        if (k0 + s2 >= g.JSTEPS) break;                // uniform
        const int jr = 4 * (k0 + s2) + slot;
        const float a = jr < J ? za[4 * (k0 + s2)] : 0.f;
#pragma unroll
        for (int m = 0; m < NL; ++m)
#pragma unroll
          for (int i = 0; i < VWB; ++i) acc[m * VWB + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[s2][m][i], acc[m * VWB + i], 0, 0, 0);
      }
    };
    ld(0, ba);
    for (int k0 = 0; k0 < g.JSTEPS; k0 += 8) {
      if (k0 + 4 < g.JSTEPS) ld(k0 + 4, bc);
      mm(k0, ba);
      if (k0 + 4 < g.JSTEPS) {
        if (k0 + 8 < g.JSTEPS) ld(k0 + 8, ba);
        mm(k0 + 4, bc);
      }
    }
    // every read of Z_g by this wave is behind us (the accumulators depend on them): Y_g goes in place
#pragma unroll
    for (int m = 0; m < NL; ++m) {
      const int q0 = (m * 16 + l15) * VWB;
      if (q0 < J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v[VWB];
#pragma unroll
          for (int i = 0; i < VWB; ++i) {
            v[i] = acc[m * VWB + i][r] + bv[r];
            st1[r] += v[i]; st2[r] += v[i] * v[i];
          }
          cg_domp_st<VWB>(zs + (4 * slot + r) * g.JS + q0, v);
        }
      }
    }
  }
  if (ystats) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = cg_row16_sum(st1[r]), c = cg_row16_sum(st2[r]);
      if (l15 == 0) { atomicAdd(&sStat[2 * (4 * slot + r)], (double)a); atomicAdd(&sStat[2 * (4 * slot + r) + 1], (double)c); }
    }
  }
  __syncthreads();

  // P3: whole plane rows
  const int nq = g.TV / VW;
  float* yb = y + ((long long)b * g.Cout + o0) * g.TV;
  for (int e = tid; e < 16 * nq; e += CG_DOMP_THREADS) {
    const int o = (int)cg_domp_div((unsigned)e, g.magicNQ), pv = e - o * nq;
    if (o0 + o >= g.Cout) break;
    float v[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      int grp, j;
      cg_domp_split<DOMAIN>(g, pv * VW + i, grp, j);
      v[i] = sZ[grp * g.GSTR + o * g.JS + j];
    }
    cg_domp_st<VW>(yb + (long long)o * g.TV + pv * VW, v);
  }
  if (ystats && tid < 32 && o0 + (tid >> 1) < g.Cout) {
    double* rep = ystats + (long long)(blockIdx.x % CG_STAT_REPLICAS) * 2 * g.Cout;
    atomicAdd(&rep[2 * o0 + tid], sStat[tid]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static unsigned cg_domp_magic(int d) { return d > 1 ? (unsigned)((0x100000000ULL + d - 1) / d) : 0u; }
static int cg_domp_up(int v, int m) { return (v + m - 1) / m * m; }

int cg_domp_geom(CgDomP& g, int B, int Cin, int Cout, int T, int V, int domain) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || V <= 0 || (domain != 0 && domain != 1)) return CG_ESHAPE;
  g.B = B; g.Cin = Cin; g.Cout = Cout; g.T = T; g.V = V; g.TV = T * V;
  g.NG = domain == 1 ? T : V;
  g.J = domain == 1 ? V : T;
  if (g.J > 64 || Cin > 128 || (long long)T * V > (1 << 20)) return CG_ESHAPE;
  g.JS = g.J + ((2 - g.J) & 3);                  // == 2 (mod 4): the k-strided fragment reads (row o = lane & 15) hit 32 different banks
  g.GSTR = 16 * g.JS + 2;
  g.NOC = (Cout + 15) / 16;
  g.KS = (Cin + 3) / 4;
  g.WS = 4 * cg_domp_up(g.KS, 4) + 2;
  g.JSTEPS = (g.J + 3) / 4;
  g.zfloats = cg_domp_up(g.NG * g.GSTR + 16, 4);
  g.VW = (g.TV % 4 == 0) ? 4 : (g.TV % 2 == 0) ? 2 : 1;
  if (g.J <= 16) { g.VWB = 1; g.NL = 1; }
  else if (g.J <= 32) { if (g.J % 2 == 0) { g.VWB = 2; g.NL = 1; } else { g.VWB = 1; g.NL = 2; } }
  else { if (g.J % 4 == 0) { g.VWB = 4; g.NL = 1; } else if (g.J % 2 == 0) { g.VWB = 2; g.NL = 2; } else { g.VWB = 1; g.NL = 4; } }
  g.magicV = cg_domp_magic(V);
  g.magicNQ = cg_domp_magic(g.TV / g.VW);
  return CG_OK;
}

size_t cg_domp_fwd_lds_bytes(const CgDomP& g) { return ((size_t)g.zfloats + 16 * (size_t)g.WS) * sizeof(float) + 32 * sizeof(double); }

int cg_domp_fwd_launch(const float* x, const float* adj, const float* W, const float* bias, float* y, double* ystats,
                       int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream) {
  CgDomP g;
  int st = cg_domp_geom(g, B, Cin, Cout, T, V, domain);
  if (st != CG_OK) return st;
  const size_t lds = cg_domp_fwd_lds_bytes(g);
  if (lds > 160 * 1024 - 512) return CG_ESHAPE;
  dim3 grid((unsigned)(8 * ((B + 7) / 8) * g.NOC)), block(CG_DOMP_THREADS);
#define CG_DOMP_FWD(D, VW_, VWB_, NL_)                                                                                              \
  if (domain == D && g.VW == VW_ && g.VWB == VWB_ && g.NL == NL_) {                                                               \
    hipError_t e = hipFuncSetAttribute((const void*)cg_stgcn_planes_fwd_kernel<D, VW_, VWB_, NL_>,                                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                       \
    if (e != hipSuccess) return (int)e;                                                                                             \
    hipLaunchKernelGGL((cg_stgcn_planes_fwd_kernel<D, VW_, VWB_, NL_>), grid, block, lds, stream, x, adj, W, bias, y, ystats, g);   \
    return cg_launch_status();                                                                                                      \
  }
  // instantiated for the (T, V) families of the model's workloads; anything else falls back to the tile kernels
  CG_DOMP_FWD(0, 4, 2, 2)      // T = 50 (V even or odd as long as T*V % 4 == 0), space
  CG_DOMP_FWD(0, 2, 2, 2)      // T = 50, V = 25
  CG_DOMP_FWD(0, 4, 1, 1)      // T = 10
  CG_DOMP_FWD(0, 2, 1, 1)
  CG_DOMP_FWD(1, 4, 2, 1)      // V = 22 / 18, time
  CG_DOMP_FWD(1, 2, 1, 2)      // V = 25
#undef CG_DOMP_FWD
  return CG_ESHAPE;
}
