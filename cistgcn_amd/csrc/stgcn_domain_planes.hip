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

// Column swizzle of the dY image sY[joint][16 channels o][16 frames q] (row stride 16 floats).  The image is read two ways by the
// matrix cores: (q across the 16 lanes of a row, o across the four lane rows) for dA and (o across lanes, q across rows) for dZ.
// Rows of opposite parity sit in opposite halves of the 32 banks (stride 16), which makes the first pattern conflict free; the
// XOR below spreads the 8 rows of one parity over all 16 columns of their half for the second one: a lane row reads columns
// {q, q ^ 4}, so the swizzle must differ between rows in the bits other than bit 2.
__device__ __forceinline__ int cg_domp_swz(int o) { const int h = o >> 1; return (h & 3) | ((h & 4) << 1); }

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
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, slot = lane >> 4;
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
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
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
  // BatchNorm channel sums: fp32 inside one group (at most J values per lane), f64 across the groups and lanes - the variance is formed as
  // E[y^2] - E[y]^2, activations with |mean| >> std would lose it in fp32 sums over the whole plane
  double st1[4] = {0.0, 0.0, 0.0, 0.0}, st2[4] = {0.0, 0.0, 0.0, 0.0};
  for (int gi = wave; gi < g.NG; gi += CG_DOMP_NW) {
    float gs1[4] = {0.f, 0.f, 0.f, 0.f}, gs2[4] = {0.f, 0.f, 0.f, 0.f};
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
            gs1[r] += v[i]; gs2[r] += v[i] * v[i];
          }
          cg_domp_st<VWB>(zs + (4 * slot + r) * g.JS + q0, v);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { st1[r] += (double)gs1[r]; st2[r] += (double)gs2[r]; }
  }
  if (ystats) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double a = cg_row16_sum(st1[r]), c = cg_row16_sum(st2[r]);
      if (l15 == 0) { atomicAdd(&sStat[2 * (4 * slot + r)], a); atomicAdd(&sStat[2 * (4 * slot + r) + 1], c); }
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
// backward, space domain (Adj (B,V,T,T), one T x T slab per joint)
//
// With Z = W . x and y = Z (*) Adj + bias:
//   dZ[o,t,v] = sum_q dY[o,q,v] A[v,t,q]        dA[v,t,q] = sum_o Z[o,t,v] dY[o,q,v]
//   dx[c,t,v] = sum_o W[o,c] dZ[o,t,v]          dW[o,c] = sum_{b,t,v} dZ[o,t,v] x[c,t,v]        db[o] = sum dY[o,.,.]
// Every result that is indexed by the frame t of x needs only rows t of the slabs, so a workgroup owns (sample b, chunk
// of TC <= 16 frames): its piece of x ([Cin][TC*V], whole plane rows) stays in LDS, dY[b] streams through LDS in pieces
// of [16 output channels][16 frames q][V] (whole plane rows again; the NTC workgroups of a sample share them through L2),
// rows t of the slabs come straight from HBM/L2 as matrix-core operands, and dA / dx accumulate in registers over the
// stream.  Nothing is touched as a 4-byte column; HBM sees x, dY and Adj once and dx, dA once.
//
//   per chunk oc of 16 output channels:
//     S2  Z[o,p] = W[oc+o,:] . sX[:,p]                              -> sZ[v][o][t]
//     per piece qc (16 frames q):
//       S4  per joint v (a wave owns joints v = wave, wave + 8, ...):
//             dA_v[t, q] += sum_o sZ[v][o][t] sY[v][o][q]           (register tiles, persistent)
//             dZ_v[o, t] += sum_q sY[v][o][q] A_v[t0 + t, q]        (register tiles, per oc)
//     S5  dZ -> sdZ[o][p]
//     S6  waves 0-3: dW[oc+o, c] += sum_p sdZ[o][p] sX[c][p]  -> fp32 atomics into a replica of (dW, db)
//         waves 4-7: dx[c, p]    += sum_o W[oc+o, c] sdZ[o][p]      (register tiles, persistent)
//   S7  dA tiles -> HBM (64-byte runs of slab rows); dx through sX -> HBM (whole plane rows)
// ---------------------------------------------------------------------------------------------------------------------
// Diagnostic build only (tools/stamps_planes.py compiles a private copy of the library with -DCG_DOMP_STAMPS): thread 0 of every
// workgroup stores the shader clock at the phase boundaries into a buffer of its own; no result depends on it and no stamp
// executes in the shipped library.
#ifdef CG_DOMP_STAMPS
__device__ unsigned long long* cg_domp_stamp_buf = nullptr;
extern "C" int cg_domp_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(cg_domp_stamp_buf), &p, sizeof(p)); }
#define CG_STAMP()                                                                                      \
  do {                                                                                                  \
    if (threadIdx.x == 0 && cg_domp_stamp_buf && nst < 255) cg_domp_stamp_buf[blockIdx.x * 256 + (++nst)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define CG_STAMP_END() do { if (threadIdx.x == 0 && cg_domp_stamp_buf) cg_domp_stamp_buf[blockIdx.x * 256] = nst; } while (0)
#else
#define CG_STAMP() do { } while (0)
#define CG_STAMP_END() do { } while (0)
#endif

#define CG_DOMPB_THREADS 512
#define CG_DOMPB_NW 8
#define CG_DOMPB_NDX 10      // dx tiles per wave (upper bound over the instantiations)
#define CG_DOMPB_NOC 4       // chunks of 16 output channels (dW tiles per wave)

template <int VWP, int VWY, int VWA, int NJW, int PF, int NDX>
__global__ __launch_bounds__(CG_DOMPB_THREADS, 2) void cg_stgcn_planes_bwd_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                  const float* __restrict__ W, const float* __restrict__ dy,
                                                                  float* __restrict__ dx, float* __restrict__ dadj,
                                                                  float* __restrict__ ws, int replicas, CgDomP g) {
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);       // [CinR][XS]
  float* sdZ = sX + g.CinR * g.XS;                        // [16][XS]; the same memory is the SECOND dY image (odd pieces)
  float* sZ = sdZ + g.dzfl;                               // [V][GZ]   ([16 o][16 t] per joint)
  float* sY = sZ + g.zfl;                                 // [V][GY]   ([16 o][16 q] per joint, column-swizzled): even pieces
  float* sW = sY + g.yfl;                                 // [16][WS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform for the compiler: scalar address arithmetic
  const int l15_ = lane & 15, slot_ = lane >> 4;
  const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
  const int sb = sidx / g.NTC, tc = sidx - sb * g.NTC, b = sb * 8 + xcd;
  if (b >= g.B) return;
  const int T = g.T, V = g.V, Cin = g.Cin, Cout = g.Cout, XS = g.XS, GZ = g.GZ, GY = g.GY, YS = g.YS, WS = g.WS;
  const int t0 = tc * g.TC, TCn = min(g.TC, T - t0), Pn = TCn * V, NPT = (Pn + 15) / 16;
  const int NQC = (T + 15) / 16;
  const long long TV = g.TV;

  int nst = 0; (void)nst;
  CG_STAMP();                                         // 1: start
  {   // the workgroup's piece of x: rows [t0, t0 + TCn) of every plane.  A wave moves 8 vectors per round: 4 planes x 2 vectors
      // per lane (128 vectors per plane row cover nvp <= 128); all requests of a round are in flight before the first LDS store,
      // and the first round travels while the LDS image is zeroed
    const float* xb = x + (long long)b * Cin * TV + (long long)t0 * V;
    const int nvp = Pn / VWP;
    float xr[8][VWP];
    auto x_load = [&](int c0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + (u >> 1), i = lane + 64 * (u & 1);
        cg_domp_ld<VWP>(xb + ((long long)c * TV + i * VWP), c < Cin && i < nvp, xr[u]);
      }
    };
    auto x_store = [&](int c0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + (u >> 1), i = lane + 64 * (u & 1);
        if (c < Cin && i < nvp) cg_domp_st<VWP>(sX + c * XS + i * VWP, xr[u]);
      }
    };
    x_load(4 * wave);
    // zeroed once: the pad columns [Pn, XS) of sX and its channel rows beyond Cin (operands of the dW product and of tiles that
    // reach beyond the piece), sdZ, sZ (frames beyond the chunk stay zero), both dY images (frames beyond T must stay finite)
    {
      const int padc = XS - Pn, nrow = g.CinR;
      for (int e = tid; e < nrow * padc; e += CG_DOMPB_THREADS) {
        const int r = e / padc, c = e - r * padc;
        sX[r * XS + Pn + c] = 0.f;
      }
      for (int e = tid; e < (g.CinR - Cin) * Pn; e += CG_DOMPB_THREADS) sX[(Cin + e / Pn) * XS + e % Pn] = 0.f;
      // sdZ / second dY image, sZ, sY, sW in one sweep (they are contiguous)
      for (int e = tid; e < g.dzfl + g.zfl + g.yfl + 16 * WS; e += CG_DOMPB_THREADS) sdZ[e] = 0.f;
    }
    __syncthreads();
    CG_STAMP();                                       // 2: LDS zeroed
    for (int c0 = 4 * wave; c0 < Cin; c0 += 4 * CG_DOMPB_NW) {
      x_store(c0);
      if (c0 + 4 * CG_DOMPB_NW < Cin) x_load(c0 + 4 * CG_DOMPB_NW);
    }
  }

  // a piece of dY: [16 channels][QCn frames][V] = 16 runs of QCn * V floats; thread (o = tid / 32, k = tid % 32 + 32 i)
  // moves vector k of run o
  float pf[PF][VWY];
  const int po = tid >> 5, pk = tid & 31;
  // pieces are numbered n = oc * NQC + qc; piece n lives in the dY image (n % NQC) & 1 (0: sY, 1: the sdZ memory)
  const int NPIECES = g.NOC * NQC;
  auto piece_load = [&](int n) {
    const int oc = n / NQC, qc = n - oc * NQC;
    const int q0 = 16 * qc, runv = min(16, T - q0) * V / VWY, co = oc * 16 + po;
    const float* src = dy + ((long long)(b * Cout + oc * 16) * T + q0) * V;        // uniform
    const int voff = po * g.TV + pk * VWY;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int k = pk + 32 * i;
      if (k < runv) cg_domp_ld<VWY>(src + (voff + 32 * i * VWY), co < Cout, pf[i]);
    }
  };
  // LDS offsets of the PF * VWY elements a thread moves per piece: they do not depend on the piece (a short last piece uses a
  // prefix of them).  Frames beyond T keep the finite values of an earlier piece: the dZ product multiplies them with slab rows
  // that are masked to zero, the dA product only fills columns q >= T that are never stored, the bias gradient masks them.
  int pso[PF][VWY];
  {
    const int sw = cg_domp_swz(po);
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
      for (int jj = 0; jj < VWY; ++jj) {
        const int idx = (pk + 32 * i) * VWY + jj, q = (int)cg_domp_div((unsigned)idx, g.magicV), v = idx - q * V;
        pso[i][jj] = v * GY + po * 16 + ((q & 15) ^ sw);
      }
  }
  auto piece_store = [&](int n) {
    const int qc = n % NQC, runv = min(16, T - 16 * qc) * V / VWY;
    float* img = (qc & 1) ? sdZ : sY;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (pk + 32 * i < runv) {
#pragma unroll
        for (int jj = 0; jj < VWY; ++jj) img[pso[i][jj]] = pf[i][jj];
      }
    }
  };
  int pfn = 0;                                          // piece held (loaded, not yet stored) by the pf registers; NPIECES: none

  cg_f32x4 dAacc[NJW][4];
#pragma unroll
  for (int a = 0; a < NJW; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) dAacc[a][q] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  cg_f32x4 dxacc[NDX];
#pragma unroll
  for (int i = 0; i < NDX; ++i) dxacc[i] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  cg_f32x4 dWhold = cg_f32x4{0.f, 0.f, 0.f, 0.f};          // slice 0 of the dW tile of the previous chunk, flushed behind the next barrier
  const int MTc = g.CinR / 16, ndx = MTc * NPT;
  // dW: wave -> (channel tile ct, slice ks of the position chunks)
  const int nks = CG_DOMPB_NW / MTc, wct = wave % MTc, wks = wave / MTc;
  float* wsr = ws + (long long)(blockIdx.x % replicas) * ((long long)Cout * Cin + Cout);
  float* sPart = sW + 16 * WS;                            // [wave][64 lanes][4]: dW tiles of the position slices ks > 0
  // adds the slices of output chunk `ocp` (slice 0 in dWhold, the others in sPart) into the replica; call behind a barrier
  auto dw_flush = [&](int ocp) {
    if (wks == 0) {
      cg_f32x4 sum = dWhold;
      for (int ks = 1; ks < nks; ++ks) sum += *reinterpret_cast<const cg_f32x4*>(sPart + ((wct + (ks - 1) * MTc) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = ocp * 16 + 4 * slot_ + r, c = 16 * wct + l15_;
        if (co < Cout && c < Cin) atomicAdd(&wsr[(long long)co * Cin + c], sum[r]);
      }
    }
  };

  // rows [t0, t0 + TCn) of the slabs of this wave's joints, 16 columns q of piece qc: the B operand of the dZ product
  auto rows_load = [&](int qc, float (&dst)[NJW][4]) {
    const int q0 = 16 * qc;
#pragma unroll
    for (int a = 0; a < NJW; ++a) {
      const int v = wave + CG_DOMPB_NW * a;
      const float* rowb = adj + (((long long)b * V + v) * T + t0) * T + q0;              // uniform
      const int voff = l15_ * T + 4 * slot_;
#pragma unroll
      for (int s0 = 0; s0 < 4; s0 += VWA) {
        float tmp[VWA];
        cg_domp_ld<VWA>(rowb + (voff + s0), v < V && l15_ < TCn && q0 + 4 * slot_ + s0 < T, tmp);
#pragma unroll
        for (int i = 0; i < VWA; ++i) dst[a][s0 + i] = tmp[i];
      }
    }
  };
  float ar[NJW][4];
  rows_load(0, ar);
  piece_load(0);
  // weights of the next chunk of output channels travel in registers while the current chunk computes
  float wq[3];
  auto w_load = [&](int oc) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int e = tid + i * CG_DOMPB_THREADS, o = e / WS, c = e - o * WS;
      wq[i] = (e < 16 * WS && oc * 16 + o < Cout && c < Cin) ? W[(long long)(oc * 16 + o) * Cin + c] : 0.f;
    }
  };
  w_load(0);
#pragma nounroll
  for (int oc = 0; oc < g.NOC; ++oc) {
    int l15 = l15_, slot = slot_;
    CG_STAMP();                                       // oc+0: before B0 (first chunk: setup + x piece issued)
    __syncthreads();                                  // B0: the previous chunk's readers of sW / sZ / sY / sdZ are done
    CG_STAMP();                                       // oc+1: after B0
    if (oc > 0) dw_flush(oc - 1);
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (tid + i * CG_DOMPB_THREADS < 16 * WS) sW[tid + i * CG_DOMPB_THREADS] = wq[i];
    if (oc + 1 < g.NOC) w_load(oc + 1);
    if (pfn == oc * NQC) {                             // the chunk's first piece is still in registers (first chunk; one piece per chunk; odd NQC)
      piece_store(pfn);
      if (++pfn < NPIECES) piece_load(pfn);
    }
    cg_f32x4 dZacc[NJW];
#pragma unroll
    for (int a = 0; a < NJW; ++a) dZacc[a] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    float dbp = 0.f;
    CG_STAMP();                                       // oc+2: sW + piece 0 stored
    __syncthreads();                                  // B1
    CG_STAMP();                                       // oc+3

    // S2: two position tiles per pass share the weight fragments; four K steps of operands are in flight at a time
    // (the lane coordinates are made opaque per phase: the LDS addresses derived from them are then recomputed where they are
    // used - a few integer operations - instead of being hoisted out of the loops into ~60 registers that push the accumulators
    // into scratch; scratch reloads count in vmcnt and would make every phase wait for the global prefetches in flight)
    CG_OPAQUE_V(l15); CG_OPAQUE_V(slot);
    for (int pt = wave; pt < NPT; pt += 2 * CG_DOMPB_NW) {
      const bool pair = pt + CG_DOMPB_NW < NPT;                           // the last tiles have no partner: one accumulator then
      const int pt1 = pair ? pt + CG_DOMPB_NW : pt;
      cg_f32x4 acc0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      const float* ap = sW + l15 * WS + slot;
      const float* bp0 = sX + slot * XS + 16 * pt + l15;
      const float* bp1 = sX + slot * XS + 16 * pt1 + l15;
      // the images are zero padded to a multiple of four steps; the operands of the next four steps are read while the
      // matrix cores work on the current ones
      float fa[12], fb[12];
      auto s2_ld = [&](int k0, float (&f)[12]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { f[s2] = ap[4 * (k0 + s2)]; f[4 + s2] = bp0[4 * (k0 + s2) * XS]; f[8 + s2] = bp1[4 * (k0 + s2) * XS]; }
      };
      auto s2_mm = [&](const float (&f)[12]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s2], f[4 + s2], acc0, 0, 0, 0);
          if (pair) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s2], f[8 + s2], acc1, 0, 0, 0);
        }
      };
      s2_ld(0, fa);
      for (int k0 = 0; k0 < g.KS; k0 += 8) {
        if (k0 + 4 < g.KS) s2_ld(k0 + 4, fb);
        s2_mm(fa);
        if (k0 + 4 < g.KS) {
          if (k0 + 8 < g.KS) s2_ld(k0 + 8, fa);
          s2_mm(fb);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pp = 16 * (h ? pt1 : pt) + l15;
        if (pp < Pn && (h == 0 || pt1 != pt)) {
          const int t = (int)cg_domp_div((unsigned)pp, g.magicV), v = pp - t * V;
          float* dst = sZ + v * GZ + 4 * slot * 16 + t;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[r * 16] = h ? acc1[r] : acc0[r];
        }
      }
    }
    CG_STAMP();                                       // oc+4: S2 done
    __syncthreads();                                  // B2
    CG_STAMP();                                       // oc+5

#pragma nounroll
    for (int qc = 0; qc < NQC; ++qc) {
      {
        // one barrier per piece: every wave is done with S4 of the previous piece, so the OTHER dY image is free, and the stores
        // of the current piece (made during that S4) are visible.  The piece in the pf registers (requested a whole S4 ago) goes
        // into the other image now - unless that is the image being read (first piece of the next chunk when NQC is odd or 1:
        // it waits for the top of the next chunk) - and the one after it is requested.
        if (qc > 0) __syncthreads();
        const int n = oc * NQC + qc;
        if (pfn == n + 1 && pfn < NPIECES && (((pfn % NQC) ^ qc) & 1)) {
          piece_store(pfn);
          if (++pfn < NPIECES) piece_load(pfn);
        }
        const float* sYc = (qc & 1) ? sdZ : sY;
        // S4
        const int q0 = 16 * qc;
        CG_OPAQUE_V(l15); CG_OPAQUE_V(slot);
        // no branches around the joints: the three (four) products of a wave interleave on the matrix pipe.  A joint index
        // beyond V (last round of some waves) is clamped: its tiles are computed from joint V - 1 and never stored.
        {
          cg_f32x4 part[NJW];
          {
            float za[NJW][4], ya[NJW][4];
#pragma unroll
            for (int a = 0; a < NJW; ++a) {
              const int v = min(wave + CG_DOMPB_NW * a, V - 1);
              const float* zp = sZ + v * GZ + slot * 16 + l15;
              const float* yp = sYc + v * GY + slot * 16;
#pragma unroll
              for (int st = 0; st < 4; ++st) {
                za[a][st] = zp[4 * st * 16];
                ya[a][st] = yp[4 * st * 16 + (l15 ^ cg_domp_swz(slot + 4 * st))];      // row o = slot + 4 st
              }
            }
#pragma unroll
            for (int a = 0; a < NJW; ++a) part[a] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
              for (int a = 0; a < NJW; ++a) part[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[a][st], ya[a][st], part[a], 0, 0, 0);
          }
          {
            float yb[NJW][4];
            const int sw = cg_domp_swz(l15);
#pragma unroll
            for (int a = 0; a < NJW; ++a) {
              const int v = min(wave + CG_DOMPB_NW * a, V - 1);
              const float* yq = sYc + v * GY + l15 * 16;
#pragma unroll
              for (int st = 0; st < 4; ++st) yb[a][st] = yq[(4 * slot + st) ^ sw];
            }
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
              for (int a = 0; a < NJW; ++a) {
                dZacc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(yb[a][st], ar[a][st], dZacc[a], 0, 0, 0);
                dbp += (q0 + 4 * slot + st < T && wave + CG_DOMPB_NW * a < V) ? yb[a][st] : 0.f;
              }
          }
#pragma unroll
          for (int a = 0; a < NJW; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (qc == k) dAacc[a][k] += part[a];    // register tiles need compile-time indices
        }
        // slab rows of the next piece: requested behind the dY piece (vmcnt retires in order: the piece is waited for first, by
        // piece_store, and these rows are not needed before the next S4 has done its dA products)
        rows_load(qc + 1 < NQC ? qc + 1 : 0, ar);
        CG_STAMP();                                   // S4 of the piece done
      }
    }
    __syncthreads();                                  // every S4 of the chunk is done: the sdZ memory may take dZ now

    // S5
    CG_OPAQUE_V(l15); CG_OPAQUE_V(slot);
#pragma unroll
    for (int a = 0; a < NJW; ++a) {
      const int v = wave + CG_DOMPB_NW * a;
      if (v < V && l15 < TCn) {
        float* dst = sdZ + 4 * slot * XS + l15 * V + v;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r * XS] = dZacc[a][r];
      }
    }
    if (tc == 0) {                                    // the bias gradient is counted by one chunk of frames per sample
      dbp += __shfl_xor(dbp, 16, 64);
      dbp += __shfl_xor(dbp, 32, 64);
      if (slot == 0 && oc * 16 + l15 < Cout) atomicAdd(&wsr[(long long)Cout * Cin + oc * 16 + l15], dbp);
    }
    CG_STAMP();                                       // S5 done
    __syncthreads();                                  // B3
    CG_STAMP();

    // S6: dW (a slice of the positions per wave) and dx (tiles dealt to the waves)
    CG_OPAQUE_V(l15); CG_OPAQUE_V(slot);
    if (wks < nks) {
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f};
      const float* ap = sdZ + l15 * XS + 4 * slot;
      const float* bp = sX + (16 * wct + l15) * XS + 4 * slot;
      for (int kc = wks; kc < NPT; kc += nks) {
        cg_f32x4 a0 = *reinterpret_cast<const cg_f32x4*>(ap + 16 * kc);
        const cg_f32x4 b0 = *reinterpret_cast<const cg_f32x4*>(bp + 16 * kc);
        if (16 * kc + 16 > Pn) {                      // beyond the chunk's positions the row stride wraps into the next row
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) a0[s2] = (16 * kc + 4 * slot + s2 < Pn) ? a0[s2] : 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s2], b0[s2], c0, 0, 0, 0);
      }
      if (wks == 0) dWhold = c0;
      else *reinterpret_cast<cg_f32x4*>(sPart + ((wave - MTc) * 64 + lane) * 4) = c0;
    }
    {
      // dx: the channel tile of a wave is fixed (8 % MTc == 0), its weight fragments are read once per chunk
      const int ct = wave % MTc;
      const float* ap = sW + slot * WS + 16 * ct + l15;
      float aw[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) aw[st] = ap[4 * st * WS];
#pragma unroll
      for (int i = 0; i < NDX; ++i) {
        const int id = wave + CG_DOMPB_NW * i;
        if (id < ndx) {                               // uniform
          const float* bp = sdZ + slot * XS + 16 * (id / MTc) + l15;
#pragma unroll
          for (int st = 0; st < 4; ++st) dxacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[st], bp[4 * st * XS], dxacc[i], 0, 0, 0);
        }
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four tiles' operands in flight, not all ten (register budget)
      }
    }
  }

  // S7
  const int l15 = l15_, slot = slot_;
  CG_STAMP();                                         // last S6 done
  __syncthreads();
  CG_STAMP();
  dw_flush(g.NOC - 1);
#pragma unroll
  for (int i = 0; i < NDX; ++i) {
    const int id = wave + CG_DOMPB_NW * i;
    if (id < ndx) {
      const int pt = id / MTc, ct = id - pt * MTc;
      float* dst = sX + (16 * ct + 4 * slot) * XS + 16 * pt + l15;
      if (16 * pt + l15 < Pn) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r * XS] = dxacc[i][r];
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NJW; ++a) {
    const int v = wave + CG_DOMPB_NW * a;
    if (v < V) {
#pragma unroll
      for (int qc = 0; qc < 4; ++qc) {
        const int q = 16 * qc + l15;
        if (qc < NQC && q < T) {
          float* db_ = dadj + (((long long)b * V + v) * T + t0) * T;                     // uniform
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int t = 4 * slot + r;
            if (t < TCn) db_[t * T + q] = dAacc[a][qc][r];
          }
        }
      }
    }
  }
  __syncthreads();
  {
    float* dxb = dx + (long long)b * Cin * TV + (long long)t0 * V;
    const int nvp = Pn / VWP;
    for (int c0 = 4 * wave; c0 < Cin; c0 += 4 * CG_DOMPB_NW) {
      float xr[8][VWP];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + (u >> 1), i = lane + 64 * (u & 1);
        if (c < Cin && i < nvp) {
          if constexpr (VWP == 4) {
            const cg_f32x4 t = *reinterpret_cast<const cg_f32x4*>(sX + c * XS + i * VWP);
#pragma unroll
            for (int jj = 0; jj < VWP; ++jj) xr[u][jj] = t[jj];
          } else {
#pragma unroll
            for (int jj = 0; jj < VWP; ++jj) xr[u][jj] = sX[c * XS + i * VWP + jj];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + (u >> 1), i = lane + 64 * (u & 1);
        if (c < Cin && i < nvp) cg_domp_st<VWP>(dxb + ((long long)c * TV + i * VWP), xr[u]);
      }
    }
  }
  CG_STAMP();
  CG_STAMP_END();
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
    hipError_t e = cg_lds_limit((const void*)cg_stgcn_planes_fwd_kernel<D, VW_, VWB_, NL_>,                                \
                                       lds);                                       \
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

// ---------------------------------------------------------------------------------------------------------------------
// backward, time domain (Adj (B,T,V,V), one V x V slab per frame)
//
//   dZ[o,t,v] = sum_w dY[o,t,w] A[t,v,w]        dA[t,v,w] = sum_o Z[o,t,v] dY[o,t,w]        dx = W^T dZ   dW = dZ x^T   db = sum dY
// Frames are independent and a chunk of frames is a contiguous piece of every (T,V) plane, so a workgroup owns (sample b, chunk
// of TC = 8 frames) and everything it touches is local: the pieces of x and dY ([channels][TC*V], whole plane rows) and Z = W . x
// sit in LDS, one wave owns one frame for the two slab products (dZ_t replaces Z_t in place, no barrier inside the phase), and
// dx / dW read dZ back as matrix-core operands.  Four barriers per workgroup.
//   P1  Z[o,p]   = W . sX                       -> sZ[o][p]
//   P2  per frame t (one wave):  dA_t = Z_t^T . dY_t  -> HBM;   dZ_t = dY_t . A_t^T  -> sZ (over Z_t);   db from the dY fragments
//   P3  dx[c,p]  = W^T . sZ                     -> sDY (dY is dead) -> HBM as whole plane rows
//       dW[o,c] += sZ . sX^T                    -> fp32 atomics into a replica of (dW, db)
// ---------------------------------------------------------------------------------------------------------------------
#define CG_DOMPT_THREADS 512
#define CG_DOMPT_NW 8

template <int VWP, int VWA>
__global__ __launch_bounds__(CG_DOMPT_THREADS, 2) void cg_stgcn_planes_bwd_time_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                       const float* __restrict__ W, const float* __restrict__ dy,
                                                                       float* __restrict__ dx, float* __restrict__ dadj,
                                                                       float* __restrict__ ws, int replicas, CgDomP g) {
  const int T = g.T, V = g.V, Cin = g.Cin, Cout = g.Cout, XS = g.XS, WS = g.WS, CinR = g.CinR, CoR = 16 * g.NOC;
  const int DR = max(CinR, CoR);                          // the dY image takes dx in P3
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);       // [CinR][XS]
  float* sDY = sX + CinR * XS;                            // [DR][XS]
  float* sZ = sDY + DR * XS;                              // [CoR][XS]
  float* sW = sZ + CoR * XS;                              // [CoR][WS]
  float* sdb = sW + CoR * WS;                             // [CoR]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
  const int sb = sidx / g.NTC, tc = sidx - sb * g.NTC, b = sb * 8 + xcd;
  if (b >= g.B) return;
  const int t0 = tc * g.TC, TCn = min(g.TC, T - t0), Pn = TCn * V, NPT = (Pn + 15) / 16;
  const long long TV = g.TV;
  const int MTc = CinR / 16, MTo = g.NOC;

  int nst = 0; (void)nst;
  CG_STAMP();
  // P0: pieces of x and dY: a wave owns 16 plane rows, all of their requests are in flight while the pads of the LDS image are
  // zeroed (columns beyond the piece, rows beyond the channel counts, the weight image); one vector per lane and plane row
  {
    const int nvp = Pn / VWP;            // vectors per plane row (<= 64)
    const float* xb = x + (long long)b * Cin * TV + (long long)t0 * V;
    const float* yb = dy + (long long)b * Cout * TV + (long long)t0 * V;
    const int rows = Cin + Cout;
    float buf[16][VWP];
    const int r0 = 16 * wave;                        // rows <= 128 = 16 rows x 8 waves
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = r0 + u;
      const float* src = r < Cin ? xb + (long long)r * TV : yb + (long long)(r - Cin) * TV;
      cg_domp_ld<VWP>(src + lane * VWP, r < rows && lane < nvp, buf[u]);
    }
    const int padc = XS - Pn, nrow = CinR + DR + CoR;
    for (int e = tid; e < nrow * padc; e += CG_DOMPT_THREADS) {
      const int r = e / padc, c = e - r * padc;
      sX[r * XS + Pn + c] = 0.f;
    }
    for (int e = tid; e < (CinR - Cin) * Pn; e += CG_DOMPT_THREADS) sX[Cin * XS + (e / Pn) * XS + e % Pn] = 0.f;
    for (int e = tid; e < (DR - Cout) * Pn; e += CG_DOMPT_THREADS) sDY[Cout * XS + (e / Pn) * XS + e % Pn] = 0.f;
    for (int e = tid; e < CoR * (WS + 1); e += CG_DOMPT_THREADS) sW[e] = 0.f;
    __syncthreads();
    CG_STAMP();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = r0 + u;
      float* dst = r < Cin ? sX + r * XS : sDY + (r - Cin) * XS;
      if (r < rows && lane < nvp) {                   // (the row stride is == 2 mod 4: no vector stores)
#pragma unroll
        for (int jj = 0; jj < VWP; ++jj) dst[lane * VWP + jj] = buf[u][jj];
      }
    }
    for (int e = tid; e < Cout * Cin; e += CG_DOMPT_THREADS) {
      const int o = e / Cin, c = e - o * Cin;
      sW[o * WS + c] = W[e];
    }
  }
  CG_STAMP();
  __syncthreads();
  CG_STAMP();

  // P1: Z = W . sX, units of (two channel tiles, one position tile)
  {
    const int OP = (MTo + 1) / 2, nun = OP * NPT;
    for (int u = wave; u < nun; u += CG_DOMPT_NW) {
      const int pt = u / OP, op = u - pt * OP, ot0 = 2 * op, ot1 = ot0 + 1 < MTo ? ot0 + 1 : ot0;
      cg_f32x4 a0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const float* ap0 = sW + (16 * ot0 + l15) * WS + slot;
      const float* ap1 = sW + (16 * ot1 + l15) * WS + slot;
      const float* bp = sX + slot * XS + 16 * pt + l15;
      float f0[12], f1[12];
      auto ld = [&](int k0, float (&f)[12]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { f[s2] = ap0[4 * (k0 + s2)]; f[4 + s2] = ap1[4 * (k0 + s2)]; f[8 + s2] = bp[4 * (k0 + s2) * XS]; }
      };
      auto mm = [&](const float (&f)[12]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s2], f[8 + s2], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[4 + s2], f[8 + s2], a1, 0, 0, 0);
        }
      };
      const int KS4 = CinR / 4;                       // steps, multiple of 4 (rows of sX beyond Cin are zero)
      ld(0, f0);
      for (int k0 = 0; k0 < KS4; k0 += 8) {
        if (k0 + 4 < KS4) ld(k0 + 4, f1);
        mm(f0);
        if (k0 + 4 < KS4) {
          if (k0 + 8 < KS4) ld(k0 + 8, f0);
          mm(f1);
        }
      }
      if (16 * pt + l15 < Pn) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sZ[(16 * ot0 + 4 * slot + r) * XS + 16 * pt + l15] = a0[r];
          if (ot1 != ot0) sZ[(16 * ot1 + 4 * slot + r) * XS + 16 * pt + l15] = a1[r];
        }
      }
    }
  }
  CG_STAMP();
  __syncthreads();
  CG_STAMP();

  // P2: one wave per frame
  float dbw[4] = {0.f, 0.f, 0.f, 0.f};                  // bias-gradient partial sums of this lane (channel 16 i + l15) over its frames
  for (int tl = wave; tl < TCn; tl += CG_DOMPT_NW) {
    const int pb = tl * V;                            // first position of the frame inside the pieces
    const int NVT = (V + 15) / 16;                    // 1 or 2 tiles of joints
    // slab rows of the frame for the dZ product, requested before the dA product runs
    const float* at = adj + (((long long)b * T + t0 + tl) * V) * V;
    float bv[2][2][4];                               // [chunk of 16 columns w][tile of joints v][step]
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int v = 16 * j + l15, w0 = 16 * kc + 4 * slot;
#pragma unroll
        for (int s0 = 0; s0 < 4; s0 += VWA) {
          float tmp[VWA];
          cg_domp_ld<VWA>(at + v * V + w0 + s0, kc < NVT && j < NVT && v < V && w0 + s0 < V, tmp);
#pragma unroll
          for (int q = 0; q < VWA; ++q) bv[kc][j][s0 + q] = tmp[q];
        }
      }
    // dA_t[v, w] = sum_o Z[o, pb + v] dY[o, pb + w]
    {
      cg_f32x4 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
      const float* zp = sZ + slot * XS + pb + l15;
      const float* yp = sDY + slot * XS + pb + l15;
      float fa[16], fb[16];                           // four steps x (two Z tiles, two dY tiles)
      auto ld = [&](int k0, float (&f)[16]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
          for (int i = 0; i < 2; ++i) { f[4 * s2 + i] = zp[4 * (k0 + s2) * XS + 16 * i]; f[4 * s2 + 2 + i] = yp[4 * (k0 + s2) * XS + 16 * i]; }
      };
      auto mm = [&](const float (&f)[16]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (i < NVT && j < NVT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[4 * s2 + i], f[4 * s2 + 2 + j], acc[i][j], 0, 0, 0);
      };
      const int KS4 = CoR / 4;
      ld(0, fa);
      for (int k0 = 0; k0 < KS4; k0 += 8) {
        if (k0 + 4 < KS4) ld(k0 + 4, fb);
        mm(fa);
        if (k0 + 4 < KS4) {
          if (k0 + 8 < KS4) ld(k0 + 8, fa);
          mm(fb);
        }
      }
      float* da = dadj + (((long long)b * T + t0 + tl) * V) * V;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int w = 16 * j + l15;
          if (i < NVT && j < NVT && w < V) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int v = 16 * i + 4 * slot + r;
              if (v < V) da[v * V + w] = acc[i][j][r];
            }
          }
        }
    }
    // dZ_t[o, v] = sum_w dY[o, pb + w] A_t[v, w]; inside a chunk of 16 columns step s takes w = 16 kc + 4 slot + s on both operands
    {
      cg_f32x4 acc[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
      float dbp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        if (kc < NVT) {
          const int w0 = 16 * kc + 4 * slot;
          float av[4][4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) av[i][s2] = (i < MTo && w0 + s2 < V) ? sDY[(16 * i + l15) * XS + pb + w0 + s2] : 0.f;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              dbp[i] += av[i][s2];
#pragma unroll
              for (int j = 0; j < 2; ++j)
                if (i < MTo && j < NVT) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s2], bv[kc][j][s2], acc[i][j], 0, 0, 0);
            }
        }
      }
      // every read of Z_t by this wave is behind us: dZ_t goes in place
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int v = 16 * j + l15;
          if (i < MTo && j < NVT && v < V) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sZ[(16 * i + 4 * slot + r) * XS + pb + v] = acc[i][j][r];
          }
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) dbw[i] += dbp[i];       // a lane's output channels are the same in every frame: registers until the frames are done
    }
  }
  // (an LDS atomic per channel tile and FRAME used to sit inside the loop: ~1000 cycles each under load)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float sdbp = dbw[i];
    sdbp += __shfl_xor(sdbp, 16, 64);
    sdbp += __shfl_xor(sdbp, 32, 64);
    if (i < MTo && slot == 0) atomicAdd(&sdb[16 * i + l15], sdbp);
  }
  CG_STAMP();
  __syncthreads();
  CG_STAMP();

  // P3: dx -> sDY (dY is dead), dW -> atomics
  {
    const int CP = (MTc + 1) / 2, nun = CP * NPT;
    for (int u = wave; u < nun; u += CG_DOMPT_NW) {
      const int pt = u / CP, cp = u - pt * CP, ct0 = 2 * cp, ct1 = ct0 + 1 < MTc ? ct0 + 1 : ct0;
      cg_f32x4 a0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const float* ap0 = sW + slot * WS + 16 * ct0 + l15;
      const float* ap1 = sW + slot * WS + 16 * ct1 + l15;
      const float* bp = sZ + slot * XS + 16 * pt + l15;
      float f0[12], f1[12];
      auto ld = [&](int k0, float (&f)[12]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { f[s2] = ap0[4 * (k0 + s2) * WS]; f[4 + s2] = ap1[4 * (k0 + s2) * WS]; f[8 + s2] = bp[4 * (k0 + s2) * XS]; }
      };
      auto mm = [&](const float (&f)[12]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s2], f[8 + s2], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[4 + s2], f[8 + s2], a1, 0, 0, 0);
        }
      };
      const int KS4 = CoR / 4;
      ld(0, f0);
      for (int k0 = 0; k0 < KS4; k0 += 8) {
        if (k0 + 4 < KS4) ld(k0 + 4, f1);
        mm(f0);
        if (k0 + 4 < KS4) {
          if (k0 + 8 < KS4) ld(k0 + 8, f0);
          mm(f1);
        }
      }
      if (16 * pt + l15 < Pn) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sDY[(16 * ct0 + 4 * slot + r) * XS + 16 * pt + l15] = a0[r];
          if (ct1 != ct0) sDY[(16 * ct1 + 4 * slot + r) * XS + 16 * pt + l15] = a1[r];
        }
      }
    }
    // dW[o, c] = sum_p dZ[o, p] x[c, p]: tiles dealt to the waves, two at a time sharing the dZ fragments
    float* wsr = ws + (long long)(blockIdx.x % replicas) * ((long long)Cout * Cin + Cout);
    const int CP2 = (MTc + 1) / 2;
    for (int u = wave; u < MTo * CP2; u += CG_DOMPT_NW) {
      const int ot = u / CP2, cp = u - ot * CP2, ct0 = 2 * cp, ct1 = ct0 + 1 < MTc ? ct0 + 1 : ct0;
      cg_f32x4 a0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const float* ap = sZ + (16 * ot + l15) * XS + 4 * slot;
      const float* bp0 = sX + (16 * ct0 + l15) * XS + 4 * slot;
      const float* bp1 = sX + (16 * ct1 + l15) * XS + 4 * slot;
      float fa[12], fb[12];
      auto ld = [&](int kc, float (&f)[12]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          const bool ok = 16 * kc + 4 * slot + s2 < Pn;      // beyond the piece the row stride wraps into the next row
          f[s2] = ok ? ap[16 * kc + s2] : 0.f; f[4 + s2] = bp0[16 * kc + s2]; f[8 + s2] = bp1[16 * kc + s2];
        }
      };
      auto mm = [&](const float (&f)[12]) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s2], f[4 + s2], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s2], f[8 + s2], a1, 0, 0, 0);
        }
      };
      ld(0, fa);
      for (int kc = 0; kc < NPT; kc += 2) {
        if (kc + 1 < NPT) ld(kc + 1, fb);
        mm(fa);
        if (kc + 1 < NPT) {
          if (kc + 2 < NPT) ld(kc + 2, fa);
          mm(fb);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * ot + 4 * slot + r;
        if (co < Cout && 16 * ct0 + l15 < Cin) atomicAdd(&wsr[(long long)co * Cin + 16 * ct0 + l15], a0[r]);
        if (ct1 != ct0 && co < Cout && 16 * ct1 + l15 < Cin) atomicAdd(&wsr[(long long)co * Cin + 16 * ct1 + l15], a1[r]);
      }
    }
    if (tid < Cout) atomicAdd(&wsr[(long long)Cout * Cin + tid], sdb[tid]);
  }
  CG_STAMP();
  __syncthreads();
  CG_STAMP();
  {
    float* dxb = dx + (long long)b * Cin * TV + (long long)t0 * V;
    const int nvp = Pn / VWP;
    for (int r0 = 8 * wave; r0 < Cin; r0 += 8 * CG_DOMPT_NW) {
      float buf[8][VWP];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u;
        if (r < Cin && lane < nvp) {
#pragma unroll
          for (int jj = 0; jj < VWP; ++jj) buf[u][jj] = sDY[r * XS + lane * VWP + jj];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u;
        if (r < Cin && lane < nvp) cg_domp_st<VWP>(dxb + ((long long)r * TV + lane * VWP), buf[u]);
      }
    }
  }
  CG_STAMP();
  CG_STAMP_END();
}

// row stride of the x piece and of sdZ: == 4 (mod 8) floats, so that the 16 rows of a float4 fragment read (dW product) start in 16
// different 4-bank groups; tiles that reach beyond the chunk's positions wrap into the next row and are masked where that matters
static int cg_domp_xs(int p) { int xs = cg_domp_up(p, 4); return (xs % 8 == 4) ? xs : xs + 4; }

// geometry of the backward: frames per chunk, LDS images
int cg_domp_bwd_geom(CgDomP& g, int B, int Cin, int Cout, int T, int V) {
  int st = cg_domp_geom(g, B, Cin, Cout, T, V, 0);
  if (st != CG_OK) return st;
  if (Cin > 64 || T > 64) return CG_ESHAPE;       // register tiles: dx, 4 pieces of 16 frames
  g.CinR = cg_domp_up(Cin, 16);
  g.YS = 16; g.GZ = 16 * 16 + 1; g.GY = 16 * 16 + 1;
  g.zfl = cg_domp_up(V * g.GZ, 4); g.yfl = cg_domp_up(V * g.GY, 4);
  g.WS = cg_domp_up(g.WS, 2);
  const size_t limit = 160 * 1024 - 1024;
  for (int ntc = (T + 15) / 16; ntc <= T; ++ntc) {
    const int tc0 = (T + ntc - 1) / ntc;
    int best = 0, best_al = 0;
    for (int tc = tc0; tc <= 16 && tc <= tc0 + 2; ++tc) {
      if ((long long)(ntc - 1) * tc >= T) continue;                 // the last chunk would be empty
      const int last = T - (ntc - 1) * tc;
      const int al = ((tc * V) % 4 == 0 && (last * V) % 4 == 0 && g.TV % 4 == 0) ? 4 : ((tc * V) % 2 == 0 && (last * V) % 2 == 0 && g.TV % 2 == 0) ? 2 : 1;
      const int xs = cg_domp_xs(tc * V);
      const int dzfl = 16 * xs > g.yfl ? 16 * xs : g.yfl;          // sdZ [16][XS] / second dY image
      const size_t bytes = ((size_t)g.CinR * xs + dzfl + g.zfl + g.yfl + 16 * (size_t)g.WS + CG_DOMPB_NW * 256 + 64) * sizeof(float);
      if (bytes > limit || (g.CinR / 16) * ((tc * V + 15) / 16) > CG_DOMPB_NDX * CG_DOMPB_NW || tc * V > 128 * al) continue;      // x piece: two vectors per lane and plane row
      if (al > best_al) { best = tc; best_al = al; }
    }
    if (best) {
      g.NTC = ntc; g.TC = best; g.VWP = best_al;
      g.XS = cg_domp_xs(best * V);
      g.dzfl = 16 * g.XS > g.yfl ? 16 * g.XS : g.yfl;
      g.bwd_floats = g.CinR * g.XS + g.dzfl + g.zfl + g.yfl + 16 * g.WS + CG_DOMPB_NW * 256 + 64;
      const int lastq = T - 16 * ((T + 15) / 16 - 1);
      g.VWY = (g.TV % 4 == 0 && (lastq * V) % 4 == 0) ? 4 : (g.TV % 2 == 0 && (lastq * V) % 2 == 0) ? 2 : 1;
      g.VWA = T % 4 == 0 ? 4 : T % 2 == 0 ? 2 : 1;
      return CG_OK;
    }
  }
  return CG_ESHAPE;
}

// launches the backward kernel; `ws` holds `replicas` zeroed copies of (dW, db), folded by the caller
int cg_domp_bwd_launch(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj, float* ws,
                       int replicas, int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream) {
  if (domain != 0) return CG_ESHAPE;
  CgDomP g;
  int st = cg_domp_bwd_geom(g, B, Cin, Cout, T, V);
  if (st != CG_OK) return st;
  const size_t lds = (size_t)g.bwd_floats * sizeof(float);
  const int njw = (V + CG_DOMPB_NW - 1) / CG_DOMPB_NW;
  dim3 grid((unsigned)(8 * ((B + 7) / 8) * g.NTC)), block(CG_DOMPB_THREADS);
  const int pfn = (16 * V / g.VWY + 31) / 32;          // vectors per thread of one dY piece
  if (CG_DOMPB_NW % (g.CinR / 16) != 0 || 16 * g.WS > 3 * CG_DOMPB_THREADS) return CG_ESHAPE;      // a wave keeps one channel tile of dx; weight chunk in three registers per thread
  const int ndxw = ((g.CinR / 16) * ((g.TC * V + 15) / 16) + CG_DOMPB_NW - 1) / CG_DOMPB_NW;      // dx tiles per wave
#define CG_DOMP_BWD(VWP_, VWY_, VWA_, NJW_, PF_, NDX_)                                                                             \
  if (g.VWP == VWP_ && g.VWY == VWY_ && g.VWA == VWA_ && njw <= NJW_ && pfn <= PF_ && ndxw <= NDX_) {                             \
    hipError_t e = cg_lds_limit((const void*)cg_stgcn_planes_bwd_kernel<VWP_, VWY_, VWA_, NJW_, PF_, NDX_>,                \
                                       lds);                                      \
    if (e != hipSuccess) return (int)e;                                                                                            \
    hipLaunchKernelGGL((cg_stgcn_planes_bwd_kernel<VWP_, VWY_, VWA_, NJW_, PF_, NDX_>), grid, block, lds, stream, x, adj, W, dy,  \
                       dx, dadj, ws, replicas, g);                                                                                 \
    return cg_launch_status();                                                                                                     \
  }
  CG_DOMP_BWD(4, 4, 2, 3, 3, 10)     // T*V % 4 == 0, T even, V <= 24  (H3.6M 22 joints, AMASS 18), up to 64 input channels
  CG_DOMP_BWD(2, 2, 2, 4, 7, 6)      // V = 25, up to 32 input channels
#undef CG_DOMP_BWD
  return CG_ESHAPE;
}

// backward of the time domain: frames per chunk and LDS image
int cg_domp_bwd_time_launch(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj, float* ws,
                            int replicas, int B, int Cin, int Cout, int T, int V, hipStream_t stream) {
  CgDomP g;
  int st = cg_domp_geom(g, B, Cin, Cout, T, V, 1);
  if (st != CG_OK) return st;
  if (Cin > 64 || Cout > 64 || V > 32) return CG_ESHAPE;            // four channel tiles, two joint tiles per frame
  g.CinR = cg_domp_up(Cin, 16);
  const int CoR = 16 * g.NOC;
  g.WS = cg_domp_up(g.CinR, 4) + 2;
  g.TC = CG_DOMPT_NW;                                               // one frame per wave
  if (g.TC > T) g.TC = T;
  g.NTC = (T + g.TC - 1) / g.TC;
  const int last = T - (g.NTC - 1) * g.TC;
  g.VWP = ((g.TC * V) % 4 == 0 && (last * V) % 4 == 0 && g.TV % 4 == 0) ? 4 : ((g.TC * V) % 2 == 0 && (last * V) % 2 == 0 && g.TV % 2 == 0) ? 2 : 1;
  if (g.TC * V > 64 * g.VWP || Cin + Cout > 16 * CG_DOMPT_NW) return CG_ESHAPE;      // one vector per lane and plane row, 16 plane rows per wave
  g.VWA = V % 4 == 0 ? 4 : V % 2 == 0 ? 2 : 1;
  g.XS = g.TC * V + ((2 - g.TC * V) & 3);                           // == 2 (mod 4): see JS
  if (g.XS < cg_domp_up(g.TC * V, 16)) g.XS += 4 * ((cg_domp_up(g.TC * V, 16) - g.XS + 3) / 4);   // tiles reach up to a multiple of 16 positions
  g.bwd_floats = (g.CinR + (g.CinR > CoR ? g.CinR : CoR) + CoR) * g.XS + CoR * (g.WS + 1) + 64;
  const size_t lds = (size_t)g.bwd_floats * sizeof(float);
  if (lds > 160 * 1024 - 512) return CG_ESHAPE;
  dim3 grid((unsigned)(8 * ((B + 7) / 8) * g.NTC)), block(CG_DOMPT_THREADS);
#define CG_DOMP_BWDT(VWP_, VWA_)                                                                                                   \
  if (g.VWP == VWP_ && g.VWA == VWA_) {                                                                                            \
    hipError_t e = cg_lds_limit((const void*)cg_stgcn_planes_bwd_time_kernel<VWP_, VWA_>,                                   \
                                       lds);                                      \
    if (e != hipSuccess) return (int)e;                                                                                            \
    hipLaunchKernelGGL((cg_stgcn_planes_bwd_time_kernel<VWP_, VWA_>), grid, block, lds, stream, x, adj, W, dy, dx, dadj, ws,       \
                       replicas, g);                                                                                               \
    return cg_launch_status();                                                                                                     \
  }
  CG_DOMP_BWDT(4, 2)      // V = 22 / 18
  CG_DOMP_BWDT(2, 1)      // V = 25
#undef CG_DOMP_BWDT
  return CG_ESHAPE;
}
