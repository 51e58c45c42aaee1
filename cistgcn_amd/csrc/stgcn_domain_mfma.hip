// Matrix-core kernels of the fused ST-GCN stage (reference: Domain_GCNN_layer.forward, CISTGCN.py:265-266, with
// ConvTemporalGraphical.forward :122-124 and the 1x1 `tcn` convolution :229-234) for wide layers: every product of the
// stage runs on v_mfma_f32_16x16x4_f32 (exact f32 fma chains), one persistent 512-thread workgroup per CU.
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
//   sW [CoM][WS]                   mixing weights
// Row strides are == 4 (mod 8): the k-strided fragment reads (one float per lane per MFMA, `cg_frag<1>`) are bank-conflict
// free and the k-contiguous ones (one ds_read_b128 per four MFMAs, `cg_frag<0>`) see a 2-way conflict (tools: bank check in
// DESIGN.md).  Inside a 16-wide k chunk the four MFMA steps take k = 4*slot + step (slot = lane / 16): the same permutation
// on both operands, so a lane's four values are contiguous.
#include "cg_common.h"
#include "stgcn_domain.h"

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
__device__ __forceinline__ long long cg_domm_off(const CgDomM& g, int domain, int grp_abs, int j) {
  return domain == 1 ? (long long)grp_abs * g.V + j : (long long)j * g.V + grp_abs;
}

// stage a [C][GT][J] tile of a contiguous (B,C,T,V) tensor into a [C][RS] image (zeros for groups >= ng)
template <int DOMAIN>
__device__ __forceinline__ void cg_domm_stage_tile(const CgDomM& g, const float* __restrict__ src, int C, int g0, int ng, float* dst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const long long TV = (long long)g.T * g.V;
  const int run = g.GT * g.J;
  for (int c = wave; c < C; c += nw) {
    const float* sc = src + (long long)c * TV;
    float* dc = dst + c * g.RS;
    for (int r = lane; r < run; r += 64) {
      int grp, j;
      if (DOMAIN == 1) { grp = (int)cg_fastdiv((unsigned)r, g.magicJ); j = r - grp * g.J; }        // j fastest: contiguous in HBM
      else { j = (int)cg_fastdiv((unsigned)r, g.magicGT); grp = r - j * g.GT; }                      // joint fastest
      dc[grp * g.Js + j] = grp < ng ? sc[cg_domm_off(g, DOMAIN, g0 + grp, j)] : 0.f;
    }
  }
}

template <int DOMAIN, int MAXW>
__global__ __launch_bounds__(512) void cg_stgcn_domain_bwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ adj,
                                                                       const float* __restrict__ W, const float* __restrict__ dy,
                                                                       float* __restrict__ dx, float* __restrict__ dadj,
                                                                       float* __restrict__ ws, int replicas, CgDomM g) {
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);
  float* sDY = sX + g.CiM * g.RS;
  float* sBuf = sDY + g.CoM * g.RS;
  float* sA = sBuf + g.CiM * g.RS;
  float* sW = sA + g.GT * g.Jr * g.Jsa;
  const int lds_floats = (2 * g.CiM + g.CoM) * g.RS + g.GT * g.Jr * g.Jsa + g.CoM * g.WS;

  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6, l15 = lane & 15, slot = lane >> 4;
  // XCD-aware order (placement affects speed only): hardware blocks b and b+8 share an L2; every XCD walks a contiguous
  // range of workgroups and every workgroup a contiguous range of tiles, so the tiles of one sample - which touch the same
  // cache lines of x, dy and dx in the space domain - meet in one L2, close in time
  const int chunk = gridDim.x / 8;
  const int wg = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
  const int nwg = (g.total + g.per - 1) / g.per;
  if (wg >= nwg) return;

  for (int e = tid; e < lds_floats; e += nt) sX[e] = 0.f;
  __syncthreads();
  for (int e = tid; e < g.Cout * g.Cin; e += nt) {
    const int co = e / g.Cin, ci = e - co * g.Cin;
    sW[co * g.WS + ci] = W[e];
  }

  const int MTi = g.CiM / 16, MTo = g.CoM / 16, NT = g.Jr / 16, NP = (NT + 1) / 2;
  const int NTp = (g.GT * g.Js + 15) / 16, NPp = (NTp + 1) / 2;
  const long long TV = (long long)g.T * g.V;

  cg_f32x4 wacc[MAXW];
#pragma unroll
  for (int u = 0; u < MAXW; ++u) wacc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  float bacc[MAXW];
#pragma unroll
  for (int u = 0; u < MAXW; ++u) bacc[u] = 0.f;

  for (int it = 0; it < g.per; ++it) {
    const int lid = wg * g.per + it;
    if (lid >= g.total) break;                   // uniform across the workgroup
    const int b = lid / g.ntiles, tile = lid - b * g.ntiles;
    const int g0 = tile * g.GT, ng = min(g.GT, g.NG - g0);
    __syncthreads();                             // previous tile fully consumed
    cg_domm_stage_tile<DOMAIN>(g, x + (long long)b * g.Cin * TV, g.Cin, g0, ng, sX);
    cg_domm_stage_tile<DOMAIN>(g, dy + (long long)b * g.Cout * TV, g.Cout, g0, ng, sDY);
    {
      const float* ab = adj + ((long long)b * g.NG + g0) * g.J * g.J;       // contiguous (ng, J, J)
      const int n = g.GT * g.J * g.J;
      for (int e = tid; e < n; e += nt) {
        const int r = (int)cg_fastdiv((unsigned)e, g.magicJ), o = e - r * g.J;      // r = grp*J + j
        const int grp = (int)cg_fastdiv((unsigned)r, g.magicJ), j = r - grp * g.J;
        sA[(grp * g.Jr + j) * g.Jsa + o] = grp < ng ? ab[e] : 0.f;
      }
    }
    __syncthreads();

    // P1: G[ci][grp, o] = sum_j X[ci][grp, j] A[grp][j][o]
    for (int w = wave; w < g.GT * MTi * NP; w += nw) {
      const int np = w % NP, r = w / NP, mt = r % MTi, grp = r / MTi;
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
      const int np = w % NPp, mt = w / NPp;
      const int n0 = 32 * np, n1 = (n0 + 16 < 16 * NTp) ? n0 + 16 : n0;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      cg_mma_pair<1, 1>(sW + 16 * mt, g.WS, sDY + n0, sDY + n1, g.RS, g.Cout, c0, c1);
      float* out = sBuf + (16 * mt + 4 * slot) * g.RS + l15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        out[q * g.RS + n0] = c0[q];
        if (n1 != n0) out[q * g.RS + n1] = c1[q];
      }
    }
    __syncthreads();

    // P2: dX[ci][grp, j] = sum_o dG[ci][grp, o] A[grp][j][o]      P3: dA[grp][j][o] = sum_ci X[ci][grp, j] dG[ci][grp, o]
    const int n2 = g.GT * MTi * NP, n3 = g.GT * NT * NP;
    float* dxb = dx + (long long)b * g.Cin * TV;
    float* dab = dadj + ((long long)b * g.NG + g0) * g.J * g.J;
    for (int w = wave; w < n2 + n3; w += nw) {
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      if (w < n2) {
        const int np = w % NP, r = w / NP, mt = r % MTi, grp = r / MTi;
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
        const int np = v % NP, r = v / NP, mt = r % NT, grp = r / NT;
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

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_up(int v, int m) { return (v + m - 1) / m * m; }
static int cg_up_4mod8(int v) { int r = cg_up(v, 4); return (r % 8 == 4) ? r : r + 4; }

size_t cg_domm_lds_bytes(const CgDomM& g) {
  return ((size_t)(2 * g.CiM + g.CoM) * g.RS + (size_t)g.GT * g.Jr * g.Jsa + (size_t)g.CoM * g.WS) * sizeof(float);
}

int cg_domm_geom(CgDomM& g, int B, int Cin, int Cout, int T, int V, int domain) {
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
  g.magicJ = g.J > 1 ? (unsigned)((0x100000000ULL + g.J - 1) / g.J) : 0u;
  const size_t limit = 150 * 1024;
  int best = 0;
  for (int gt = g.NG; gt >= 1; --gt) {
    g.GT = gt;
    g.RS = cg_up_4mod8(((gt - 1) * g.Js + g.Jr) > cg_up(gt * g.Js, 16) ? ((gt - 1) * g.Js + g.Jr) : cg_up(gt * g.Js, 16));
    if (cg_domm_lds_bytes(g) > limit) continue;
    if (gt * g.J * g.J >= (1 << 24)) continue;
    const int ntiles = (g.NG + gt - 1) / gt;
    const double waste = (double)ntiles * gt / g.NG;
    const long long total = (long long)B * ntiles, want = (long long)B * g.NG < 512 ? (long long)B * g.NG : 512;
    if (gt > 1 && (waste > 1.15 || total < want)) continue;
    best = gt;
    break;
  }
  if (best == 0) return CG_ESHAPE;
  g.GT = best;
  g.RS = cg_up_4mod8(((best - 1) * g.Js + g.Jr) > cg_up(best * g.Js, 16) ? ((best - 1) * g.Js + g.Jr) : cg_up(best * g.Js, 16));
  g.magicGT = g.GT > 1 ? (unsigned)((0x100000000ULL + g.GT - 1) / g.GT) : 0u;
  g.ntiles = (g.NG + best - 1) / best;
  const long long total = (long long)B * g.ntiles;
  if (total > 2147483647LL) return CG_ESHAPE;
  g.total = (int)total;
  g.per = (int)((total + 255) / 256);            // one workgroup per CU, a contiguous range of tiles each
  return CG_OK;
}

// launches the backward kernel; `ws` holds `replicas` zeroed copies of (dW, db)
int cg_domm_bwd_launch(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj, float* ws,
                       int replicas, int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream) {
  CgDomM g;
  int st = cg_domm_geom(g, B, Cin, Cout, T, V, domain);
  if (st != CG_OK) return st;
  const int tilesW = (g.CoM / 16) * (g.CiM / 16);
  if (tilesW > 8 * 8) return CG_ESHAPE;          // 8 waves x 8 register tiles (Cin, Cout <= 128)
  const size_t lds = cg_domm_lds_bytes(g);
  const long long nwg = ((long long)g.total + g.per - 1) / g.per;
  dim3 grid((unsigned)(((nwg + 7) / 8) * 8)), block(512);
  const int maxw = (tilesW + 7) / 8;
#define CG_DOMM_LAUNCH(D, M)                                                                                              \
  do {                                                                                                                    \
    hipError_t e = hipFuncSetAttribute((const void*)cg_stgcn_domain_bwd_mfma_kernel<D, M>,                                \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
    if (e != hipSuccess) return (int)e;                                                                                   \
    hipLaunchKernelGGL((cg_stgcn_domain_bwd_mfma_kernel<D, M>), grid, block, lds, stream, x, adj, W, dy, dx, dadj, ws,    \
                       replicas, g);                                                                                      \
  } while (0)
  if (domain == 0) {
    if (maxw <= 1) CG_DOMM_LAUNCH(0, 1); else if (maxw <= 2) CG_DOMM_LAUNCH(0, 2); else if (maxw <= 4) CG_DOMM_LAUNCH(0, 4); else CG_DOMM_LAUNCH(0, 8);
  } else {
    if (maxw <= 1) CG_DOMM_LAUNCH(1, 1); else if (maxw <= 2) CG_DOMM_LAUNCH(1, 2); else if (maxw <= 4) CG_DOMM_LAUNCH(1, 4); else CG_DOMM_LAUNCH(1, 8);
  }
#undef CG_DOMM_LAUNCH
  return cg_launch_status();
}
