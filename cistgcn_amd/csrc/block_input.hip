// Head of a DSTD_GC block (reference DSTD_GC.forward, CISTGCN.py:375-379, with _get_stats_ :360-371):
//     xn = global_norm(x)            BatchNorm2d over (B,T,V) per channel
//     stats = _get_stats_(xn)        (B, 2 + 2T): mean / unbiased std over (T,V) and over V, each reduced over the channels
// and its backward.  As separate operators the block input travelled five times forward (BatchNorm apply: read + write, statistics: read)
// and ~16 times backward (statistics backward: read + write, fan-in sum of the eight consumers' gradients: 8 reads + write, BatchNorm
// reduce: 2 reads, BatchNorm apply: 2 reads + write), the statistics kernels at ~1 TB/s (one workgroup per sample, four 4-byte loads in
// flight per lane).  Here:
//   forward   A  streaming: a workgroup takes a contiguous span of a few channel planes, writes xn (16-byte accesses) and, from the LDS
//                image of the span, the mean / centred sum of squares of every row of V joints (rm, rq: (B,C,T), kept for the backward)
//             B  per sample, tiny: the four statistics from rm / rq
//   backward  C  per sample, tiny: the statistics' gradient as a slope and an offset per row:  d xn[c,t,v] += xn P[c,t] + Q[c,t]
//             D  streaming: G = sum_i g_i + xn P + Q  (xn recomputed from x, bit for bit), G stored once, BatchNorm sums in f64
//             E  streaming: dx = gamma rstd (G - mean G - xhat mean(G xhat))
#include <initializer_list>
#include "cg_common.h"
#include "cg_phase.h"
#include "block_input.h"

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

#define CG_BIN_THREADS 256
#define CG_BIN_SPAN 4608                     // floats of a workgroup's span (whole channel planes)

struct CgBinGeom { int TV, NPL, cps, L; unsigned magicTV, magicV; };      // plane size, planes per span, spans per sample, floats of a full span
__device__ __forceinline__ int cg_bin_div(int n, unsigned magic) { return (int)(((unsigned long long)(unsigned)n * magic) >> 32); }

template <int VW>
__device__ __forceinline__ void cg_bin_ld(const float* p, float v[4]) {
  if (VW == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else if (VW == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
  else v[0] = *p;
}
template <int VW>
__device__ __forceinline__ void cg_bin_st(float* p, const float v[4]) {
  if (VW == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else if (VW == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  else *p = v[0];
}

// ======================================================================================================================
// forward A
// ======================================================================================================================
template <int VW>
__global__ __launch_bounds__(CG_BIN_THREADS) void cg_bin_fwd_kernel(CgBlockInput t, CgBinGeom g) {
  float* sXn = reinterpret_cast<float*>(cg_dyn_lds);          // [NPL * TV]
  float* sAff = sXn + g.L;                                    // [NPL][4]: mean, scale, shift, -
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.cps, c0 = (blockIdx.x - b * g.cps) * g.NPL, npl = min(g.NPL, t.C - c0);
  if (tid < npl) {
    const CgAff a = cg_tail_aff(t.bn, c0 + tid, t.C, (double)t.B * (double)g.TV, t.train, false, b == 0);
    sAff[4 * tid] = a.mean; sAff[4 * tid + 1] = a.gamma * a.rstd; sAff[4 * tid + 2] = a.beta;
  }
  __syncthreads();
  const long long base = ((long long)b * t.C + c0) * g.TV;
  const int L = npl * g.TV;
  const float* __restrict__ xp = t.x + base;
  float* __restrict__ yp = t.xn + base;
  for (int e = tid * VW; e < L; e += CG_BIN_THREADS * VW) {
    float v[4];
    cg_bin_ld<VW>(xp + e, v);
    const float* a = sAff + 4 * cg_bin_div(e, g.magicTV);      // TV % VW == 0: a vector never straddles two planes
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      v[j] = (v[j] - a[0]) * a[1] + a[2];                       // mean first, as cg_norm_act
      sXn[e + j] = v[j];
    }
    cg_bin_st<VW>(yp + e, v);
  }
  __syncthreads();
  const long long rbase = ((long long)b * t.C + c0) * t.T;
  for (int r = tid; r < npl * t.T; r += CG_BIN_THREADS) {
    const float* row = sXn + r * t.V;
    float s = 0.f;
    for (int v = 0; v < t.V; ++v) s += row[v];
    const float m = s / (float)t.V;
    float q = 0.f;
    for (int v = 0; v < t.V; ++v) { const float d = row[v] - m; q += d * d; }
    t.rm[rbase + r] = m; t.rq[rbase + r] = q;
  }
}

// rm / rq of one sample into LDS, then per-channel mean and unbiased std over (T,V)
__device__ __forceinline__ void cg_bin_channel_stats(const CgBlockInput& t, int b, float* rm, float* rq, float* cm, float* cs) {
  const int C = t.C, T = t.T, V = t.V;
  for (int r = threadIdx.x; r < C * T; r += blockDim.x) { rm[r] = t.rm[(long long)b * C * T + r]; rq[r] = t.rq[(long long)b * C * T + r]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m = 0.f;
    for (int k = 0; k < T; ++k) m += rm[c * T + k];
    m /= (float)T;
    float q = 0.f;
    for (int k = 0; k < T; ++k) { const float d = rm[c * T + k] - m; q += rq[c * T + k] + (float)V * d * d; }
    cm[c] = m;
    cs[c] = sqrtf(q / (float)(T * V - 1));
  }
  __syncthreads();
}

// forward B: out[b] = [ mean_c mean_{t,v} | mean_c mean_v (T) | std_c std_{t,v} | std_c std_v (T) ], every std unbiased
__global__ __launch_bounds__(CG_BIN_THREADS) void cg_bin_stats_kernel(CgBlockInput t) {
  const int C = t.C, T = t.T, V = t.V;
  float* rm = reinterpret_cast<float*>(cg_dyn_lds);
  float* rq = rm + C * T;
  float* cm = rq + C * T;
  float* cs = cm + C;
  const int b = blockIdx.x;
  cg_bin_channel_stats(t, b, rm, rq, cm, cs);
  float* o = t.out + (long long)b * (2 + 2 * T);
  if (threadIdx.x == 0) {
    float m = 0.f;
    for (int c = 0; c < C; ++c) m += cm[c];
    o[0] = m / (float)C;
    float sm = 0.f;
    for (int c = 0; c < C; ++c) sm += cs[c];
    sm /= (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float d = cs[c] - sm; q += d * d; }
    o[1 + T] = sqrtf(q / (float)(C - 1));
  }
  for (int k = threadIdx.x; k < T; k += blockDim.x) {
    float m = 0.f, sm = 0.f;
    for (int c = 0; c < C; ++c) { m += rm[c * T + k]; sm += sqrtf(rq[c * T + k] / (float)(V - 1)); }
    o[1 + k] = m / (float)C;
    sm /= (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float d = sqrtf(rq[c * T + k] / (float)(V - 1)) - sm; q += d * d; }
    o[2 + T + k] = sqrtf(q / (float)(C - 1));
  }
}

// ======================================================================================================================
// backward C: d xn[c,t,v] (from the statistics) = xn * P[c,t] + Q[c,t]
// ======================================================================================================================
__global__ __launch_bounds__(CG_BIN_THREADS) void cg_bin_coef_kernel(CgBlockInput t) {
  const int C = t.C, T = t.T, V = t.V;
  float* rm = reinterpret_cast<float*>(cg_dyn_lds);
  float* rq = rm + C * T;
  float* cm = rq + C * T;
  float* cs = cm + C;
  float* tS = cs + C;       // [T] std over c of the row stds
  float* tM = tS + T;       // [T] mean over c of the row stds
  float* sg = tM + T;       // [2 + 2T] summed gradient of the statistics
  __shared__ float gS, gSm;
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < 2 + 2 * T; i += blockDim.x) {
    float v = 0.f;
    if (t.dout[0]) v += t.dout[0][(long long)b * t.dout_ld[0] + i];
    if (t.dout[1]) v += t.dout[1][(long long)b * t.dout_ld[1] + i];
    sg[i] = v;
  }
  cg_bin_channel_stats(t, b, rm, rq, cm, cs);
  if (threadIdx.x == 0) {
    float sm = 0.f;
    for (int c = 0; c < C; ++c) sm += cs[c];
    sm /= (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float d = cs[c] - sm; q += d * d; }
    gS = sqrtf(q / (float)(C - 1));
    gSm = sm;
  }
  for (int k = threadIdx.x; k < T; k += blockDim.x) {
    float sm = 0.f;
    for (int c = 0; c < C; ++c) sm += sqrtf(rq[c * T + k] / (float)(V - 1));
    sm /= (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float d = sqrtf(rq[c * T + k] / (float)(V - 1)) - sm; q += d * d; }
    tS[k] = sqrtf(q / (float)(C - 1));
    tM[k] = sm;
  }
  __syncthreads();
  const float g0 = sg[0] / (float)(C * T * V);
  const float gall = sg[1 + T];
  float* pq = t.pq + (long long)b * C * T * 2;
  for (int r = threadIdx.x; r < C * T; r += blockDim.x) {
    const int c = r / T, k = r - c * T;
    // std over c of s_c, s_c = std over (t,v);  std over c of s_ct, s_ct = std over v
    const float pc = gall * (cs[c] - gSm) / ((float)(C - 1) * gS) / ((float)(T * V - 1) * cs[c]);
    const float sct = sqrtf(rq[r] / (float)(V - 1));
    const float pr = sg[2 + T + k] * (sct - tM[k]) / ((float)(C - 1) * tS[k]) / ((float)(V - 1) * sct);
    pq[2 * r] = pc + pr;
    pq[2 * r + 1] = g0 + sg[1 + k] / (float)(C * V) - pc * cm[c] - pr * rm[r];
  }
}

// ======================================================================================================================
// backward D: G = sum_i g_i + xn P + Q over a span of channel planes; f64 sums of G and G xhat per channel
// ======================================================================================================================
template <int VW>
__global__ __launch_bounds__(CG_BIN_THREADS) void cg_bin_bwd_sum_kernel(CgBlockInput t, CgBinGeom g) {
  float* sPQ = reinterpret_cast<float*>(cg_dyn_lds);           // [NPL * T][2]
  float* sAff = sPQ + 2 * g.NPL * t.T;                        // [NPL][4]: mean, rstd, scale, shift
  double* sRed = reinterpret_cast<double*>(sAff + 4 * g.NPL);      // [NPL][2]; an even number of floats in front of it
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.cps, c0 = (blockIdx.x - b * g.cps) * g.NPL, npl = min(g.NPL, t.C - c0);
  if (tid < npl) {
    const CgAff a = cg_tail_aff(t.bn, c0 + tid, t.C, 0.0, 0, true, false);
    sAff[4 * tid] = a.mean; sAff[4 * tid + 1] = a.rstd; sAff[4 * tid + 2] = a.gamma * a.rstd; sAff[4 * tid + 3] = a.beta;
    sRed[2 * tid] = 0.0; sRed[2 * tid + 1] = 0.0;
  }
  for (int i = tid; i < 2 * npl * t.T; i += CG_BIN_THREADS) sPQ[i] = t.pq ? t.pq[((long long)b * t.C + c0) * t.T * 2 + i] : 0.f;
  __syncthreads();
  const long long base = ((long long)b * t.C + c0) * g.TV;
  const int L = npl * g.TV;
  const float* __restrict__ xp = t.x + base;
  float* __restrict__ gp = t.gsum + base;
  int cur = -1;
  double s1 = 0.0, s2 = 0.0;
  const float* gsrc[CG_BIN_MAXG];                               // the consumers' gradients at this workgroup's span (null: none), read once
#pragma unroll
  for (int i = 0; i < CG_BIN_MAXG; ++i) gsrc[i] = (i < t.ng && t.g[i]) ? t.g[i] + base : nullptr;
  for (int e = tid * VW; e < L; e += CG_BIN_THREADS * VW) {
    float x[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
    cg_bin_ld<VW>(xp + e, x);
    // every gradient's vector is REQUESTED before the first one is added: with the add right behind its load the kernel made up to eight
    // memory round trips one after the other per vector (load - wait - add, found in the ISA in round 4)
    float gv[CG_BIN_MAXG][4];
#pragma unroll
    for (int i = 0; i < CG_BIN_MAXG; ++i)
      if (gsrc[i]) cg_bin_ld<VW>(gsrc[i] + e, gv[i]);
#pragma unroll
    for (int i = 0; i < CG_BIN_MAXG; ++i) {
      if (gsrc[i]) {
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += gv[i][j];
      }
    }
    const int pl = cg_bin_div(e, g.magicTV);                    // TV % VW == 0: a vector never straddles two planes
    const float* a = sAff + 4 * pl;
    if (pl != cur) {
      if (cur >= 0) { atomicAdd(&sRed[2 * cur], s1); atomicAdd(&sRed[2 * cur + 1], s2); }
      cur = pl; s1 = 0.0; s2 = 0.0;
    }
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int r = cg_bin_div(e + j, g.magicV);
      const float d = x[j] - a[0];
      const float xn = d * a[2] + a[3];                          // the forward's expression: the same xn bit for bit
      const float G = acc[j] + xn * sPQ[2 * r] + sPQ[2 * r + 1];
      acc[j] = G;
      s1 += (double)G; s2 += (double)G * (double)(d * a[1]);
    }
    cg_bin_st<VW>(gp + e, acc);
  }
  if (cur >= 0) { atomicAdd(&sRed[2 * cur], s1); atomicAdd(&sRed[2 * cur + 1], s2); }
  __syncthreads();
  if (tid < 2 * npl) {
    double* red = t.red + ((long long)(blockIdx.x % CG_STAT_REPLICAS) * t.C + c0) * 2;
    atomicAdd(&red[tid], sRed[tid]);
  }
}

// backward E: dx = gamma rstd (G - m1 - xhat m2) (train) | gamma rstd G (eval); the first span of a channel writes dgamma / dbeta
template <int VW>
__global__ __launch_bounds__(CG_BIN_THREADS) void cg_bin_bwd_apply_kernel(CgBlockInput t, CgBinGeom g) {
  float* sAff = reinterpret_cast<float*>(cg_dyn_lds);          // [NPL][6]: mean, rstd, scale, m1, m2, -
  const int tid = threadIdx.x;
  const int b = blockIdx.x / g.cps, c0 = (blockIdx.x - b * g.cps) * g.NPL, npl = min(g.NPL, t.C - c0);
  if (tid < npl) {
    const int c = c0 + tid;
    const CgAff a = cg_tail_aff(t.bn, c, t.C, 0.0, 0, true, false);
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < CG_STAT_REPLICAS; ++r) { s1 += t.red[((long long)r * t.C + c) * 2]; s2 += t.red[((long long)r * t.C + c) * 2 + 1]; }
    const double cnt = (double)t.B * (double)g.TV;
    float* k = sAff + 6 * tid;
    k[0] = a.mean; k[1] = a.rstd; k[2] = a.gamma * a.rstd;
    k[3] = t.train ? (float)(s1 / cnt) : 0.f; k[4] = t.train ? (float)(s2 / cnt) : 0.f;
    if (b == 0) { t.dgamma[c] = (float)s2; t.dbeta[c] = (float)s1; }
  }
  __syncthreads();
  const long long base = ((long long)b * t.C + c0) * g.TV;
  const int L = npl * g.TV;
  const float* __restrict__ xp = t.x + base;
  const float* __restrict__ gp = t.gsum + base;
  float* __restrict__ dp = t.dx + base;
  for (int e = tid * VW; e < L; e += CG_BIN_THREADS * VW) {
    float x[4], G[4];
    cg_bin_ld<VW>(xp + e, x);
    cg_bin_ld<VW>(gp + e, G);
    const float* k = sAff + 6 * cg_bin_div(e, g.magicTV);
#pragma unroll
    for (int j = 0; j < VW; ++j) G[j] = k[2] * (G[j] - k[3] - (x[j] - k[0]) * k[1] * k[4]);
    cg_bin_st<VW>(dp + e, G);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static unsigned cg_bin_magic(int d) { return (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

static int cg_bin_geometry(const CgBlockInput* t, CgBinGeom* g) {
  if (!t) return CG_EARG;
  if (t->B <= 0 || t->C < 2 || t->T <= 0 || t->V < 2) return CG_ESHAPE;
  const long long TV = (long long)t->T * t->V;
  if (TV > CG_BIN_SPAN || (size_t)(2 * t->C * t->T + 2 * t->C + 4 * t->T + 4) * sizeof(float) > 150 * 1024) return CG_ESHAPE;
  g->TV = (int)TV;
  g->NPL = CG_BIN_SPAN / (int)TV;
  if (g->NPL > t->C) g->NPL = t->C;
  g->cps = (t->C + g->NPL - 1) / g->NPL;
  g->L = g->NPL * g->TV;
  g->magicTV = cg_bin_magic(g->TV); g->magicV = cg_bin_magic(t->V);
  if ((long long)t->B * g->cps > 2147483647LL) return CG_ESHAPE;
  return CG_OK;
}
// vector width of the streaming kernels: every pointer they touch and the plane size
static int cg_bin_vw(const CgBinGeom& g, std::initializer_list<const void*> ptrs) {
  int vw = (g.TV & 3) == 0 ? 4 : (g.TV & 1) == 0 ? 2 : 1;
  for (const void* p : ptrs) {
    if (!p) continue;
    const uintptr_t a = (uintptr_t)p;
    while (vw > 1 && (a & (uintptr_t)(4 * vw - 1))) vw >>= 1;
  }
  return vw;
}

extern "C" int cg_block_input_supported(int B, int C, int T, int V) {
  CgBlockInput t;
  t.B = B; t.C = C; t.T = T; t.V = V;
  CgBinGeom g;
  return cg_bin_geometry(&t, &g) == CG_OK ? 1 : 0;
}

// include/cistgcn_hip.h : cg_block_input_fwd / cg_block_input_bwd
extern "C" int cg_block_input_fwd(const CgBlockInput* t, void* stream_) {
  CgBinGeom g;
  int st = cg_bin_geometry(t, &g);
  if (st != CG_OK) return st;
  if (!t->x || !t->xn || !t->rm || !t->rq || !t->out || !t->bn.gamma || !t->bn.beta || !t->bn.save) return CG_EARG;
  if (t->train ? !t->bn.stats : (!t->bn.running_mean || !t->bn.running_var)) return CG_EARG;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t lds = ((size_t)g.L + 4 * g.NPL) * sizeof(float);
  const dim3 grid((unsigned)(t->B * g.cps)), block(CG_BIN_THREADS);
  const int vw = cg_bin_vw(g, {t->x, t->xn});
  if (vw == 4) hipLaunchKernelGGL(cg_bin_fwd_kernel<4>, grid, block, lds, stream, *t, g);
  else if (vw == 2) hipLaunchKernelGGL(cg_bin_fwd_kernel<2>, grid, block, lds, stream, *t, g);
  else hipLaunchKernelGGL(cg_bin_fwd_kernel<1>, grid, block, lds, stream, *t, g);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  const size_t lds2 = (size_t)(2 * t->C * t->T + 2 * t->C) * sizeof(float);
  if (lds2 > 64 * 1024 && cg_lds_limit((const void*)cg_bin_stats_kernel, lds2) != hipSuccess) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_bin_stats_kernel, dim3((unsigned)t->B), block, lds2, stream, *t);
  return cg_launch_status();
}

extern "C" int cg_block_input_bwd(const CgBlockInput* t, void* stream_) {
  CgBinGeom g;
  int st = cg_bin_geometry(t, &g);
  if (st != CG_OK) return st;
  if (!t->x || !t->rm || !t->rq || !t->bn.gamma || !t->bn.beta || !t->bn.save || !t->gsum || !t->red || !t->dx || !t->dgamma || !t->dbeta) return CG_EARG;
  if (t->ng < 0 || t->ng > CG_BIN_MAXG) return CG_EARG;
  const bool stats_grad = t->dout[0] || t->dout[1];
  if (stats_grad && !t->pq) return CG_EARG;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 block(CG_BIN_THREADS);
  CgBlockInput a = *t;
  if (stats_grad) {
    const size_t lds = (size_t)(2 * t->C * t->T + 2 * t->C + 4 * t->T + 2) * sizeof(float);
    if (lds > 64 * 1024 && cg_lds_limit((const void*)cg_bin_coef_kernel, lds) != hipSuccess) return CG_ESHAPE;
    hipLaunchKernelGGL(cg_bin_coef_kernel, dim3((unsigned)t->B), block, lds, stream, a);
    st = cg_launch_status();
    if (st != CG_OK) return st;
  } else a.pq = nullptr;
  const dim3 grid((unsigned)(t->B * g.cps));
  const size_t lds = ((size_t)2 * g.NPL * t->T + 4 * g.NPL + 2) * sizeof(float) + (size_t)2 * g.NPL * sizeof(double);
  int vw = cg_bin_vw(g, {t->x, t->gsum, t->g[0], t->g[1], t->g[2], t->g[3], t->g[4], t->g[5], t->g[6], t->g[7]});
  if (vw == 4) hipLaunchKernelGGL(cg_bin_bwd_sum_kernel<4>, grid, block, lds, stream, a, g);
  else if (vw == 2) hipLaunchKernelGGL(cg_bin_bwd_sum_kernel<2>, grid, block, lds, stream, a, g);
  else hipLaunchKernelGGL(cg_bin_bwd_sum_kernel<1>, grid, block, lds, stream, a, g);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  const size_t lds3 = (size_t)6 * g.NPL * sizeof(float);
  vw = cg_bin_vw(g, {t->x, t->gsum, t->dx});
  if (vw == 4) hipLaunchKernelGGL(cg_bin_bwd_apply_kernel<4>, grid, block, lds3, stream, a, g);
  else if (vw == 2) hipLaunchKernelGGL(cg_bin_bwd_apply_kernel<2>, grid, block, lds3, stream, a, g);
  else hipLaunchKernelGGL(cg_bin_bwd_apply_kernel<1>, grid, block, lds3, stream, a, g);
  return cg_launch_status();
}
