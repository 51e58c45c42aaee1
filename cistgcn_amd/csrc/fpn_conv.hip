// The dilated 3x3 convolutions of the time extrapolator (reference: FPN.forward CISTGCN.py:74-79, blocks of :54-68: three
// Conv2d(k=3, padding = dilation = 1, 2, 3) of the same (B, frames, channels, joints) tensor).  A sample of that tensor is 22-44 KB:
// it fits in LDS with its zero halo, so each convolution and each of its gradients is one pass over whole samples instead of a
// generic strided contraction with 167-way split-K (5.6 MB tensors used to cost 0.2-0.8 ms per launch):
//   forward   (sample, dilation) workgroups: halo image of x[b] + W_d in LDS, y = W_d (*) x as MFMA tiles [positions][outputs]
//             with K = (c, i, j) taps gathered from the image through an offset table
//   backward  dx: one workgroup per sample walks the dilations, halo image of dy_d + transposed W_d, accumulators over all three
//             dW / db: persistent workgroups per dilation keep the [outputs][(c,i,j)] tiles in registers over their samples
#include "cg_common.h"
#include "cg_phase.h"
#include "fpn_conv.h"

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

#define CG_FPN_THREADS 256
#define CG_FPN_PAD 3
#define CG_FPN_REPLICAS 8

struct CgFpnGeom {
  int HP, WP, IMG;              // halo image of one channel: (H + 6) x (W + 6), floats per channel
  int P, PM;                    // positions H * W, rounded up to 16
  int K, KP, KS;                // taps C * 9 (forward / dW) rounded up to 16, weight row stride
  int K2, KP2, KS2;             // taps O * 9 (dx)
  int OM, CM;                   // outputs / inputs rounded up to 16
  int per;                      // dW: samples per workgroup
};
struct CgFpnArgs { CgFpnConv t; CgFpnGeom g; };

// global -> LDS staging with eight loads of a thread in flight (a load-store loop would wait out one memory latency per element)
template <typename L, typename S>
__device__ __forceinline__ void cg_fpn_stage(int n, L load, S store) {
  for (int e0 = threadIdx.x; e0 < n; e0 += 8 * CG_FPN_THREADS) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int e = e0 + CG_FPN_THREADS * j; v[j] = e < n ? load(e) : 0.f; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int e = e0 + CG_FPN_THREADS * j; if (e < n) store(e, v[j]); }
  }
}

// zero-halo image of `n` channels of one sample: img[c][HP][WP]; the halo is zeroed once per workgroup (cg_fpn_zero, floats % 4 == 0),
// the interior rewritten per sample
__device__ __forceinline__ void cg_fpn_zero(float* img, int floats) {
  for (int e = threadIdx.x; e < floats / 4; e += CG_FPN_THREADS) reinterpret_cast<float4*>(img)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void cg_fpn_fill(const float* __restrict__ src, long long sc, long long sh, int n, int H, int W, const CgFpnGeom& g, float* img) {
  const int P = H * W;
  cg_fpn_stage(n * P,
               [&](int e) { const int c = e / P, p = e - c * P, h = p / W; return src[c * sc + h * sh + (p - h * W)]; },
               [&](int e, float v) { const int c = e / P, p = e - c * P, h = p / W; img[c * g.IMG + (h + CG_FPN_PAD) * g.WP + (p - h * W) + CG_FPN_PAD] = v; });
}
// offset of position p (row-major H x W) inside a channel's halo image
__device__ __forceinline__ int cg_fpn_posoff(int p, int W, const CgFpnGeom& g) { const int h = p / W; return (h + CG_FPN_PAD) * g.WP + (p - h * W) + CG_FPN_PAD; }

// ======================================================================================================================
// forward: y[o][p] = bias[o] + sum_k W[o][k] img[off(p) + tap(k)],  k = (c, i, j)
// ======================================================================================================================
__global__ __launch_bounds__(CG_FPN_THREADS) void cg_fpn_fwd_kernel(CgFpnArgs a) {
  const CgFpnConv& t = a.t; const CgFpnGeom& g = a.g;
  const int b = blockIdx.x, di = blockIdx.y, d = t.dil[di];
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);              // [C][IMG]
  float* sW = sX + t.C * g.IMG;                                   // [OM][KS]
  int* sTap = reinterpret_cast<int*>(sW + g.OM * g.KS);           // [KP] image offset of tap k
  int* sPos = sTap + g.KP;                                        // [PM] image offset of position p
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, slot = lane >> 4;
  cg_fpn_zero(sX, t.C * g.IMG);
  cg_fpn_zero(sW, g.OM * g.KS);
  __syncthreads();
  cg_fpn_fill(t.x + (long long)b * t.xs[0], t.xs[1], t.xs[2], t.C, t.H, t.W, g, sX);
  cg_fpn_stage(t.O * g.K, [&](int e) { return t.w[di][e]; }, [&](int e, float v) { const int o = e / g.K; sW[o * g.KS + e - o * g.K] = v; });
  for (int k = tid; k < g.KP; k += CG_FPN_THREADS) {
    const int c = k / 9, ij = k - 9 * c, i = ij / 3, j = ij - 3 * i;
    sTap[k] = k < g.K ? c * g.IMG + d * (i - 1) * g.WP + d * (j - 1) : 0;
  }
  for (int p = tid; p < g.PM; p += CG_FPN_THREADS) sPos[p] = cg_fpn_posoff(p < g.P ? p : 0, t.W, g);
  __syncthreads();
  const int PT = g.PM / 16, OT = g.OM / 16, pairs = (PT + 1) / 2;
  float* yb = t.y[di] + (long long)b * t.O * g.P;
  for (int w = wave; w < pairs * OT; w += CG_FPN_THREADS / 64) {
    const int pr = w / OT, ot = w - pr * OT, pt0 = 2 * pr, pt1 = min(PT - 1, pt0 + 1);
    const int pa = sPos[16 * pt0 + l15], pb = sPos[16 * pt1 + l15];
    cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    const float* wp = sW + (16 * ot + l15) * g.KS + 4 * slot;
    for (int k0 = 0; k0 < g.KP; k0 += 16) {
      const float4 w4 = *reinterpret_cast<const float4*>(wp + k0);
      const int4 t4 = *reinterpret_cast<const int4*>(sTap + k0 + 4 * slot);
      const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
      const int tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {                       // C[position][output]
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sX[pa + tv[s]], wv[s], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sX[pb + tv[s]], wv[s], c1, 0, 0, 0);
      }
    }
    const int o = 16 * ot + l15;
    if (o < t.O) {
      const float bias = t.bias[di] ? t.bias[di][o] : 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && pt1 == pt0) break;
        const int p = 16 * (h ? pt1 : pt0) + 4 * slot;
        const cg_f32x4 c = h ? c1 : c0;
        if (p < g.P) *reinterpret_cast<float4*>(yb + (long long)o * g.P + p) = make_float4(c[0] + bias, c[1] + bias, c[2] + bias, c[3] + bias);
      }
    }
  }
}

// ======================================================================================================================
// backward, input gradient: dx[c][p] = sum_d sum_k' Wt_d[c][k'] dyimg_d[off(p) - tap_d(k')],  k' = (o, i, j)
// ======================================================================================================================
#define CG_FPN_DXT 14        // accumulator tiles per wave: ceil(16 position tiles / 2 pairs...) see host check

__global__ __launch_bounds__(CG_FPN_THREADS) void cg_fpn_dx_kernel(CgFpnArgs a) {
  const CgFpnConv& t = a.t; const CgFpnGeom& g = a.g;
  const int b = blockIdx.x;
  float* sD = reinterpret_cast<float*>(cg_dyn_lds);              // [O][IMG]
  float* sW = sD + t.O * g.IMG;                                   // [CM][KS2]  Wt[c][(o,i,j)]
  int* sTap = reinterpret_cast<int*>(sW + g.CM * g.KS2);          // [KP2]
  int* sPos = sTap + g.KP2;                                       // [PM]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, slot = lane >> 4, nw = CG_FPN_THREADS / 64;
  const int PT = g.PM / 16, CT = g.CM / 16;
  cg_f32x4 acc[CG_FPN_DXT];
#pragma unroll
  for (int u = 0; u < CG_FPN_DXT; ++u) acc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int p = tid; p < g.PM; p += CG_FPN_THREADS) sPos[p] = cg_fpn_posoff(p < g.P ? p : 0, t.W, g);
  cg_fpn_zero(sD, t.O * g.IMG);
  cg_fpn_zero(sW, g.CM * g.KS2);
  for (int di = 0; di < t.n; ++di) {
    const int d = t.dil[di];
    __syncthreads();
    cg_fpn_fill(t.dy[di] + (long long)b * t.O * g.P, g.P, t.W, t.O, t.H, t.W, g, sD);
    // W_d[o][c][ij] (read in its own order) -> Wt[c][o * 9 + ij]; the padding of sW was zeroed once
    cg_fpn_stage(t.O * t.C * 9, [&](int e) { return t.w[di][e]; },
                 [&](int e, float v) { const int oc = e / 9, ij = e - 9 * oc, o = oc / t.C, c = oc - o * t.C; sW[c * g.KS2 + o * 9 + ij] = v; });
    for (int k = tid; k < g.KP2; k += CG_FPN_THREADS) {
      const int o = k / 9, ij = k - 9 * o, i = ij / 3, j = ij - 3 * i;
      sTap[k] = k < g.K2 ? o * g.IMG - d * (i - 1) * g.WP - d * (j - 1) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CG_FPN_DXT; ++u) {
      const int id = u * nw + wave;
      if (id < PT * CT) {
        const int pt = id / CT, ct = id - pt * CT;
        const int pa = sPos[16 * pt + l15];
        const float* wp = sW + (16 * ct + l15) * g.KS2 + 4 * slot;
        cg_f32x4 c0 = acc[u];
        for (int k0 = 0; k0 < g.KP2; k0 += 16) {
          const float4 w4 = *reinterpret_cast<const float4*>(wp + k0);
          const int4 t4 = *reinterpret_cast<const int4*>(sTap + k0 + 4 * slot);
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sD[pa + t4.x], w4.x, c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sD[pa + t4.y], w4.y, c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sD[pa + t4.z], w4.z, c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(sD[pa + t4.w], w4.w, c0, 0, 0, 0);
        }
        acc[u] = c0;
      }
    }
  }
  float* dxb = t.dx + (long long)b * t.C * g.P;
#pragma unroll
  for (int u = 0; u < CG_FPN_DXT; ++u) {
    const int id = u * nw + wave;
    if (id < PT * CT) {
      const int pt = id / CT, ct = id - pt * CT, c = 16 * ct + l15, p = 16 * pt + 4 * slot;
      if (c < t.C && p < g.P) *reinterpret_cast<float4*>(dxb + (long long)c * g.P + p) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
    }
  }
}

// ======================================================================================================================
// backward, weight and bias gradients: dW_d[o][k] += sum_{b,p} dy_d[b][o][p] img_b[off(p) + tap_d(k)],  db_d[o] += sum dy_d
// ======================================================================================================================
#define CG_FPN_DWT 16        // [outputs][taps] register tiles per wave: 2 * ceil16(64 * 9) / 16 / 4 waves = 18 at C = 64; host checks

__global__ __launch_bounds__(CG_FPN_THREADS) void cg_fpn_dw_kernel(CgFpnArgs a) {
  const CgFpnConv& t = a.t; const CgFpnGeom& g = a.g;
  const int di = blockIdx.y, d = t.dil[di];
  const int b0 = blockIdx.x * g.per, b1 = min(t.B, b0 + g.per);
  if (b0 >= t.B) return;
  const int DS = g.PM + 4;
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);              // [C][IMG]
  float* sD = sX + t.C * g.IMG;                                   // [OM][DS]  dy rows, positions padded with zeros
  int* sTap = reinterpret_cast<int*>(sD + g.OM * DS);             // [KP]
  int* sPos = sTap + g.KP;                                        // [PM]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, slot = lane >> 4, nw = CG_FPN_THREADS / 64;
  const int OT = g.OM / 16, KT = g.KP / 16;
  cg_f32x4 acc[CG_FPN_DWT];
#pragma unroll
  for (int u = 0; u < CG_FPN_DWT; ++u) acc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int k = tid; k < g.KP; k += CG_FPN_THREADS) {
    const int c = k / 9, ij = k - 9 * c, i = ij / 3, j = ij - 3 * i;
    sTap[k] = k < g.K ? c * g.IMG + d * (i - 1) * g.WP + d * (j - 1) : 0;
  }
  for (int p = tid; p < g.PM; p += CG_FPN_THREADS) sPos[p] = cg_fpn_posoff(p < g.P ? p : 0, t.W, g);
  for (int e = tid; e < g.OM * DS; e += CG_FPN_THREADS) sD[e] = 0.f;
  cg_fpn_zero(sX, t.C * g.IMG);
  for (int b = b0; b < b1; ++b) {
    __syncthreads();
    cg_fpn_fill(t.x + (long long)b * t.xs[0], t.xs[1], t.xs[2], t.C, t.H, t.W, g, sX);
    const float* dyb = t.dy[di] + (long long)b * t.O * g.P;
#pragma unroll 4
    for (int e = tid; e < t.O * (g.P / 4); e += CG_FPN_THREADS) {
      const int o = e / (g.P / 4), p = 4 * (e - o * (g.P / 4));
      *reinterpret_cast<float4*>(sD + o * DS + p) = *reinterpret_cast<const float4*>(dyb + (long long)o * g.P + p);
    }
    __syncthreads();
    if (tid < t.O) {
      float s = 0.f;
      for (int p = 0; p < g.P; ++p) s += sD[tid * DS + p];
      bsum += s;
    }
#pragma unroll
    for (int u = 0; u < CG_FPN_DWT; ++u) {
      const int id = u * nw + wave;
      if (id < OT * KT) {
        const int ot = id / KT, kt = id - ot * KT;
        const int tap = sTap[16 * kt + l15];
        const float* dp = sD + (16 * ot + l15) * DS + 4 * slot;
        cg_f32x4 c0 = acc[u];
        for (int p0 = 0; p0 < g.PM; p0 += 16) {
          const float4 d4 = *reinterpret_cast<const float4*>(dp + p0);
          const int4 q4 = *reinterpret_cast<const int4*>(sPos + p0 + 4 * slot);
          // positions beyond P carry dy = 0 (their image offset is position 0's: any finite value)
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d4.x, sX[tap + q4.x], c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d4.y, sX[tap + q4.y], c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d4.z, sX[tap + q4.z], c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d4.w, sX[tap + q4.w], c0, 0, 0, 0);
        }
        acc[u] = c0;
      }
    }
  }
  // [replica][dilation][O * K + O] accumulators
  float* ws = t.ws + ((long long)(blockIdx.x % CG_FPN_REPLICAS) * t.n + di) * (t.O * g.K + t.O);
#pragma unroll
  for (int u = 0; u < CG_FPN_DWT; ++u) {
    const int id = u * nw + wave;
    if (id < OT * KT) {
      const int ot = id / KT, kt = id - ot * KT, k = 16 * kt + l15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = 16 * ot + 4 * slot + q;
        if (o < t.O && k < g.K) atomicAdd(&ws[o * g.K + k], acc[u][q]);
      }
    }
  }
  if (tid < t.O) atomicAdd(&ws[t.O * g.K + tid], bsum);
}

__global__ void cg_fpn_fold_kernel(CgFpnArgs a) {
  const CgFpnConv& t = a.t; const CgFpnGeom& g = a.g;
  const int di = blockIdx.y, n = t.O * g.K + t.O;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < CG_FPN_REPLICAS; ++r) s += t.ws[((long long)r * t.n + di) * n + e];
    if (e < t.O * g.K) { if (t.dw[di]) t.dw[di][e] = s; }
    else if (t.db[di]) t.db[di][e - t.O * g.K] = s;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_fpn_geometry(const CgFpnConv* t, CgFpnGeom* g) {
  if (!t || t->n <= 0 || t->n > 3) return CG_EARG;
  if (t->B <= 0 || t->C <= 0 || t->C > 64 || t->O <= 0 || t->O > 32 || t->H <= 0 || t->W <= 0) return CG_ESHAPE;
  for (int i = 0; i < t->n; ++i) if (t->dil[i] < 1 || t->dil[i] > CG_FPN_PAD) return CG_ESHAPE;
  g->HP = t->H + 2 * CG_FPN_PAD; g->WP = t->W + 2 * CG_FPN_PAD; g->IMG = g->HP * g->WP;
  g->P = t->H * t->W; g->PM = (g->P + 15) & ~15;
  if ((g->P & 3) || g->PM > 256) return CG_ESHAPE;
  g->K = t->C * 9; g->KP = (g->K + 15) & ~15; g->KS = g->KP + 4;
  g->K2 = t->O * 9; g->KP2 = (g->K2 + 15) & ~15; g->KS2 = g->KP2 + 4;
  g->OM = (t->O + 15) & ~15; g->CM = (t->C + 15) & ~15;
  const int nwg = t->B < 170 ? t->B : 170;           // dW: ~512 workgroups over the three dilations
  g->per = (t->B + nwg - 1) / nwg;
  if ((g->PM / 16) * (g->CM / 16) > CG_FPN_DXT * 4) return CG_ESHAPE;
  if ((g->OM / 16) * (g->KP / 16) > CG_FPN_DWT * 4) return CG_ESHAPE;
  return CG_OK;
}
static size_t cg_fpn_lds_fwd(const CgFpnConv* t, const CgFpnGeom& g) { return ((size_t)t->C * g.IMG + (size_t)g.OM * g.KS + g.KP + g.PM) * 4; }
static size_t cg_fpn_lds_dx(const CgFpnConv* t, const CgFpnGeom& g) { return ((size_t)t->O * g.IMG + (size_t)g.CM * g.KS2 + g.KP2 + g.PM) * 4; }
static size_t cg_fpn_lds_dw(const CgFpnConv* t, const CgFpnGeom& g) { return ((size_t)t->C * g.IMG + (size_t)g.OM * (g.PM + 4) + g.KP + g.PM) * 4; }

// 1 when cg_fpn_conv_* takes the shape (else the caller uses the generic contraction)
extern "C" int cg_fpn_conv_supported(int B, int C, int O, int H, int W) {
  CgFpnConv t = {};
  t.B = B; t.C = C; t.O = O; t.H = H; t.W = W; t.n = 1; t.dil[0] = 1;
  CgFpnGeom g;
  if (cg_fpn_geometry(&t, &g) != CG_OK) return 0;
  const size_t cap = 160 * 1024;
  return cg_fpn_lds_fwd(&t, g) <= cap && cg_fpn_lds_dx(&t, g) <= cap && cg_fpn_lds_dw(&t, g) <= cap;
}
extern "C" long long cg_fpn_conv_ws_floats(int C, int O, int n) { return (long long)CG_FPN_REPLICAS * n * (O * C * 9 + O); }

// include/cistgcn_hip.h : cg_fpn_conv_fwd / cg_fpn_conv_bwd
extern "C" int cg_fpn_conv_fwd(const CgFpnConv* t, void* stream_) {
  CgFpnArgs a;
  int st = cg_fpn_geometry(t, &a.g);
  if (st != CG_OK) return st;
  a.t = *t;
  if (!t->x) return CG_EARG;
  for (int i = 0; i < t->n; ++i) if (!t->w[i] || !t->y[i]) return CG_EARG;
  const size_t lds = cg_fpn_lds_fwd(t, a.g);
  if (lds > 160 * 1024) return CG_ESHAPE;
  hipError_t e = hipFuncSetAttribute((const void*)cg_fpn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(cg_fpn_fwd_kernel, dim3((unsigned)t->B, (unsigned)t->n), dim3(CG_FPN_THREADS), lds, (hipStream_t)stream_, a);
  return cg_launch_status();
}

extern "C" int cg_fpn_conv_bwd(const CgFpnConv* t, void* stream_) {
  CgFpnArgs a;
  int st = cg_fpn_geometry(t, &a.g);
  if (st != CG_OK) return st;
  a.t = *t;
  if (!t->x || !t->ws) return CG_EARG;
  for (int i = 0; i < t->n; ++i) if (!t->w[i] || !t->dy[i]) return CG_EARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (t->dx) {
    const size_t lds = cg_fpn_lds_dx(t, a.g);
    if (lds > 160 * 1024) return CG_ESHAPE;
    hipError_t e = hipFuncSetAttribute((const void*)cg_fpn_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_fpn_dx_kernel, dim3((unsigned)t->B), dim3(CG_FPN_THREADS), lds, stream, a);
    st = cg_launch_status();
    if (st != CG_OK) return st;
  }
  const size_t lds = cg_fpn_lds_dw(t, a.g);
  if (lds > 160 * 1024) return CG_ESHAPE;
  hipError_t e = hipFuncSetAttribute((const void*)cg_fpn_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  const int nwg = (t->B + a.g.per - 1) / a.g.per;
  hipLaunchKernelGGL(cg_fpn_dw_kernel, dim3((unsigned)nwg, (unsigned)t->n), dim3(CG_FPN_THREADS), lds, stream, a);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  hipLaunchKernelGGL(cg_fpn_fold_kernel, dim3(16, (unsigned)t->n), dim3(256), 0, stream, a);
  return cg_launch_status();
}
