// ContextLayer heads 1 and 3 without their (B, hidden, T_out, 3V) activations (reference: ContextLayer.context_conv1 / context_conv3,
// CISTGCN.py:408-418, consumed by `.max(-1)[0].max(-1)[0]` :465 and `.mean((2, 3))` :467).
//
// Both heads are  Conv2d(1, C, 1, bias=False) -> BatchNorm2d(C) -> PReLU  of the ONE-channel tensor x (B,1,T_out,3V), reduced over all
// positions right away.  With one input channel the convolution output is w[c] * x[b,p]: its batch statistics are w[c] * mean(x) and
// w[c]^2 * var(x), and every element of the (B,C,P) activation is a function of x[b,p] and five per-channel constants.  As composite
// operators the two heads wrote and re-read ~2 GB per training step at B = 256 (two 108 MB tensors: convolution, BatchNorm + PReLU,
// reduction, and all of it again backwards); here a sample's 1650 positions sit in LDS and the activations exist in registers only.
//
//   forward   (one workgroup per sample)  y0[b,c] = max_p z0, first arg-max | y1[b,c] = mean_p z1,   z_h = PReLU(BN(w_h[c] x[b,p]))
//   backward  phase 1 (per sample): per-channel sums of the gradient in front of the BatchNorm: S1 = sum gu, G = sum gu * x, and the
//             slope gradient; phase 2 (per sample): dx[b,p]; workgroup 0 also writes dw, dgamma, dbeta, dalpha in closed form:
//             with xhat = (w x - m) r:   S2 = sum gu xhat = r (w G - m S1),   dw = gamma r (G - S1 E[x] - S2 r (w E[x^2] - m E[x]))
#include "cg_common.h"
#include "context_heads.h"

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

#define CG_CTX_THREADS 1024                                 // sixteen slices of the positions: a sample's loop is 100 iterations per lane, not 400
#define CG_CTX_SLICES (CG_CTX_THREADS / CG_CTX_MAXC)      // a wave = one slice of the positions, a lane = one channel

struct CgCtxChan { float w, m, r, s, beta, alpha; };      // s = gamma * r

// constants of channel c of head h.  Forward, train mode: batch statistics derived from the moments of x; workgroup 0 records them
// (save, running statistics exactly like nn.BatchNorm2d: unbiased variance into running_var).
__device__ __forceinline__ CgCtxChan cg_ctx_chan(const CgCtxHeads& t, int h, int c, bool backward, float mean_x, float var_x, bool owner) {
  CgCtxChan k;
  const CgTailBN& bn = t.bn[h];
  k.w = t.w[h][c]; k.beta = bn.beta[c]; k.alpha = t.alpha[h][0];
  const float gamma = bn.gamma[c];
  if (backward) { k.m = bn.save[c]; k.r = bn.save[t.C + c]; }
  else if (t.train) {
    const double var = (double)k.w * (double)k.w * (double)var_x;
    k.m = k.w * mean_x;
    k.r = (float)(1.0 / sqrt(var + (double)bn.eps));
    if (owner) {
      bn.save[c] = k.m; bn.save[t.C + c] = k.r;
      if (bn.running_mean) {
        const double cnt = (double)t.B * (double)t.P;
        const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        bn.running_mean[c] = (1.f - bn.momentum) * bn.running_mean[c] + bn.momentum * k.m;
        bn.running_var[c] = (1.f - bn.momentum) * bn.running_var[c] + bn.momentum * (float)unb;
        if (c == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
      }
    }
  } else {
    k.m = bn.running_mean[c];
    k.r = 1.0f / sqrtf(bn.running_var[c] + bn.eps);
    if (owner) { bn.save[c] = k.m; bn.save[t.C + c] = k.r; }
  }
  k.s = gamma * k.r;
  return k;
}
__device__ __forceinline__ float cg_ctx_u(const CgCtxChan& k, float x) { return (k.w * x - k.m) * k.s + k.beta; }      // mean first, as nn.BatchNorm

__device__ __forceinline__ void cg_ctx_moments(const CgCtxHeads& t, float& mean_x, float& var_x) {
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < CG_STAT_REPLICAS; ++r) { s1 += t.xstats[2 * r]; s2 += t.xstats[2 * r + 1]; }
  const double cnt = (double)t.B * (double)t.P, mean = s1 / cnt;
  double var = s2 / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  mean_x = (float)mean; var_x = (float)var;
}

__device__ __forceinline__ void cg_ctx_stage(const CgCtxHeads& t, int b, float* sX) {
  const float* __restrict__ xb = t.x + (long long)b * t.P;
  for (int p = threadIdx.x; p < t.P; p += CG_CTX_THREADS) sX[p] = xb[p];
}

// ======================================================================================================================
__global__ __launch_bounds__(CG_CTX_THREADS) void cg_ctx_heads_fwd_kernel(CgCtxHeads t) {
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);                   // [P]
  float* sMax = sX + ((t.P + 3) & ~3);                                // [SLICES][C]
  int* sArg = reinterpret_cast<int*>(sMax + CG_CTX_SLICES * CG_CTX_MAXC);
  float* sSum = reinterpret_cast<float*>(sArg + CG_CTX_SLICES * CG_CTX_MAXC);
  const int b = blockIdx.x, c = threadIdx.x & (CG_CTX_MAXC - 1), q = threadIdx.x / CG_CTX_MAXC;
  cg_ctx_stage(t, b, sX);
  float mean_x = 0.f, var_x = 0.f;
  if (t.train) cg_ctx_moments(t, mean_x, var_x);
  if (b == 0 && threadIdx.x == 0 && t.xsave) { t.xsave[0] = mean_x; t.xsave[1] = var_x; }
  const bool live = c < t.C;
  const int cc = live ? c : 0;
  const CgCtxChan k0 = cg_ctx_chan(t, 0, cc, false, mean_x, var_x, b == 0 && q == 0 && live);
  const CgCtxChan k1 = cg_ctx_chan(t, 1, cc, false, mean_x, var_x, b == 0 && q == 0 && live);
  __syncthreads();
  const int per = (t.P + CG_CTX_SLICES - 1) / CG_CTX_SLICES, p0 = q * per, p1 = min(t.P, p0 + per);
  float best = -INFINITY, sum = 0.f;
  int arg = p0 < t.P ? p0 : 0;
  float* tap0 = (t.tap[0] && live) ? t.tap[0] + ((long long)b * t.C + c) * t.P : nullptr;
  float* tap1 = (t.tap[1] && live) ? t.tap[1] + ((long long)b * t.C + c) * t.P : nullptr;
  for (int p = p0; p < p1; ++p) {
    const float x = sX[p];                                             // the same address in every lane: a broadcast
    const float u0 = cg_ctx_u(k0, x), u1 = cg_ctx_u(k1, x);
    const float z0 = u0 > 0.f ? u0 : k0.alpha * u0, z1 = u1 > 0.f ? u1 : k1.alpha * u1;
    if (z0 > best) { best = z0; arg = p; }
    sum += z1;
    if (tap0) tap0[p] = z0;
    if (tap1) tap1[p] = z1;
  }
  sMax[q * CG_CTX_MAXC + c] = best; sArg[q * CG_CTX_MAXC + c] = arg; sSum[q * CG_CTX_MAXC + c] = sum;
  __syncthreads();
  if (q == 0 && live) {
    float tot = 0.f;
    for (int j = 0; j < CG_CTX_SLICES; ++j) {                          // slices in increasing position order: `>` keeps the first arg-max
      const float v = sMax[j * CG_CTX_MAXC + c];
      if (j == 0 || v > best) { best = v; arg = sArg[j * CG_CTX_MAXC + c]; }
      tot += sSum[j * CG_CTX_MAXC + c];
    }
    t.y[0][(long long)b * t.C + c] = best;
    t.arg[(long long)b * t.C + c] = arg;
    t.y[1][(long long)b * t.C + c] = tot / (float)t.P;
  }
}

// ======================================================================================================================
// backward phase 1: per-channel sums over this sample
__global__ __launch_bounds__(CG_CTX_THREADS) void cg_ctx_heads_bwd_sums_kernel(CgCtxHeads t) {
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);
  double* sRed = reinterpret_cast<double*>(sX + ((t.P + 3) & ~3));      // [SLICES][C][3]; P rounded up to 4 floats keeps it 8-byte aligned
  const int b = blockIdx.x, c = threadIdx.x & (CG_CTX_MAXC - 1), q = threadIdx.x / CG_CTX_MAXC;
  cg_ctx_stage(t, b, sX);
  const bool live = c < t.C;
  const int cc = live ? c : 0;
  const CgCtxChan k1 = cg_ctx_chan(t, 1, cc, true, 0.f, 0.f, false);
  __syncthreads();
  // head 1 (mean): g = dy / P at every position
  const int per = (t.P + CG_CTX_SLICES - 1) / CG_CTX_SLICES, p0 = q * per, p1 = min(t.P, p0 + per);
  double f1 = 0.0, fx = 0.0, neg = 0.0;
  for (int p = p0; p < p1; ++p) {
    const float x = sX[p];
    const float u = cg_ctx_u(k1, x);
    const bool pos = u > 0.f;
    const float f = pos ? 1.f : k1.alpha;
    f1 += (double)f; fx += (double)(f * x);
    if (!pos) neg += (double)u;
  }
  double* slot = sRed + ((long long)q * CG_CTX_MAXC + c) * 3;
  slot[0] = f1; slot[1] = fx; slot[2] = neg;
  __syncthreads();
  if (q == 0 && live) {
    double a1 = 0.0, ax = 0.0, an = 0.0;
    for (int j = 0; j < CG_CTX_SLICES; ++j) { const double* s = sRed + ((long long)j * CG_CTX_MAXC + c) * 3; a1 += s[0]; ax += s[1]; an += s[2]; }
    const double g = (double)t.dy[1][(long long)b * t.C + c] / (double)t.P;
    double* red = t.red + (((long long)(b % CG_STAT_REPLICAS) * 2 + 1) * t.C + c) * 3;
    atomicAdd(&red[0], g * a1); atomicAdd(&red[1], g * ax); atomicAdd(&red[2], g * an);
  }
  if (q == 1 && live) {
    // head 0 (max): the gradient lands on the arg-max position only
    const CgCtxChan k0 = cg_ctx_chan(t, 0, c, true, 0.f, 0.f, false);
    const float g = t.dy[0][(long long)b * t.C + c];
    const float x = sX[t.arg[(long long)b * t.C + c]];
    const float u = cg_ctx_u(k0, x);
    const bool pos = u > 0.f;
    const float gu = pos ? g : k0.alpha * g;
    double* red = t.red + (((long long)(b % CG_STAT_REPLICAS) * 2 + 0) * t.C + c) * 3;
    atomicAdd(&red[0], (double)gu); atomicAdd(&red[1], (double)gu * (double)x);
    if (!pos) atomicAdd(&red[2], (double)g * (double)u);
  }
}

// backward phase 2: dx of this sample; workgroup 0 also writes the parameter gradients
__global__ __launch_bounds__(CG_CTX_THREADS) void cg_ctx_heads_bwd_apply_kernel(CgCtxHeads t) {
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);                   // [P]
  float* sDx = sX + ((t.P + 3) & ~3);                                 // [P] scatter target of head 0
  float* sK = sDx + ((t.P + 3) & ~3);                                 // [C][6]: w, m, s, beta of head 1; e = w s dy1 / P; -
  double* sRed = reinterpret_cast<double*>(sK + 6 * CG_CTX_MAXC);      // [16] block sums (an even number of floats in front of it)
  __shared__ float sConst[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  cg_ctx_stage(t, b, sX);
  for (int p = tid; p < t.P; p += CG_CTX_THREADS) sDx[p] = 0.f;
  // totals of the channel sums; thread (h, c) for tid < 2 * MAXC
  const int h = tid / CG_CTX_MAXC, c = tid & (CG_CTX_MAXC - 1);
  const bool live = h < 2 && c < t.C;
  double k1 = 0.0, k2 = 0.0, k3 = 0.0;
  const double cnt = (double)t.B * (double)t.P;
  const float mean_x = t.train ? t.xsave[0] : 0.f, var_x = t.train ? t.xsave[1] : 0.f;
  const double ex2 = (double)var_x + (double)mean_x * (double)mean_x;
  CgCtxChan k = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  double S1 = 0.0, G = 0.0, A = 0.0;
  if (live) {
    k = cg_ctx_chan(t, h, c, true, 0.f, 0.f, false);
    for (int r = 0; r < CG_STAT_REPLICAS; ++r) {
      const double* red = t.red + (((long long)r * 2 + h) * t.C + c) * 3;
      S1 += red[0]; G += red[1]; A += red[2];
    }
  }
  const double S2 = (double)k.r * ((double)k.w * G - (double)k.m * S1);
  if (live && t.train) {
    const double ws = (double)k.w * (double)k.s;
    k1 = ws * S1 / cnt;
    k2 = ws * (double)k.w * (double)k.r * S2 / cnt;
    k3 = ws * (double)k.r * (double)k.m * S2 / cnt;
  }
  k1 = cg_block_sum(k1, sRed); k2 = cg_block_sum(k2, sRed); k3 = cg_block_sum(k3, sRed);
  if (tid == 0) { sConst[0] = (float)k1; sConst[1] = (float)k2; sConst[2] = (float)k3; }
  if (live && h == 1) {
    float* kk = sK + 6 * c;
    kk[0] = k.w; kk[1] = k.m; kk[2] = k.s; kk[3] = k.beta;
    kk[4] = k.w * k.s * (t.dy[1][(long long)b * t.C + c] / (float)t.P);
  }
  __syncthreads();
  if (live && h == 0) {                       // head 0: scatter to the arg-max position
    const int a = t.arg[(long long)b * t.C + c];
    const float g = t.dy[0][(long long)b * t.C + c];
    const float u = cg_ctx_u(k, sX[a]);
    atomicAdd(&sDx[a], k.w * k.s * (u > 0.f ? g : k.alpha * g));
  }
  const float alpha1 = t.alpha[1][0];
  __syncthreads();
  float* __restrict__ dxb = t.dx + (long long)b * t.P;
  for (int p = tid; p < t.P; p += CG_CTX_THREADS) {
    const float x = sX[p];
    float acc = sDx[p];
    for (int j = 0; j < t.C; ++j) {
      const float* kk = sK + 6 * j;                                    // the same address in every lane
      const float u = (kk[0] * x - kk[1]) * kk[2] + kk[3];
      acc += u > 0.f ? kk[4] : alpha1 * kk[4];
    }
    dxb[p] = acc - sConst[0] - sConst[1] * x + sConst[2];
  }
  if (b == 0 && live) {
    float dw;
    if (t.train) dw = (float)((double)k.s * (G - S1 * (double)mean_x - S2 * (double)k.r * ((double)k.w * ex2 - (double)k.m * (double)mean_x)));
    else dw = (float)((double)k.s * G);
    t.dw[h][c] = dw; t.dgamma[h][c] = (float)S2; t.dbeta[h][c] = (float)S1;
  }
  // slope gradients: sum over the channels (workgroup 0)
  double a0 = (live && h == 0) ? A : 0.0, a1 = (live && h == 1) ? A : 0.0;
  a0 = cg_block_sum(a0, sRed); a1 = cg_block_sum(a1, sRed);
  if (b == 0 && tid == 0) { t.dalpha[0][0] = (float)a0; t.dalpha[1][0] = (float)a1; }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_ctx_check(const CgCtxHeads* t, bool bwd) {
  if (!t) return CG_EARG;
  if (t->B <= 0 || t->P <= 0 || t->C <= 0 || t->C > CG_CTX_MAXC || t->P > 16384) return CG_ESHAPE;
  if (!t->x || !t->arg) return CG_EARG;
  if (t->train && (!t->xsave || (!bwd && !t->xstats))) return CG_EARG;
  for (int h = 0; h < 2; ++h) {
    if (!t->w[h] || !t->bn[h].gamma || !t->bn[h].beta || !t->bn[h].save || !t->alpha[h]) return CG_EARG;
    if (!bwd && !t->train && (!t->bn[h].running_mean || !t->bn[h].running_var)) return CG_EARG;
    if (!bwd && !t->y[h]) return CG_EARG;
    if (bwd && (!t->dy[h] || !t->dw[h] || !t->dgamma[h] || !t->dbeta[h] || !t->dalpha[h])) return CG_EARG;
  }
  if (bwd && (!t->red || !t->dx)) return CG_EARG;
  return CG_OK;
}

extern "C" long long cg_context_heads_red_doubles(int C) { return (long long)CG_STAT_REPLICAS * 2 * C * 3; }

// include/cistgcn_hip.h : cg_context_heads_fwd / cg_context_heads_bwd
extern "C" int cg_context_heads_fwd(const CgCtxHeads* t, void* stream_) {
  int st = cg_ctx_check(t, false);
  if (st != CG_OK) return st;
  const size_t P4 = (size_t)((t->P + 3) & ~3);
  const size_t lds = (P4 + (size_t)3 * CG_CTX_SLICES * CG_CTX_MAXC) * sizeof(float);
  hipLaunchKernelGGL(cg_ctx_heads_fwd_kernel, dim3((unsigned)t->B), dim3(CG_CTX_THREADS), lds, (hipStream_t)stream_, *t);
  return cg_launch_status();
}

extern "C" int cg_context_heads_bwd(const CgCtxHeads* t, void* stream_) {
  int st = cg_ctx_check(t, true);
  if (st != CG_OK) return st;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t P4 = (size_t)((t->P + 3) & ~3);
  size_t lds = P4 * sizeof(float) + (size_t)3 * CG_CTX_SLICES * CG_CTX_MAXC * sizeof(double);
  hipLaunchKernelGGL(cg_ctx_heads_bwd_sums_kernel, dim3((unsigned)t->B), dim3(CG_CTX_THREADS), lds, stream, *t);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  lds = (2 * P4 + (size_t)6 * CG_CTX_MAXC) * sizeof(float) + 16 * sizeof(double);
  hipLaunchKernelGGL(cg_ctx_heads_bwd_apply_kernel, dim3((unsigned)t->B), dim3(CG_CTX_THREADS), lds, stream, *t);
  return cg_launch_status();
}
