// Tail of the two gate paths of a DSTD_GC block (reference: conv_s / conv_t slots 5-7 and map_s / map_t, CISTGCN.py:337-352, applied
// by :378-384):   z (B,C) -> BatchNorm2d -> Dropout -> PReLU -> cat with the block statistics (B, 2+2T) -> Linear -> BatchNorm1d ->
// Dropout -> PReLU -> Linear = the gate w1 / w2 (B,C).
//
// Every tensor here is (B, <= C + S) - a few hundred KB at B = 256 - but as separate operators the chain was seven launches forward
// (channel sums, row kernel, two concatenation copies, contraction, row kernel, contraction) and as many backward, each 10-45 us of
// launch latency and split-K bookkeeping: ~110 us forward and ~90 us backward per block, 1.1 ms of the 22 ms step at B = 256 and a
// quarter of the launches of the B = 16 step.  Here ONE workgroup per gate path holds the whole batch: the three batch statistics
// are workgroup reductions, the two Linear layers walk the batch in chunks of sixteen samples (a wave = a sample, a lane = an
// output column, weights transposed in LDS), the concatenation is an index.  Forward and backward are one launch each.
#include "cg_common.h"
#include "cg_phase.h"
#include "gate_head.h"

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

#define CG_GATE_THREADS 1024
#define CG_GATE_WAVES (CG_GATE_THREADS / CG_WAVE)
#define CG_GATE_CP 64                      // padded column count = lanes of a wave

struct CgGateAff { float mean, rstd, scale, beta; };

// BatchNorm constants of column c from the workgroup's own batch sums (train) or the running statistics (eval); thread c of wave 0
// calls this (forward): it records save / running statistics exactly like nn.BatchNorm
__device__ __forceinline__ CgGateAff cg_gate_aff_fwd(const CgTailBN& bn, int c, int C, int B, int train, double s1, double s2) {
  CgGateAff a;
  const float gamma = bn.gamma[c];
  a.beta = bn.beta[c];
  if (train) {
    const double cnt = (double)B, mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    a.mean = (float)mean;
    a.rstd = (float)(1.0 / sqrt(var + (double)bn.eps));
    if (bn.running_mean) {
      const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      bn.running_mean[c] = (1.f - bn.momentum) * bn.running_mean[c] + bn.momentum * (float)mean;
      bn.running_var[c] = (1.f - bn.momentum) * bn.running_var[c] + bn.momentum * (float)unb;
      if (c == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
    }
  } else {
    a.mean = bn.running_mean[c];
    a.rstd = 1.0f / sqrtf(bn.running_var[c] + bn.eps);
  }
  bn.save[c] = a.mean; bn.save[C + c] = a.rstd;
  a.scale = gamma * a.rstd;
  return a;
}
__device__ __forceinline__ CgGateAff cg_gate_aff_bwd(const CgTailBN& bn, int c, int C) {
  CgGateAff a;
  a.mean = bn.save[c]; a.rstd = bn.save[C + c]; a.scale = bn.gamma[c] * a.rstd; a.beta = bn.beta[c];
  return a;
}
// pre-activation behind BatchNorm and Dropout (cg_norm_act's expression)
__device__ __forceinline__ float cg_gate_u(const float* aff, float x, float keep) { return ((x - aff[0]) * aff[2] + aff[3]) * keep; }

// column sums over the waves: sRed[wave][lane] (f64) -> total in every thread of wave 0 (valid for lane < C); two barriers
__device__ __forceinline__ double cg_gate_colsum(double v, double* sRed, int wave, int lane) {
  __syncthreads();
  sRed[wave * CG_GATE_CP + lane] = v;
  __syncthreads();
  double s = 0.0;
  if (wave == 0) for (int w = 0; w < CG_GATE_WAVES; ++w) s += sRed[w * CG_GATE_CP + lane];
  return s;
}

// ======================================================================================================================
__global__ __launch_bounds__(CG_GATE_THREADS) void cg_gate_head_fwd_kernel(CgGateHead t) {
  const CgGatePath& p = t.p[blockIdx.x];
  const int B = t.B, C = t.C, S = t.S, K = C + S, KP = (K + 3) & ~3;
  float* sWl = reinterpret_cast<float*>(cg_dyn_lds);                 // [K][CP]   Wl transposed
  float* sW2 = sWl + K * CG_GATE_CP;                                  // [C][CP]   W2 transposed
  float* sU = sW2 + C * CG_GATE_CP;                                   // [WAVES][KP] input rows of the current chunk
  float* sAff = sU + CG_GATE_WAVES * KP;                              // [2][CP][4] mean, rstd, scale, beta of bn2 / bn3
  double* sRed = reinterpret_cast<double*>(sAff + 2 * CG_GATE_CP * 4 + ((K * CG_GATE_CP + C * CG_GATE_CP + CG_GATE_WAVES * KP) & 1));
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const bool drop = t.train && t.drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  for (int e = tid; e < C * K; e += CG_GATE_THREADS) { const int o = e / K, i = e - o * K; sWl[i * CG_GATE_CP + o] = p.Wl[e]; }
  for (int e = tid; e < C * C; e += CG_GATE_THREADS) { const int o = e / C, j = e - o * C; sW2[j * CG_GATE_CP + o] = p.W2[e]; }
  // ---- BatchNorm of z over the batch
  double s1 = 0.0, s2 = 0.0;
  if (t.train && lane < C)
    for (int b = wave; b < B; b += CG_GATE_WAVES) { const double v = (double)p.z[(long long)b * C + lane]; s1 += v; s2 += v * v; }
  s1 = cg_gate_colsum(s1, sRed, wave, lane);
  s2 = cg_gate_colsum(s2, sRed, wave, lane);
  if (wave == 0 && lane < C) {
    const CgGateAff a = cg_gate_aff_fwd(p.bn2, lane, C, B, t.train, s1, s2);
    float* k = sAff + 4 * lane;
    k[0] = a.mean; k[1] = a.rstd; k[2] = a.scale; k[3] = a.beta;
  }
  __syncthreads();
  const float alpha2 = p.alpha2[0], alpha3 = p.alpha3[0];
  // ---- y = Wl [PReLU(Dropout(BN(z))) | stats], sixteen samples per round (a wave = a sample)
  s1 = 0.0; s2 = 0.0;
  for (int b0 = 0; b0 < B; b0 += CG_GATE_WAVES) {
    const int b = b0 + wave;
    __syncthreads();
    if (b < B)
      for (int i = lane; i < K; i += CG_WAVE) {
        float v;
        if (i < C) {
          const float keep = drop ? cg_drop_scale(t.drop_p, seed, p.salt2, (unsigned long long)b * C + i) : 1.f;
          const float u = cg_gate_u(sAff + 4 * i, p.z[(long long)b * C + i], keep);
          v = u > 0.f ? u : alpha2 * u;
          if (p.tap2) p.tap2[(long long)b * C + i] = v;
        } else v = p.stats[(long long)b * p.stats_ld + (i - C)];
        sU[wave * KP + i] = v;
      }
    __syncthreads();
    if (b < B && lane < C) {
      const float* u = sU + wave * KP;
      float acc = 0.f;
      for (int i = 0; i < K; ++i) acc += sWl[i * CG_GATE_CP + lane] * u[i];
      p.y[(long long)b * C + lane] = acc;
      s1 += (double)acc; s2 += (double)acc * (double)acc;
    }
  }
  // ---- BatchNorm1d of y over the batch
  s1 = cg_gate_colsum(s1, sRed, wave, lane);
  s2 = cg_gate_colsum(s2, sRed, wave, lane);
  if (wave == 0 && lane < C) {
    const CgGateAff a = cg_gate_aff_fwd(p.bn3, lane, C, B, t.train, s1, s2);
    float* k = sAff + 4 * (CG_GATE_CP + lane);
    k[0] = a.mean; k[1] = a.rstd; k[2] = a.scale; k[3] = a.beta;
  }
  __syncthreads();
  // ---- w = W2 PReLU(Dropout(BN(y)))
  for (int b0 = 0; b0 < B; b0 += CG_GATE_WAVES) {
    const int b = b0 + wave;
    __syncthreads();
    if (b < B && lane < C) {
      const float keep = drop ? cg_drop_scale(t.drop_p, seed, p.salt3, (unsigned long long)b * C + lane) : 1.f;
      const float u = cg_gate_u(sAff + 4 * (CG_GATE_CP + lane), p.y[(long long)b * C + lane], keep);   // written by this thread above
      const float v = u > 0.f ? u : alpha3 * u;
      if (p.tap3) p.tap3[(long long)b * C + lane] = v;
      sU[wave * KP + lane] = v;
    }
    __syncthreads();
    if (b < B && lane < C) {
      const float* h = sU + wave * KP;
      float acc = 0.f;
      for (int j = 0; j < C; ++j) acc += sW2[j * CG_GATE_CP + lane] * h[j];
      p.w[(long long)b * C + lane] = acc;
    }
  }
}

// ======================================================================================================================
#define CG_GATE_ACC2 4          // C * C <= THREADS * ACC2
#define CG_GATE_ACCL 16         // C * (C + S) <= THREADS * ACCL

__global__ __launch_bounds__(CG_GATE_THREADS) void cg_gate_head_bwd_kernel(CgGateHead t) {
  const CgGatePath& p = t.p[blockIdx.x];
  const int B = t.B, C = t.C, S = t.S, K = C + S, KP = (K + 3) & ~3;
  float* sWl = reinterpret_cast<float*>(cg_dyn_lds);                 // [C][K]   Wl as it lies in memory
  float* sW2 = sWl + C * K;                                           // [C][C]
  float* sA = sW2 + C * C;                                            // [WAVES][CP] dw rows, then dy rows
  float* sH = sA + CG_GATE_WAVES * CG_GATE_CP;                        // [WAVES][KP] h3 rows, then u rows
  float* sAff = sH + CG_GATE_WAVES * KP;                              // [2][CP][4]
  double* sRed = reinterpret_cast<double*>(sAff + 2 * CG_GATE_CP * 4 + ((C * K + C * C + CG_GATE_WAVES * KP) & 1));
  float* G3 = p.scratch;                                              // (B,C) gradient in front of bn3
  float* G2 = p.scratch + (long long)B * C;                           // (B,C) gradient in front of bn2
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const bool drop = t.train && t.drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  for (int e = tid; e < C * K; e += CG_GATE_THREADS) sWl[e] = p.Wl[e];
  for (int e = tid; e < C * C; e += CG_GATE_THREADS) sW2[e] = p.W2[e];
  if (wave == 0 && lane < C) {
    const CgGateAff a2 = cg_gate_aff_bwd(p.bn2, lane, C), a3 = cg_gate_aff_bwd(p.bn3, lane, C);
    float* k = sAff + 4 * lane;
    k[0] = a2.mean; k[1] = a2.rstd; k[2] = a2.scale; k[3] = a2.beta;
    k = sAff + 4 * (CG_GATE_CP + lane);
    k[0] = a3.mean; k[1] = a3.rstd; k[2] = a3.scale; k[3] = a3.beta;
  }
  __syncthreads();
  const float alpha2 = p.alpha2[0], alpha3 = p.alpha3[0];
  float acc2[CG_GATE_ACC2], accl[CG_GATE_ACCL];
#pragma unroll
  for (int q = 0; q < CG_GATE_ACC2; ++q) acc2[q] = 0.f;
#pragma unroll
  for (int q = 0; q < CG_GATE_ACCL; ++q) accl[q] = 0.f;

  // ---- phase 1: through the last Linear and the PReLU / Dropout behind bn3; dW2
  double S1 = 0.0, S2 = 0.0, SA = 0.0;
  const float* k3 = sAff + 4 * (CG_GATE_CP + lane);
  for (int b0 = 0; b0 < B; b0 += CG_GATE_WAVES) {
    const int b = b0 + wave, nb = min(CG_GATE_WAVES, B - b0);
    __syncthreads();
    float u3 = 0.f, keep3 = 1.f, yv = 0.f;
    if (b < B && lane < C) {
      sA[wave * CG_GATE_CP + lane] = p.dw[(long long)b * C + lane];
      keep3 = drop ? cg_drop_scale(t.drop_p, seed, p.salt3, (unsigned long long)b * C + lane) : 1.f;
      yv = p.y[(long long)b * C + lane];
      u3 = cg_gate_u(k3, yv, keep3);
      sH[wave * KP + lane] = u3 > 0.f ? u3 : alpha3 * u3;
    }
    __syncthreads();
    if (b < B && lane < C) {
      const float* dwr = sA + wave * CG_GATE_CP;
      float dh = 0.f;
      for (int o = 0; o < C; ++o) dh += dwr[o] * sW2[o * C + lane];
      const float gu = u3 > 0.f ? dh : alpha3 * dh;
      if (!(u3 > 0.f)) SA += (double)dh * (double)u3;
      const float g = gu * keep3;
      S1 += (double)g; S2 += (double)g * (double)((yv - k3[0]) * k3[1]);
      G3[(long long)b * C + lane] = g;
    }
#pragma unroll
    for (int q = 0; q < CG_GATE_ACC2; ++q) {
      const int e = tid + CG_GATE_THREADS * q;
      if (e < C * C) {
        const int o = e / C, j = e - o * C;
        float a = 0.f;
        for (int w = 0; w < nb; ++w) a += sA[w * CG_GATE_CP + o] * sH[w * KP + j];
        acc2[q] += a;
      }
    }
  }
  S1 = cg_gate_colsum(S1, sRed, wave, lane);
  S2 = cg_gate_colsum(S2, sRed, wave, lane);
  SA = cg_gate_colsum(SA, sRed, wave, lane);
  __shared__ float sM[2][CG_GATE_CP][2];
  __shared__ double sAlpha[CG_GATE_CP];
  if (wave == 0) {
    if (lane < C) {
      sM[1][lane][0] = t.train ? (float)(S1 / (double)B) : 0.f; sM[1][lane][1] = t.train ? (float)(S2 / (double)B) : 0.f;
      p.dgamma3[lane] = (float)S2; p.dbeta3[lane] = (float)S1;
    }
    sAlpha[lane] = lane < C ? SA : 0.0;
  }
  __syncthreads();
  if (tid == 0) { double s = 0.0; for (int c = 0; c < C; ++c) s += sAlpha[c]; p.dalpha3[0] = (float)s; }

  // ---- phase 2: through bn3 and the first Linear; the statistics' gradient; PReLU / Dropout behind bn2; dWl
  S1 = 0.0; S2 = 0.0; SA = 0.0;
  const float* k2 = sAff + 4 * lane;
  for (int b0 = 0; b0 < B; b0 += CG_GATE_WAVES) {
    const int b = b0 + wave, nb = min(CG_GATE_WAVES, B - b0);
    __syncthreads();
    float u2 = 0.f, keep2 = 1.f, zv = 0.f;
    if (b < B) {
      if (lane < C) {
        const float g = G3[(long long)b * C + lane];                        // written by this thread in phase 1
        const float xh = (p.y[(long long)b * C + lane] - k3[0]) * k3[1];
        sA[wave * CG_GATE_CP + lane] = k3[2] * (g - sM[1][lane][0] - xh * sM[1][lane][1]);
        keep2 = drop ? cg_drop_scale(t.drop_p, seed, p.salt2, (unsigned long long)b * C + lane) : 1.f;
        zv = p.z[(long long)b * C + lane];
        u2 = cg_gate_u(k2, zv, keep2);
        sH[wave * KP + lane] = u2 > 0.f ? u2 : alpha2 * u2;
      }
      for (int i = C + lane; i < K; i += CG_WAVE) sH[wave * KP + i] = p.stats[(long long)b * p.stats_ld + (i - C)];
    }
    __syncthreads();
    if (b < B) {
      const float* dyr = sA + wave * CG_GATE_CP;
      for (int i = lane; i < K; i += CG_WAVE) {
        float du = 0.f;
        for (int o = 0; o < C; ++o) du += dyr[o] * sWl[o * K + i];
        if (i >= C) p.dstats[(long long)b * S + (i - C)] = du;
        else {                                                            // i == lane
          const float gu = u2 > 0.f ? du : alpha2 * du;
          if (!(u2 > 0.f)) SA += (double)du * (double)u2;
          const float g = gu * keep2;
          S1 += (double)g; S2 += (double)g * (double)((zv - k2[0]) * k2[1]);
          G2[(long long)b * C + lane] = g;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CG_GATE_ACCL; ++q) {
      const int e = tid + CG_GATE_THREADS * q;
      if (e < C * K) {
        const int o = e / K, i = e - o * K;
        float a = 0.f;
        for (int w = 0; w < nb; ++w) a += sA[w * CG_GATE_CP + o] * sH[w * KP + i];
        accl[q] += a;
      }
    }
  }
  S1 = cg_gate_colsum(S1, sRed, wave, lane);
  S2 = cg_gate_colsum(S2, sRed, wave, lane);
  SA = cg_gate_colsum(SA, sRed, wave, lane);
  if (wave == 0) {
    if (lane < C) {
      sM[0][lane][0] = t.train ? (float)(S1 / (double)B) : 0.f; sM[0][lane][1] = t.train ? (float)(S2 / (double)B) : 0.f;
      p.dgamma2[lane] = (float)S2; p.dbeta2[lane] = (float)S1;
    }
    sAlpha[lane] = lane < C ? SA : 0.0;
  }
  __syncthreads();
  if (tid == 0) { double s = 0.0; for (int c = 0; c < C; ++c) s += sAlpha[c]; p.dalpha2[0] = (float)s; }

  // ---- phase 3: through bn2
  if (lane < C)
    for (int b = wave; b < B; b += CG_GATE_WAVES) {
      const float xh = (p.z[(long long)b * C + lane] - k2[0]) * k2[1];
      p.dz[(long long)b * C + lane] = k2[2] * (G2[(long long)b * C + lane] - sM[0][lane][0] - xh * sM[0][lane][1]);     // G2: this thread's own
    }
#pragma unroll
  for (int q = 0; q < CG_GATE_ACC2; ++q) { const int e = tid + CG_GATE_THREADS * q; if (e < C * C) p.dW2[e] = acc2[q]; }
#pragma unroll
  for (int q = 0; q < CG_GATE_ACCL; ++q) { const int e = tid + CG_GATE_THREADS * q; if (e < C * K) p.dWl[e] = accl[q]; }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_gate_check(const CgGateHead* t, bool bwd) {
  if (!t || t->n < 1 || t->n > 2) return CG_EARG;
  if (t->B <= 0 || t->C <= 0 || t->C > CG_GATE_MAXC || t->S < 0 || t->S > CG_GATE_MAXS) return CG_ESHAPE;
  if (t->train && t->B < 2) return CG_ESHAPE;
  if (t->train && t->drop_p > 0.f && !t->seed) return CG_EARG;
  if (t->drop_p < 0.f || t->drop_p >= 1.f) return CG_EARG;
  for (int i = 0; i < t->n; ++i) {
    const CgGatePath& p = t->p[i];
    if (!p.z || (t->S > 0 && !p.stats) || !p.Wl || !p.W2 || !p.alpha2 || !p.alpha3 || !p.y) return CG_EARG;
    if (!p.bn2.gamma || !p.bn2.beta || !p.bn2.save || !p.bn3.gamma || !p.bn3.beta || !p.bn3.save) return CG_EARG;
    if (!bwd && (!p.w || (!t->train && (!p.bn2.running_mean || !p.bn2.running_var || !p.bn3.running_mean || !p.bn3.running_var)))) return CG_EARG;
    if (bwd && (!p.dw || !p.dz || (t->S > 0 && !p.dstats) || !p.dWl || !p.dW2 || !p.dgamma2 || !p.dbeta2 || !p.dalpha2 || !p.dgamma3 || !p.dbeta3 ||
                !p.dalpha3 || !p.scratch)) return CG_EARG;
  }
  return CG_OK;
}

extern "C" int cg_gate_head_supported(int B, int C, int S) { return (B > 0 && C > 0 && C <= CG_GATE_MAXC && S >= 0 && S <= CG_GATE_MAXS) ? 1 : 0; }
extern "C" long long cg_gate_head_scratch_floats(int B, int C, int S) { return (long long)B * (2 * C + S); }

// include/cistgcn_hip.h : cg_gate_head_fwd / cg_gate_head_bwd
extern "C" int cg_gate_head_fwd(const CgGateHead* t, void* stream_) {
  int st = cg_gate_check(t, false);
  if (st != CG_OK) return st;
  const int K = t->C + t->S, KP = (K + 3) & ~3;
  const size_t lds = ((size_t)K * CG_GATE_CP + (size_t)t->C * CG_GATE_CP + (size_t)CG_GATE_WAVES * KP + 2 * CG_GATE_CP * 4 + 1) * sizeof(float) +
                     (size_t)CG_GATE_WAVES * CG_GATE_CP * sizeof(double);
  if (lds > 64 * 1024 && cg_lds_limit((const void*)cg_gate_head_fwd_kernel, lds) != hipSuccess) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_gate_head_fwd_kernel, dim3((unsigned)t->n), dim3(CG_GATE_THREADS), lds, (hipStream_t)stream_, *t);
  return cg_launch_status();
}

extern "C" int cg_gate_head_bwd(const CgGateHead* t, void* stream_) {
  int st = cg_gate_check(t, true);
  if (st != CG_OK) return st;
  const int K = t->C + t->S, KP = (K + 3) & ~3;
  if ((long long)t->C * t->C > (long long)CG_GATE_THREADS * CG_GATE_ACC2 || (long long)t->C * K > (long long)CG_GATE_THREADS * CG_GATE_ACCL) return CG_ESHAPE;
  const size_t lds = ((size_t)t->C * K + (size_t)t->C * t->C + (size_t)CG_GATE_WAVES * CG_GATE_CP + (size_t)CG_GATE_WAVES * KP + 2 * CG_GATE_CP * 4 + 1) * sizeof(float) +
                     (size_t)CG_GATE_WAVES * CG_GATE_CP * sizeof(double);
  if (lds > 64 * 1024 && cg_lds_limit((const void*)cg_gate_head_bwd_kernel, lds) != hipSuccess) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_gate_head_bwd_kernel, dim3((unsigned)t->n), dim3(CG_GATE_THREADS), lds, (hipStream_t)stream_, *t);
  return cg_launch_status();
}
