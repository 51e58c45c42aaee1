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
#define CG_GATE_NS 16                      // samples per thread: B <= WAVES * NS = 256

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
  float* sWl = reinterpret_cast<float*>(cg_dyn_lds);                 // [KP][CP]  Wl transposed, rows K .. KP-1 zero
  float* sW2 = sWl + KP * CG_GATE_CP;                                 // [C][CP]   W2 transposed
  float* sU = sW2 + C * CG_GATE_CP;                                   // [WAVES][KP] input rows of the current chunk
  float* sAff = sU + CG_GATE_WAVES * KP;                              // [2][CP][4] mean, rstd, scale, beta of bn2 / bn3
  double* sRed = reinterpret_cast<double*>(sAff + 2 * CG_GATE_CP * 4);      // an even number of floats in front of it
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const bool drop = t.train && t.drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  for (int e = tid; e < (KP - K) * CG_GATE_CP; e += CG_GATE_THREADS) sWl[K * CG_GATE_CP + e] = 0.f;
  for (int e = tid; e < C * K; e += CG_GATE_THREADS) { const int o = e / K, i = e - o * K; sWl[i * CG_GATE_CP + o] = p.Wl[e]; }
  for (int e = tid; e < C * C; e += CG_GATE_THREADS) { const int o = e / C, j = e - o * C; sW2[j * CG_GATE_CP + o] = p.W2[e]; }
  // this thread's share of the batch: samples b = wave + 16 k, column `lane`.  z is loaded once, y stays in registers between the two Linear
  // layers: inside the rounds below nothing waits for HBM but the statistics rows, and those are requested one round ahead (with the loads
  // inside the rounds a 256-sample batch took 140 us: sixteen dependent round trips per pass)
  float zreg[CG_GATE_NS], yreg[CG_GATE_NS];
#pragma unroll
  for (int k = 0; k < CG_GATE_NS; ++k) {
    const int b = wave + CG_GATE_WAVES * k;
    zreg[k] = (b < B && lane < C) ? p.z[(long long)b * C + lane] : 0.f;
    yreg[k] = 0.f;
  }
  // ---- BatchNorm of z over the batch
  double s1 = 0.0, s2 = 0.0;
  if (t.train) {
#pragma unroll
    for (int k = 0; k < CG_GATE_NS; ++k) if (wave + CG_GATE_WAVES * k < B) { const double v = (double)zreg[k]; s1 += v; s2 += v * v; }
  }
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
  float st0 = 0.f, st1 = 0.f, st2 = 0.f;                 // statistics columns C + lane (+ 64, + 128) of the next round's sample
  auto load_stats = [&](int b) {
    const float* row = p.stats + (long long)b * p.stats_ld;
    st0 = (b < B && lane < S) ? row[lane] : 0.f;
    st1 = (b < B && lane + 64 < S) ? row[lane + 64] : 0.f;
    st2 = (b < B && lane + 128 < S) ? row[lane + 128] : 0.f;
  };
  load_stats(wave);
#pragma unroll
  for (int k = 0; k < CG_GATE_NS; ++k) {
    const int b = wave + CG_GATE_WAVES * k;
    if (CG_GATE_WAVES * k >= B) break;                                 // uniform
    __syncthreads();
    if (b < B) {
      float* u = sU + wave * KP;
      if (lane < C) {
        const float keep = drop ? cg_drop_scale(t.drop_p, seed, p.salt2, (unsigned long long)b * C + lane) : 1.f;
        const float uu = cg_gate_u(sAff + 4 * lane, zreg[k], keep);
        const float v = uu > 0.f ? uu : alpha2 * uu;
        if (p.tap2) p.tap2[(long long)b * C + lane] = v;
        u[lane] = v;
      }
      if (lane < S) u[C + lane] = st0;
      if (lane + 64 < S) u[C + lane + 64] = st1;
      if (lane + 128 < S) u[C + lane + 128] = st2;
      for (int i = K + lane; i < KP; i += CG_WAVE) u[i] = 0.f;
    }
    load_stats(b + CG_GATE_WAVES);
    __syncthreads();
    if (b < B && lane < C) {
      const float* u = sU + wave * KP;
      float acc = 0.f;
      for (int i = 0; i < KP; i += 4) {
        const float4 uv = *reinterpret_cast<const float4*>(u + i);      // the same address in every lane: a broadcast
        acc += sWl[i * CG_GATE_CP + lane] * uv.x;
        acc += sWl[(i + 1) * CG_GATE_CP + lane] * uv.y;
        acc += sWl[(i + 2) * CG_GATE_CP + lane] * uv.z;
        acc += sWl[(i + 3) * CG_GATE_CP + lane] * uv.w;
      }
      yreg[k] = acc;
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
#pragma unroll
  for (int k = 0; k < CG_GATE_NS; ++k) {
    const int b = wave + CG_GATE_WAVES * k;
    if (CG_GATE_WAVES * k >= B) break;
    __syncthreads();
    if (b < B && lane < C) {
      const float keep = drop ? cg_drop_scale(t.drop_p, seed, p.salt3, (unsigned long long)b * C + lane) : 1.f;
      const float uu = cg_gate_u(sAff + 4 * (CG_GATE_CP + lane), yreg[k], keep);
      const float v = uu > 0.f ? uu : alpha3 * uu;
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
#define CG_GATE_ACCL 11         // C * (C + S) <= THREADS * ACCL (64 x 166 = 10624 <= 11264)

__global__ __launch_bounds__(CG_GATE_THREADS) void cg_gate_head_bwd_kernel(CgGateHead t) {
  const CgGatePath& p = t.p[blockIdx.x];
  const int B = t.B, C = t.C, S = t.S, K = C + S, KP = (K + 3) & ~3;
  float* sWl = reinterpret_cast<float*>(cg_dyn_lds);                 // [C][K]   Wl as it lies in memory
  float* sW2 = sWl + C * K;                                           // [C][C]
  float* sA = sW2 + C * C;                                            // [WAVES][CP] dw rows, then dy rows
  float* sH = sA + CG_GATE_WAVES * CG_GATE_CP;                        // [WAVES][KP] h3 rows, then u rows
  float* sAff = sH + CG_GATE_WAVES * KP;                              // [2][CP][4]
  double* sRed = reinterpret_cast<double*>(sAff + 2 * CG_GATE_CP * 4 + ((C * K + C * C) & 1));
  __shared__ float sM[2][CG_GATE_CP][2];
  __shared__ double sAlpha[CG_GATE_CP];
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const bool drop = t.train && t.drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  // this thread's share of the batch (samples wave + 16 k, column lane) lives in registers through all three phases: dw / y / z are loaded
  // once, the gradients in front of the two BatchNorms (g3, g2) never leave the registers
  float ra[CG_GATE_NS], ry[CG_GATE_NS], rg[CG_GATE_NS];               // ra: dw, later z; ry: y; rg: g3, later g2
#pragma unroll
  for (int k = 0; k < CG_GATE_NS; ++k) {
    const int b = wave + CG_GATE_WAVES * k;
    const bool ok = b < B && lane < C;
    ra[k] = ok ? p.dw[(long long)b * C + lane] : 0.f;
    ry[k] = ok ? p.y[(long long)b * C + lane] : 0.f;
    rg[k] = 0.f;
  }
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
#pragma unroll
  for (int k = 0; k < CG_GATE_NS; ++k) {
    const int b0 = CG_GATE_WAVES * k, b = b0 + wave, nb = min(CG_GATE_WAVES, B - b0);
    if (b0 >= B) break;                                                // uniform
    __syncthreads();
    float u3 = 0.f, keep3 = 1.f;
    if (b < B && lane < C) {
      sA[wave * CG_GATE_CP + lane] = ra[k];
      keep3 = drop ? cg_drop_scale(t.drop_p, seed, p.salt3, (unsigned long long)b * C + lane) : 1.f;
      u3 = cg_gate_u(k3, ry[k], keep3);
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
      S1 += (double)g; S2 += (double)g * (double)((ry[k] - k3[0]) * k3[1]);
      rg[k] = g;
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
  // z of this thread's samples replaces dw; the statistics rows of the first round are requested
#pragma unroll
  for (int k = 0; k < CG_GATE_NS; ++k) {
    const int b = wave + CG_GATE_WAVES * k;
    ra[k] = (b < B && lane < C) ? p.z[(long long)b * C + lane] : 0.f;
  }
  float st0 = 0.f, st1 = 0.f, st2 = 0.f;
  auto load_stats = [&](int b) {
    const float* row = p.stats + (long long)b * p.stats_ld;
    st0 = (b < B && lane < S) ? row[lane] : 0.f;
    st1 = (b < B && lane + 64 < S) ? row[lane + 64] : 0.f;
    st2 = (b < B && lane + 128 < S) ? row[lane + 128] : 0.f;
  };
  load_stats(wave);
  S1 = cg_gate_colsum(S1, sRed, wave, lane);
  S2 = cg_gate_colsum(S2, sRed, wave, lane);
  SA = cg_gate_colsum(SA, sRed, wave, lane);
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
#pragma unroll
  for (int k = 0; k < CG_GATE_NS; ++k) {
    const int b0 = CG_GATE_WAVES * k, b = b0 + wave, nb = min(CG_GATE_WAVES, B - b0);
    if (b0 >= B) break;
    __syncthreads();
    float u2 = 0.f, keep2 = 1.f;
    if (b < B) {
      float* u = sH + wave * KP;
      if (lane < C) {
        const float xh = (ry[k] - k3[0]) * k3[1];
        sA[wave * CG_GATE_CP + lane] = k3[2] * (rg[k] - sM[1][lane][0] - xh * sM[1][lane][1]);
        keep2 = drop ? cg_drop_scale(t.drop_p, seed, p.salt2, (unsigned long long)b * C + lane) : 1.f;
        u2 = cg_gate_u(k2, ra[k], keep2);
        u[lane] = u2 > 0.f ? u2 : alpha2 * u2;
      }
      if (lane < S) u[C + lane] = st0;
      if (lane + 64 < S) u[C + lane + 64] = st1;
      if (lane + 128 < S) u[C + lane + 128] = st2;
    }
    load_stats(b + CG_GATE_WAVES);
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
          S1 += (double)g; S2 += (double)g * (double)((ra[k] - k2[0]) * k2[1]);
          rg[k] = g;
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
  if (lane < C) {
#pragma unroll
    for (int k = 0; k < CG_GATE_NS; ++k) {
      const int b = wave + CG_GATE_WAVES * k;
      if (b < B) {
        const float xh = (ra[k] - k2[0]) * k2[1];
        p.dz[(long long)b * C + lane] = k2[2] * (rg[k] - sM[0][lane][0] - xh * sM[0][lane][1]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < CG_GATE_ACC2; ++q) { const int e = tid + CG_GATE_THREADS * q; if (e < C * C) p.dW2[e] = acc2[q]; }
#pragma unroll
  for (int q = 0; q < CG_GATE_ACCL; ++q) { const int e = tid + CG_GATE_THREADS * q; if (e < C * K) p.dWl[e] = accl[q]; }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_gate_check(const CgGateHead* t, bool bwd) {
  if (!t || t->n < 1 || t->n > 2) return CG_EARG;
  if (t->B <= 0 || t->B > CG_GATE_WAVES * CG_GATE_NS || t->C <= 0 || t->C > CG_GATE_MAXC || t->S < 0 || t->S > CG_GATE_MAXS) return CG_ESHAPE;
  if (t->train && t->B < 2) return CG_ESHAPE;
  if (t->train && t->drop_p > 0.f && !t->seed) return CG_EARG;
  if (t->drop_p < 0.f || t->drop_p >= 1.f) return CG_EARG;
  for (int i = 0; i < t->n; ++i) {
    const CgGatePath& p = t->p[i];
    if (!p.z || (t->S > 0 && !p.stats) || !p.Wl || !p.W2 || !p.alpha2 || !p.alpha3 || !p.y) return CG_EARG;
    if (!p.bn2.gamma || !p.bn2.beta || !p.bn2.save || !p.bn3.gamma || !p.bn3.beta || !p.bn3.save) return CG_EARG;
    if (!bwd && (!p.w || (!t->train && (!p.bn2.running_mean || !p.bn2.running_var || !p.bn3.running_mean || !p.bn3.running_var)))) return CG_EARG;
    if (bwd && (!p.dw || !p.dz || (t->S > 0 && !p.dstats) || !p.dWl || !p.dW2 || !p.dgamma2 || !p.dbeta2 || !p.dalpha2 || !p.dgamma3 || !p.dbeta3 ||
                !p.dalpha3)) return CG_EARG;
  }
  return CG_OK;
}

extern "C" int cg_gate_head_supported(int B, int C, int S) {
  return (B > 0 && B <= CG_GATE_WAVES * CG_GATE_NS && C > 0 && C <= CG_GATE_MAXC && S >= 0 && S <= CG_GATE_MAXS) ? 1 : 0;
}
extern "C" long long cg_gate_head_scratch_floats(int B, int C, int S) { (void)B; (void)C; (void)S; return 0; }      // the backward keeps its intermediates in registers

// include/cistgcn_hip.h : cg_gate_head_fwd / cg_gate_head_bwd
extern "C" int cg_gate_head_fwd(const CgGateHead* t, void* stream_) {
  int st = cg_gate_check(t, false);
  if (st != CG_OK) return st;
  const int K = t->C + t->S, KP = (K + 3) & ~3;
  const size_t lds = ((size_t)KP * CG_GATE_CP + (size_t)t->C * CG_GATE_CP + (size_t)CG_GATE_WAVES * KP + 2 * CG_GATE_CP * 4 + 1) * sizeof(float) +
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
