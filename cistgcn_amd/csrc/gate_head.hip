// Tail of the two gate paths of a DSTD_GC block (reference: conv_s / conv_t slots 5-7 and map_s / map_t, CISTGCN.py:337-352, applied
// by :378-384):   z (B,C) -> BatchNorm2d -> Dropout -> PReLU -> cat with the block statistics (B, 2+2T) -> Linear -> BatchNorm1d ->
// Dropout -> PReLU -> Linear = the gate w1 / w2 (B,C).
//
// Every tensor here is (B, <= C + S) - a few hundred KB at B = 256 - but as separate operators the chain was seven launches forward
// (channel sums, row kernel, two concatenation copies, contraction, row kernel, contraction) and as many backward, each 10-45 us of
// launch latency and split-K bookkeeping: ~110 us forward and ~90 us backward per block, 1.1 ms of the 22 ms step at B = 256 and a
// quarter of the launches of the B = 16 step.  Here the chain is cut only at its batch statistics: two launches forward, three backward,
// for both paths.  A workgroup owns sixteen samples (a wave = a sample, a lane = a column; the weights sit transposed in LDS, the
// concatenation is an index) and computes the BatchNorm statistics of the WHOLE batch itself - the tensors are (B, 64): reading 64 KB
// again per workgroup costs less than a launch, and no workgroup waits for another.  (First version, measured: ONE workgroup per path
// holding the whole batch, one launch per direction - 112 / 278 us at B = 256, the two Linear layers bound by the LDS bandwidth of a
// single CU.)
#include "cg_common.h"
#include "cg_phase.h"
#include "gate_head.h"

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

#define CG_GATE_THREADS 1024
#define CG_GATE_WAVES (CG_GATE_THREADS / CG_WAVE)
#define CG_GATE_CP 64                      // padded column count = lanes of a wave

struct CgGateAff { float mean, rstd, scale, beta; };

// BatchNorm constants of column c from the workgroup's own batch sums (train) or the running statistics (eval); thread c of wave 0
// calls this (forward); the workgroup of the first chunk (`owner`) records save / running statistics exactly like nn.BatchNorm
__device__ __forceinline__ CgGateAff cg_gate_aff_fwd(const CgTailBN& bn, int c, int C, int B, int train, double s1, double s2, bool owner) {
  CgGateAff a;
  const float gamma = bn.gamma[c];
  a.beta = bn.beta[c];
  if (train) {
    const double cnt = (double)B, mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    a.mean = (float)mean;
    a.rstd = (float)(1.0 / sqrt(var + (double)bn.eps));
    if (owner && bn.running_mean) {
      const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      bn.running_mean[c] = (1.f - bn.momentum) * bn.running_mean[c] + bn.momentum * (float)mean;
      bn.running_var[c] = (1.f - bn.momentum) * bn.running_var[c] + bn.momentum * (float)unb;
      if (c == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
    }
  } else {
    a.mean = bn.running_mean[c];
    a.rstd = 1.0f / sqrtf(bn.running_var[c] + bn.eps);
  }
  if (owner) { bn.save[c] = a.mean; bn.save[C + c] = a.rstd; }
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

// sums over the whole batch of a (B,C) tensor per column, optionally of g and g * xhat with xhat = (x - mean) * rstd: every workgroup
// computes them for itself.  Thread (lane, wave) walks the samples wave, wave + 16, ...; result valid in wave 0
__device__ __forceinline__ void cg_gate_batch_sums(const float* __restrict__ a, const float* __restrict__ x, float mean, float rstd, int B, int C,
                                                   double* sRed, int wave, int lane, double& s1, double& s2) {
  s1 = 0.0; s2 = 0.0;
  if (lane < C)
    for (int b = wave; b < B; b += CG_GATE_WAVES) {
      const double v = (double)a[(long long)b * C + lane];
      s1 += v;
      s2 += x ? v * (double)((x[(long long)b * C + lane] - mean) * rstd) : v * v;
    }
  s1 = cg_gate_colsum(s1, sRed, wave, lane);
  s2 = cg_gate_colsum(s2, sRed, wave, lane);
}

// ======================================================================================================================
// forward, launch 1:  y = Wl [PReLU(Dropout(BN(z))) | stats]      grid (chunks of 16 samples, paths)
__global__ __launch_bounds__(CG_GATE_THREADS) void cg_gate_fwd1_kernel(CgGateHead t) {
  const CgGatePath& p = t.p[blockIdx.y];
  const int B = t.B, C = t.C, S = t.S, K = C + S, KP = (K + 3) & ~3;
  float* sWl = reinterpret_cast<float*>(cg_dyn_lds);                 // [KP][CP]  Wl transposed, rows K .. KP-1 zero
  float* sU = sWl + KP * CG_GATE_CP;                                  // [WAVES][KP] input rows of this workgroup's samples
  float* sAff = sU + CG_GATE_WAVES * KP;                              // [CP][4] mean, rstd, scale, beta
  double* sRed = reinterpret_cast<double*>(sAff + CG_GATE_CP * 4);    // an even number of floats in front of it
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const int b = (int)blockIdx.x * CG_GATE_WAVES + wave;
  const bool drop = t.train && t.drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  // this sample's row of z and of the statistics travel while the weights are staged and the batch is summed
  const float zv = (b < B && lane < C) ? p.z[(long long)b * C + lane] : 0.f;
  float st[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) st[q] = (b < B && lane + 64 * q < S) ? p.stats[(long long)b * p.stats_ld + lane + 64 * q] : 0.f;
  for (int e = tid; e < (KP - K) * CG_GATE_CP; e += CG_GATE_THREADS) sWl[K * CG_GATE_CP + e] = 0.f;
  for (int e = tid; e < C * K; e += CG_GATE_THREADS) { const int o = e / K, i = e - o * K; sWl[i * CG_GATE_CP + o] = p.Wl[e]; }
  double s1 = 0.0, s2 = 0.0;
  if (t.train) cg_gate_batch_sums(p.z, nullptr, 0.f, 0.f, B, C, sRed, wave, lane, s1, s2);
  if (wave == 0 && lane < C) {
    const CgGateAff a = cg_gate_aff_fwd(p.bn2, lane, C, B, t.train, s1, s2, blockIdx.x == 0);
    float* k = sAff + 4 * lane;
    k[0] = a.mean; k[1] = a.rstd; k[2] = a.scale; k[3] = a.beta;
  }
  __syncthreads();
  if (b < B) {
    float* u = sU + wave * KP;
    if (lane < C) {
      const float keep = drop ? cg_drop_scale(t.drop_p, seed, p.salt2, (unsigned long long)b * C + lane) : 1.f;
      const float uu = cg_gate_u(sAff + 4 * lane, zv, keep);
      const float v = uu > 0.f ? uu : p.alpha2[0] * uu;
      if (p.tap2) p.tap2[(long long)b * C + lane] = v;
      u[lane] = v;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) if (lane + 64 * q < S) u[C + lane + 64 * q] = st[q];
    for (int i = K + lane; i < KP; i += CG_WAVE) u[i] = 0.f;
  }
  __syncthreads();
  if (b < B && lane < C) {
    const float* u = sU + wave * KP;
    float acc = 0.f;
    for (int i = 0; i < KP; i += 4) {
      const float4 uv = *reinterpret_cast<const float4*>(u + i);        // the same address in every lane: a broadcast
      acc += sWl[i * CG_GATE_CP + lane] * uv.x;
      acc += sWl[(i + 1) * CG_GATE_CP + lane] * uv.y;
      acc += sWl[(i + 2) * CG_GATE_CP + lane] * uv.z;
      acc += sWl[(i + 3) * CG_GATE_CP + lane] * uv.w;
    }
    p.y[(long long)b * C + lane] = acc;
  }
}

// forward, launch 2:  w = W2 PReLU(Dropout(BN1d(y)))
__global__ __launch_bounds__(CG_GATE_THREADS) void cg_gate_fwd2_kernel(CgGateHead t) {
  const CgGatePath& p = t.p[blockIdx.y];
  const int B = t.B, C = t.C;
  float* sW2 = reinterpret_cast<float*>(cg_dyn_lds);                 // [C][CP]   W2 transposed
  float* sU = sW2 + C * CG_GATE_CP;                                   // [WAVES][CP]
  float* sAff = sU + CG_GATE_WAVES * CG_GATE_CP;
  double* sRed = reinterpret_cast<double*>(sAff + CG_GATE_CP * 4);
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const int b = (int)blockIdx.x * CG_GATE_WAVES + wave;
  const bool drop = t.train && t.drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  const float yv = (b < B && lane < C) ? p.y[(long long)b * C + lane] : 0.f;
  for (int e = tid; e < C * C; e += CG_GATE_THREADS) { const int o = e / C, j = e - o * C; sW2[j * CG_GATE_CP + o] = p.W2[e]; }
  double s1 = 0.0, s2 = 0.0;
  if (t.train) cg_gate_batch_sums(p.y, nullptr, 0.f, 0.f, B, C, sRed, wave, lane, s1, s2);
  if (wave == 0 && lane < C) {
    const CgGateAff a = cg_gate_aff_fwd(p.bn3, lane, C, B, t.train, s1, s2, blockIdx.x == 0);
    float* k = sAff + 4 * lane;
    k[0] = a.mean; k[1] = a.rstd; k[2] = a.scale; k[3] = a.beta;
  }
  __syncthreads();
  if (b < B && lane < C) {
    const float keep = drop ? cg_drop_scale(t.drop_p, seed, p.salt3, (unsigned long long)b * C + lane) : 1.f;
    const float uu = cg_gate_u(sAff + 4 * lane, yv, keep);
    const float v = uu > 0.f ? uu : p.alpha3[0] * uu;
    if (p.tap3) p.tap3[(long long)b * C + lane] = v;
    sU[wave * CG_GATE_CP + lane] = v;
  }
  __syncthreads();
  if (b < B && lane < C) {
    const float* h = sU + wave * CG_GATE_CP;
    float acc = 0.f;
    for (int j = 0; j < C; ++j) acc += sW2[j * CG_GATE_CP + lane] * h[j];
    p.w[(long long)b * C + lane] = acc;
  }
}

// ======================================================================================================================
// backward, launch 1: through the last Linear and the PReLU / Dropout behind bn3 -> G3 (gradient in front of bn3); dW2, slope sum
__global__ __launch_bounds__(CG_GATE_THREADS) void cg_gate_bwd1_kernel(CgGateHead t) {
  const CgGatePath& p = t.p[blockIdx.y];
  const int B = t.B, C = t.C;
  float* sW2 = reinterpret_cast<float*>(cg_dyn_lds);                 // [C][C] as it lies in memory
  float* sA = sW2 + C * C;                                            // [WAVES][CP] dw rows
  float* sH = sA + CG_GATE_WAVES * CG_GATE_CP;                        // [WAVES][CP] h3 rows
  double* sRed = reinterpret_cast<double*>(sH + CG_GATE_WAVES * CG_GATE_CP + ((C * C) & 1));
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const int b0 = (int)blockIdx.x * CG_GATE_WAVES, b = b0 + wave, nb = min(CG_GATE_WAVES, B - b0);
  const bool drop = t.train && t.drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  const bool ok = b < B && lane < C;
  const float dwv = ok ? p.dw[(long long)b * C + lane] : 0.f, yv = ok ? p.y[(long long)b * C + lane] : 0.f;
  for (int e = tid; e < C * C; e += CG_GATE_THREADS) sW2[e] = p.W2[e];
  float u3 = 0.f, keep3 = 1.f;
  const float alpha3 = p.alpha3[0];
  if (ok) {
    const CgGateAff a3 = cg_gate_aff_bwd(p.bn3, lane, C);
    const float k3[4] = {a3.mean, a3.rstd, a3.scale, a3.beta};
    keep3 = drop ? cg_drop_scale(t.drop_p, seed, p.salt3, (unsigned long long)b * C + lane) : 1.f;
    u3 = cg_gate_u(k3, yv, keep3);
    sA[wave * CG_GATE_CP + lane] = dwv;
    sH[wave * CG_GATE_CP + lane] = u3 > 0.f ? u3 : alpha3 * u3;
  }
  __syncthreads();
  double SA = 0.0;
  if (ok) {
    const float* dwr = sA + wave * CG_GATE_CP;
    float dh = 0.f;
    for (int o = 0; o < C; ++o) dh += dwr[o] * sW2[o * C + lane];
    const float gu = u3 > 0.f ? dh : alpha3 * dh;
    if (!(u3 > 0.f)) SA = (double)dh * (double)u3;
    p.scratch[(long long)b * C + lane] = gu * keep3;
  }
  for (int e = tid; e < C * C; e += CG_GATE_THREADS) {
    const int o = e / C, j = e - o * C;
    float a = 0.f;
    for (int w = 0; w < nb; ++w) a += sA[w * CG_GATE_CP + o] * sH[w * CG_GATE_CP + j];
    atomicAdd(&p.dW2[e], a);
  }
  SA = cg_gate_colsum(SA, sRed, wave, lane);
  if (wave == 0) {
    SA = cg_wave_sum(lane < C ? SA : 0.0);
    if (lane == 0) atomicAdd(&p.red[1], SA);
  }
}

// backward, launch 2: through bn3 and the first Linear -> the statistics' gradient, G2 (gradient in front of bn2); dWl, slope sum
__global__ __launch_bounds__(CG_GATE_THREADS) void cg_gate_bwd2_kernel(CgGateHead t) {
  const CgGatePath& p = t.p[blockIdx.y];
  const int B = t.B, C = t.C, S = t.S, K = C + S, KP = (K + 3) & ~3;
  float* sWl = reinterpret_cast<float*>(cg_dyn_lds);                 // [C][K] as it lies in memory
  float* sA = sWl + C * K;                                            // [WAVES][CP] dy rows
  float* sH = sA + CG_GATE_WAVES * CG_GATE_CP;                        // [WAVES][KP] u rows
  double* sRed = reinterpret_cast<double*>(sH + CG_GATE_WAVES * KP + ((C * K) & 1));
  __shared__ float sM[CG_GATE_CP][2];
  const float* G3 = p.scratch;
  float* G2 = p.scratch + (long long)B * C;
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const int b0 = (int)blockIdx.x * CG_GATE_WAVES, b = b0 + wave, nb = min(CG_GATE_WAVES, B - b0);
  const bool drop = t.train && t.drop_p > 0.f;
  const unsigned long long seed = drop ? *t.seed : 0ull;
  const bool ok = b < B && lane < C;
  const float gv = ok ? G3[(long long)b * C + lane] : 0.f, yv = ok ? p.y[(long long)b * C + lane] : 0.f, zv = ok ? p.z[(long long)b * C + lane] : 0.f;
  float st[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) st[q] = (b < B && lane + 64 * q < S) ? p.stats[(long long)b * p.stats_ld + lane + 64 * q] : 0.f;
  for (int e = tid; e < C * K; e += CG_GATE_THREADS) sWl[e] = p.Wl[e];
  CgGateAff a3 = {0.f, 0.f, 0.f, 0.f}, a2 = a3;
  if (lane < C) { a3 = cg_gate_aff_bwd(p.bn3, lane, C); a2 = cg_gate_aff_bwd(p.bn2, lane, C); }
  double S1, S2;
  cg_gate_batch_sums(G3, p.y, a3.mean, a3.rstd, B, C, sRed, wave, lane, S1, S2);
  if (wave == 0 && lane < C) {
    sM[lane][0] = t.train ? (float)(S1 / (double)B) : 0.f; sM[lane][1] = t.train ? (float)(S2 / (double)B) : 0.f;
    if (blockIdx.x == 0) { p.dgamma3[lane] = (float)S2; p.dbeta3[lane] = (float)S1; if (lane == 0) p.dalpha3[0] = (float)p.red[1]; }
  }
  __syncthreads();
  float u2 = 0.f, keep2 = 1.f;
  const float alpha2 = p.alpha2[0];
  if (b < B) {
    float* u = sH + wave * KP;
    if (lane < C) {
      sA[wave * CG_GATE_CP + lane] = a3.scale * (gv - sM[lane][0] - (yv - a3.mean) * a3.rstd * sM[lane][1]);
      const float k2[4] = {a2.mean, a2.rstd, a2.scale, a2.beta};
      keep2 = drop ? cg_drop_scale(t.drop_p, seed, p.salt2, (unsigned long long)b * C + lane) : 1.f;
      u2 = cg_gate_u(k2, zv, keep2);
      u[lane] = u2 > 0.f ? u2 : alpha2 * u2;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) if (lane + 64 * q < S) u[C + lane + 64 * q] = st[q];
  }
  __syncthreads();
  double SA = 0.0;
  if (b < B) {
    const float* dyr = sA + wave * CG_GATE_CP;
    for (int i = lane; i < K; i += CG_WAVE) {
      float du = 0.f;
      for (int o = 0; o < C; ++o) du += dyr[o] * sWl[o * K + i];
      if (i >= C) p.dstats[(long long)b * S + (i - C)] = du;
      else {                                                              // i == lane
        const float gu = u2 > 0.f ? du : alpha2 * du;
        if (!(u2 > 0.f)) SA = (double)du * (double)u2;
        G2[(long long)b * C + lane] = gu * keep2;
      }
    }
  }
  for (int e = tid; e < C * K; e += CG_GATE_THREADS) {
    const int o = e / K, i = e - o * K;
    float a = 0.f;
    for (int w = 0; w < nb; ++w) a += sA[w * CG_GATE_CP + o] * sH[w * KP + i];
    atomicAdd(&p.dWl[e], a);
  }
  SA = cg_gate_colsum(SA, sRed, wave, lane);
  if (wave == 0) {
    SA = cg_wave_sum(lane < C ? SA : 0.0);
    if (lane == 0) atomicAdd(&p.red[0], SA);
  }
}

// backward, launch 3: through bn2 -> dz
__global__ __launch_bounds__(CG_GATE_THREADS) void cg_gate_bwd3_kernel(CgGateHead t) {
  const CgGatePath& p = t.p[blockIdx.y];
  const int B = t.B, C = t.C;
  double* sRed = reinterpret_cast<double*>(cg_dyn_lds);
  const float* G2 = p.scratch + (long long)B * C;
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE;
  const int b = (int)blockIdx.x * CG_GATE_WAVES + wave;
  const bool ok = b < B && lane < C;
  const float gv = ok ? G2[(long long)b * C + lane] : 0.f, zv = ok ? p.z[(long long)b * C + lane] : 0.f;
  CgGateAff a2 = {0.f, 0.f, 0.f, 0.f};
  if (lane < C) a2 = cg_gate_aff_bwd(p.bn2, lane, C);
  double S1, S2;
  cg_gate_batch_sums(G2, p.z, a2.mean, a2.rstd, B, C, sRed, wave, lane, S1, S2);
  __shared__ float sM[CG_GATE_CP][2];
  if (wave == 0 && lane < C) {
    sM[lane][0] = t.train ? (float)(S1 / (double)B) : 0.f; sM[lane][1] = t.train ? (float)(S2 / (double)B) : 0.f;
    if (blockIdx.x == 0) { p.dgamma2[lane] = (float)S2; p.dbeta2[lane] = (float)S1; if (lane == 0) p.dalpha2[0] = (float)p.red[0]; }
  }
  __syncthreads();
  if (ok) p.dz[(long long)b * C + lane] = a2.scale * (gv - sM[lane][0] - (zv - a2.mean) * a2.rstd * sM[lane][1]);
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_gate_check(const CgGateHead* t, bool bwd) {
  if (!t || t->n < 1 || t->n > 2) return CG_EARG;
  if (t->B <= 0 || t->B > 65535 * CG_GATE_WAVES || t->C <= 0 || t->C > CG_GATE_MAXC || t->S < 0 || t->S > CG_GATE_MAXS) return CG_ESHAPE;
  if (t->train && t->B < 2) return CG_ESHAPE;
  if (t->train && t->drop_p > 0.f && !t->seed) return CG_EARG;
  if (t->drop_p < 0.f || t->drop_p >= 1.f) return CG_EARG;
  for (int i = 0; i < t->n; ++i) {
    const CgGatePath& p = t->p[i];
    if (!p.z || (t->S > 0 && !p.stats) || !p.Wl || !p.W2 || !p.alpha2 || !p.alpha3 || !p.y) return CG_EARG;
    if (!p.bn2.gamma || !p.bn2.beta || !p.bn2.save || !p.bn3.gamma || !p.bn3.beta || !p.bn3.save) return CG_EARG;
    if (!bwd && (!p.w || (!t->train && (!p.bn2.running_mean || !p.bn2.running_var || !p.bn3.running_mean || !p.bn3.running_var)))) return CG_EARG;
    if (bwd && (!p.dw || !p.dz || (t->S > 0 && !p.dstats) || !p.dWl || !p.dW2 || !p.dgamma2 || !p.dbeta2 || !p.dalpha2 || !p.dgamma3 || !p.dbeta3 ||
                !p.dalpha3 || !p.scratch || !p.red)) return CG_EARG;
  }
  return CG_OK;
}

extern "C" int cg_gate_head_supported(int B, int C, int S) { return (B > 0 && C > 0 && C <= CG_GATE_MAXC && S >= 0 && S <= CG_GATE_MAXS) ? 1 : 0; }
// backward scratch per path: the gradients in front of the two BatchNorms
extern "C" long long cg_gate_head_scratch_floats(int B, int C, int S) { (void)S; return (long long)2 * B * C; }

// include/cistgcn_hip.h : cg_gate_head_fwd (two launches) / cg_gate_head_bwd (three)
extern "C" int cg_gate_head_fwd(const CgGateHead* t, void* stream_) {
  int st = cg_gate_check(t, false);
  if (st != CG_OK) return st;
  hipStream_t stream = (hipStream_t)stream_;
  const int K = t->C + t->S, KP = (K + 3) & ~3;
  const dim3 grid((unsigned)((t->B + CG_GATE_WAVES - 1) / CG_GATE_WAVES), (unsigned)t->n), block(CG_GATE_THREADS);
  const size_t red = (size_t)CG_GATE_WAVES * CG_GATE_CP * sizeof(double);
  const size_t lds1 = ((size_t)KP * CG_GATE_CP + (size_t)CG_GATE_WAVES * KP + CG_GATE_CP * 4) * sizeof(float) + red;
  if (lds1 > 64 * 1024 && cg_lds_limit((const void*)cg_gate_fwd1_kernel, lds1) != hipSuccess) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_gate_fwd1_kernel, grid, block, lds1, stream, *t);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  const size_t lds2 = ((size_t)t->C * CG_GATE_CP + (size_t)CG_GATE_WAVES * CG_GATE_CP + CG_GATE_CP * 4) * sizeof(float) + red;
  hipLaunchKernelGGL(cg_gate_fwd2_kernel, grid, block, lds2, stream, *t);
  return cg_launch_status();
}

// dW2 / dWl (accumulated with float atomics over the chunks) and red[0..1] must be zero on entry
extern "C" int cg_gate_head_bwd(const CgGateHead* t, void* stream_) {
  int st = cg_gate_check(t, true);
  if (st != CG_OK) return st;
  hipStream_t stream = (hipStream_t)stream_;
  const int K = t->C + t->S, KP = (K + 3) & ~3;
  const dim3 grid((unsigned)((t->B + CG_GATE_WAVES - 1) / CG_GATE_WAVES), (unsigned)t->n), block(CG_GATE_THREADS);
  const size_t red = (size_t)CG_GATE_WAVES * CG_GATE_CP * sizeof(double);
  const size_t lds1 = ((size_t)t->C * t->C + (size_t)2 * CG_GATE_WAVES * CG_GATE_CP + 1) * sizeof(float) + red;
  hipLaunchKernelGGL(cg_gate_bwd1_kernel, grid, block, lds1, stream, *t);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  const size_t lds2 = ((size_t)t->C * K + (size_t)CG_GATE_WAVES * CG_GATE_CP + (size_t)CG_GATE_WAVES * KP + 1) * sizeof(float) + red;
  if (lds2 > 64 * 1024 && cg_lds_limit((const void*)cg_gate_bwd2_kernel, lds2) != hipSuccess) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_gate_bwd2_kernel, grid, block, lds2, stream, *t);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  hipLaunchKernelGGL(cg_gate_bwd3_kernel, grid, block, red, stream, *t);
  return cg_launch_status();
}
