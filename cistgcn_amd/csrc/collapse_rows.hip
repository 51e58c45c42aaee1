// Collapsing convolution over the frame axis (reference: nn.Conv2d(C, O, (T, 1)) of the gate paths, CISTGCN.py:331-336, and of
// Map2Adj.time_compress, :138-150): y[b,o,v] = sum_{c,t} W[o,c,t] x[b,c,t,v].  Per sample this is Y_b (O x V) = W (O x K) X_b (K x V)
// with K = C*T and X_b = x[b] as it lies in memory; the tensors are 36-72 MB, the outputs a few hundred KB.  The generic
// contraction needed split-K with atomics and strided gathers for it (0.3 TB/s); here
//   forward   one 512-thread workgroup per sample: eight waves split K, feed the matrix cores straight from global memory
//             (W rows as float4 along k, x rows as 16-lane pieces along v), partial tiles are added in LDS
//   backward  a workgroup owns 64 rows k of W for a slice of the samples: dx[b,k,v] = sum_o W[o,k] dy[b,o,v] is written as one
//             contiguous 64 x V piece per sample, dW[o,k] += sum_v dy[b,o,v] x[b,k,v] stays in registers over the slice
#include "cg_common.h"
#include "cg_phase.h"
#include "collapse_rows.h"

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

#define CG_ROWS_FWD_THREADS 512
#define CG_ROWS_BWD_THREADS 256
#define CG_ROWS_KB 64            // rows of W per backward workgroup
#define CG_ROWS_REPLICAS 8

struct CgRowsGeom { int K, OT, slices, per, kranges; };
struct CgRowsArgs { CgRowsConv t; CgRowsGeom g; };

// ======================================================================================================================
// forward
// ======================================================================================================================
// OT: 16-row tiles of the O outputs (compile time: the accumulators are registers).  The operands come straight from global memory, a
// wave walks ~K / 128 groups of 16 rows k: with one group per iteration every iteration waited out a full memory latency in front of its
// 16-32 MFMAs (one workgroup per CU: nothing else to run meanwhile; 34 us for 36 MB).  Now four groups of a wave are in flight: the loads
// of group g + 4 are issued right behind the MFMAs of group g.  No load sits behind a branch: indices outside the tensor are clamped to
// a neighbouring element (rows o >= O and columns v >= V land in results that are never stored), only rows k >= K are zeroed.
template <int OT>
__global__ __launch_bounds__(CG_ROWS_FWD_THREADS) void cg_rows_fwd_kernel(CgRowsArgs a, unsigned magicT) {
  const CgRowsConv& t = a.t; const CgRowsGeom& g = a.g;
  const int b = blockIdx.x, K = g.K, V = t.V, O = t.O;
  constexpr int nw = CG_ROWS_FWD_THREADS / 64, DEPTH = 4;
  float* sY = reinterpret_cast<float*>(cg_dyn_lds);              // [nw][16 * OT][33] partial tiles of the eight waves (no LDS atomics: 1 000 cycles each)
  float* sTab = sY + nw * 16 * OT * 33;                           // [C][4] input transform: mean, gamma * rstd, beta, -
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  const bool tr = t.in_on != 0;
  const float in_alpha = tr ? t.in_alpha[0] : 1.f;
  float* const tapb = (tr && t.in_tap) ? t.in_tap + (long long)b * K * V : nullptr;
  if (tr) {
    for (int c = tid; c < t.C; c += CG_ROWS_FWD_THREADS) {
      const CgAff af = cg_tail_aff(t.in_bn, c, t.C, (double)t.B * t.T * V, t.in_train, false, blockIdx.x == 0);
      sTab[4 * c] = af.mean; sTab[4 * c + 1] = af.gamma * af.rstd; sTab[4 * c + 2] = af.beta; sTab[4 * c + 3] = 0.f;
    }
    __syncthreads();
  }
  const float* xb = t.x + (long long)b * K * V;
  cg_f32x4 acc[OT][2];
#pragma unroll
  for (int i = 0; i < OT; ++i) { acc[i][0] = cg_f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = acc[i][0]; }
  // groups of 16 consecutive k: lane (l15, slot) takes k = k0 + 4 slot + s in MFMA step s (both operands agree, a sum does not
  // care about the order)
  const int ngroups = (K + 15) >> 4;
  const int vA = min(l15, V - 1), vB = min(16 + l15, V - 1);
  const float* wrow[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) wrow[i] = t.W + (long long)min(16 * i + l15, O - 1) * K;
  struct Frag { float4 w[OT]; float x0[4], x1[4]; float mask; int kc; };      // mask: 0 for rows k >= K (applied in front of the MFMAs: arithmetic on a
                                                                                // loaded value right behind its load would wait for it there)
  auto load = [&](int gi, Frag& f) {
    const int k = 16 * gi + 4 * slot;
    const bool kin = k < K;                                 // K % 4 == 0: a lane's four rows are inside or outside together
    const int kc = kin ? k : K - 4;
    f.mask = kin ? 1.f : 0.f; f.kc = kc;
#pragma unroll
    for (int i = 0; i < OT; ++i) f.w[i] = *reinterpret_cast<const float4*>(wrow[i] + kc);
    const float* xr = xb + (long long)kc * V;
#pragma unroll
    for (int s = 0; s < 4; ++s) { f.x0[s] = xr[s * V + vA]; f.x1[s] = xr[s * V + vB]; }
  };
  auto mma = [&](Frag& f) {
    if (tr) {                                               // PReLU(BatchNorm(x)) of the four rows k = (c, t) of this lane, constants of channel c from LDS
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int c = (int)(magicT ? (unsigned)(((unsigned long long)(unsigned)(f.kc + s) * magicT) >> 32) : (unsigned)(f.kc + s));
        const float4 k4 = *reinterpret_cast<const float4*>(sTab + 4 * c);
        f.x0[s] = cg_prelu((f.x0[s] - k4.x) * k4.y + k4.z, in_alpha);
        f.x1[s] = cg_prelu((f.x1[s] - k4.x) * k4.y + k4.z, in_alpha);
        if (tapb && f.mask != 0.f) {                        // every element of x[b] is held by exactly one lane (clamped lanes repeat a neighbour's value)
          tapb[(long long)(f.kc + s) * V + vA] = f.x0[s];
          tapb[(long long)(f.kc + s) * V + vB] = f.x1[s];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      const float w4[4] = {f.w[i].x * f.mask, f.w[i].y * f.mask, f.w[i].z * f.mask, f.w[i].w * f.mask};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[s], f.x0[s], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[s], f.x1[s], acc[i][1], 0, 0, 0);
      }
    }
  };
  Frag f[DEPTH];
#pragma unroll
  for (int j = 0; j < DEPTH; ++j)
    if (wave + j * nw < ngroups) load(wave + j * nw, f[j]);
  for (int gi = wave; gi < ngroups; gi += DEPTH * nw) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int gj = gi + j * nw;
      if (gj < ngroups) {
        mma(f[j]);
        if (gj + DEPTH * nw < ngroups) load(gj + DEPTH * nw, f[j]);
      }
    }
  }
  float* mine = sY + wave * 16 * OT * 33;
#pragma unroll
  for (int i = 0; i < OT; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = 16 * i + 4 * slot + q;
      mine[o * 33 + l15] = acc[i][0][q];
      mine[o * 33 + 16 + l15] = acc[i][1][q];
    }
  }
  __syncthreads();
  for (int e = tid; e < O * V; e += CG_ROWS_FWD_THREADS) {
    const int o = e / V, v = e - o * V;
    float y = 0.f;
#pragma unroll
    for (int w = 0; w < nw; ++w) y += sY[(w * 16 * OT + o) * 33 + v];
    t.y[(long long)b * O * V + e] = y;
    sY[o * 33 + v] = y;                                    // this thread's own element of slot 0: the channel sums below read it
  }
  if (t.stats) {                                       // f64 channel sums for the BatchNorm behind the convolution
    __syncthreads();
    for (int o = tid; o < O; o += CG_ROWS_FWD_THREADS) {
      double s1 = 0.0, s2 = 0.0;
      for (int v = 0; v < V; ++v) { const double y = (double)sY[o * 33 + v]; s1 += y; s2 += y * y; }
      double* rep = t.stats + ((long long)(b % CG_STAT_REPLICAS) * O + o) * 2;
      atomicAdd(&rep[0], s1); atomicAdd(&rep[1], s2);
    }
  }
}

// ======================================================================================================================
// backward: workgroup = (64 rows k of W, slice of the samples)
// ======================================================================================================================
// (128 rows x 512 threads per workgroup: 57 us instead of 44.)
// (Round 4 tried this kernel with compile-time OT, dy of the next sample through registers and clamped instead of conditional loads:
// 90 us instead of 100 on the 64-output gate convolutions, but 152 us instead of 47 on the 32-output tower convolutions; not understood,
// not shipped.)
template <bool RD>
__global__ __launch_bounds__(CG_ROWS_BWD_THREADS) void cg_rows_bwd_kernel(CgRowsArgs a) {
  const CgRowsConv& t = a.t; const CgRowsGeom& g = a.g;
  const int K = g.K, V = t.V, O = t.O;
  const int kr = blockIdx.x, sl = blockIdx.y, k0 = kr * CG_ROWS_KB;
  const int b0 = sl * g.per, b1 = min(t.B, b0 + g.per);
  if (b0 >= t.B) return;
  float* sW = reinterpret_cast<float*>(cg_dyn_lds);              // [16 * OT][KB + 4]  W[o][k0 ..]
  float* sDY = sW + 16 * g.OT * (CG_ROWS_KB + 4);                 // [2][16 * OT][36]   dy of the current / next sample, v padded with zeros
  const int WS = CG_ROWS_KB + 4, DS = 36;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  // eight loads of a thread in flight (a load - store loop waits out one memory latency per element: 9 to 17 of them in this prologue)
  for (int e0 = tid; e0 < 16 * g.OT * WS; e0 += 8 * CG_ROWS_BWD_THREADS) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = e0 + CG_ROWS_BWD_THREADS * j, o = e / WS, kk = e - o * WS;
      const bool in = o < O && kk < CG_ROWS_KB && k0 + kk < K;
      v[j] = t.W[in ? (long long)o * K + k0 + kk : (long long)k0];       // W[0][k0] stands in for the padding (k0 < K); zeroed at the store
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = e0 + CG_ROWS_BWD_THREADS * j, o = e / WS, kk = e - o * WS;
      if (e < 16 * g.OT * WS) sW[e] = (o < O && kk < CG_ROWS_KB && k0 + kk < K) ? v[j] : 0.f;
    }
  }
  for (int e = tid; e < 2 * 16 * g.OT * DS; e += CG_ROWS_BWD_THREADS) sDY[e] = 0.f;
  __syncthreads();
  auto stage_dy = [&](int b, int buf) {
    const float* src = t.dy + (long long)b * O * V;
    float* dst = sDY + buf * 16 * g.OT * DS;
    for (int e = tid; e < O * V; e += CG_ROWS_BWD_THREADS) { const int o = e / V, v = e - o * V; dst[o * DS + v] = src[e]; }
  };
  stage_dy(b0, 0);
  // a wave owns the 16 rows k = k0 + 16 wave + .. of the range: dW tiles [o][k] in registers over the slice
  cg_f32x4 wacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wacc[i] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  const int kw = k0 + 16 * wave;                                 // first row of this wave
  const bool v0ok = l15 < V, v1ok = 16 + l15 < V;
  // this lane's share of x[b][kw + l15][:]: four consecutive v per slot (v = 4 slot + s) and the second half (16 + ..);
  // the next sample's values are requested before the current sample's matrix work
  const int krow = kw + l15;
  auto load_x = [&](int b, float xa_[4], float xc_[4]) {
    const float* xb = t.x + (long long)b * K * V;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int v = 4 * slot + s;
      xa_[s] = (krow < K && v < V) ? xb[(long long)krow * V + v] : 0.f;
      xc_[s] = (krow < K && 16 + v < V) ? xb[(long long)krow * V + 16 + v] : 0.f;
    }
  };
  // optional input transform (see CgRowsConv.in_on): this lane's row k = (c, t) has ONE channel c; the values outside the tensor meet dy = 0
  const bool tr = t.in_on != 0;
  float tm = 0.f, ts = 1.f, tb = 0.f, ta = 1.f;
  if (tr) {
    const CgAff af = cg_tail_aff(t.in_bn, min(krow, K - 1) / t.T, t.C, 0.0, t.in_train, true, false);
    tm = af.mean; ts = af.gamma * af.rstd; tb = af.beta; ta = t.in_alpha[0];
  }
  auto act = [&](float xv[4]) {
    if (tr) {
#pragma unroll
      for (int s = 0; s < 4; ++s) xv[s] = cg_prelu((xv[s] - tm) * ts + tb, ta);
    }
  };
  // optional reduction of the BatchNorm / PReLU backward (in_red): the raw x at the positions of this lane's dx results (rows k = kw + 4 slot + q,
  // columns l15 and 16 + l15), requested a sample ahead like the operands; per-row constants; f64 sums per row over the slice
  const bool rd = RD && tr && t.in_red != nullptr;
  float qm[4], qr[4], qs[4], qb[4];
  long long qoff[4];
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0}, sal = 0.0;
  const int vA = min(l15, V - 1), vB = min(16 + l15, V - 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = min(kw + 4 * slot + q, K - 1);
    qoff[q] = (long long)k * V;
    qm[q] = 0.f; qr[q] = 1.f; qs[q] = 1.f; qb[q] = 0.f;
    if (rd) {
      const CgAff af = cg_tail_aff(t.in_bn, k / t.T, t.C, 0.0, t.in_train, true, false);
      qm[q] = af.mean; qr[q] = af.rstd; qs[q] = af.gamma * af.rstd; qb[q] = af.beta;
    }
  }
  auto load_r = [&](int b, float r0[4], float r1[4]) {
    const float* xb = t.x + (long long)b * K * V;
#pragma unroll
    for (int q = 0; q < 4; ++q) { r0[q] = xb[qoff[q] + vA]; r1[q] = xb[qoff[q] + vB]; }
  };
  float ra0[4], ra1[4], rn0[4], rn1[4];
  if (rd) load_r(b0, ra0, ra1);
  float xa[4], xc[4], xna[4], xnc[4];
  load_x(b0, xa, xc);
  act(xa); act(xc);
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    __syncthreads();                                              // dy[b] is in sDY[buf]; the other buffer is free
    if (b + 1 < b1) { stage_dy(b + 1, buf ^ 1); load_x(b + 1, xna, xnc); if (rd) load_r(b + 1, rn0, rn1); }
    const float* dyb = sDY + buf * 16 * g.OT * DS;
    float* dxb = t.dx + (long long)b * K * V;
    // dx[k][v] = sum_o W[o][k] dy[o][v]: rows k of this wave, both halves of v
    cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
    for (int oc = 0; oc < 16 * g.OT; oc += 16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int o = oc + 4 * slot + s;
        const float w = sW[o * WS + 16 * wave + l15];              // A[i = k][kk = o]
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, dyb[o * DS + l15], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, dyb[o * DS + 16 + l15], c1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kw + 4 * slot + q;
      if (k < K) {
        if (v0ok) dxb[(long long)k * V + l15] = c0[q];
        if (v1ok) dxb[(long long)k * V + 16 + l15] = c1[q];
      }
    }
    if (rd) {
      // g = dx' PReLU'(u), u = BatchNorm(x): rows k >= K and columns v >= V carry dx' = 0 (zero columns of sW, zero padding of dy)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float xr = h ? ra1[q] : ra0[q], dh = h ? c1[q] : c0[q];
          const float d = xr - qm[q], u = d * qs[q] + qb[q];
          const float gh = u > 0.f ? dh : ta * dh;
          s1[q] += (double)gh; s2[q] += (double)gh * (double)(d * qr[q]);
          if (!(u > 0.f)) sal += (double)dh * (double)u;
        }
      }
    }
    // dW[o][k] += sum_v dy[o][v] x[k][v]: A[i = o][kk = v] from LDS (float4 along v), B[kk = v][j = k] = this lane's x values
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < g.OT) {
        const float4 d0 = *reinterpret_cast<const float4*>(dyb + (16 * i + l15) * DS + 4 * slot);
        const float4 d1 = *reinterpret_cast<const float4*>(dyb + (16 * i + l15) * DS + 16 + 4 * slot);
        const float e0[4] = {d0.x, d0.y, d0.z, d0.w}, e1[4] = {d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          wacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[s], xa[s], wacc[i], 0, 0, 0);
          wacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[s], xc[s], wacc[i], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) { xa[s] = xna[s]; xc[s] = xnc[s]; }
    if (b + 1 < b1) { act(xa); act(xc); }
    if (rd) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { ra0[q] = rn0[q]; ra1[q] = rn1[q]; }
    }
  }
  if (rd) {
    // the 16 lanes of a DPP row hold the same four rows k: their sums first, then one f64 word per (row, sum) into the workgroup's
    // per-channel words (a 64-row range spans 64 / T + 2 channels at most), then one global atomic per channel and workgroup
    __syncthreads();
    double* sR = reinterpret_cast<double*>(sDY);                  // dy images are dead: [CG_ROWS_KB + 1][2] f64 + [1]
    const int c_lo = min(k0, K - 1) / t.T, c_hi = min(k0 + CG_ROWS_KB - 1, K - 1) / t.T, nch = c_hi - c_lo + 1;
    for (int e = tid; e < 2 * nch + 1; e += CG_ROWS_BWD_THREADS) sR[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double a1 = cg_row16_sum(s1[q]), a2 = cg_row16_sum(s2[q]);
      if (l15 == 0) {
        const int c = min(kw + 4 * slot + q, K - 1) / t.T - c_lo;
        atomicAdd(&sR[2 * c], a1); atomicAdd(&sR[2 * c + 1], a2);
      }
    }
    sal = cg_wave_sum(sal);
    if (lane == 0) atomicAdd(&sR[2 * nch], sal);
    __syncthreads();
    for (int e = tid; e < 2 * nch; e += CG_ROWS_BWD_THREADS) atomicAdd(&t.in_red[2 * c_lo + e], sR[e]);
    if (tid == 0) atomicAdd(&t.in_red[2 * t.C + ((kr + 5 * sl) & (CG_ALPHA_SLOTS - 1))], sR[2 * nch]);
  }
  float* ws = t.ws + (long long)(sl % CG_ROWS_REPLICAS) * O * K;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < g.OT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = 16 * i + 4 * slot + q, k = kw + l15;
        if (o < O && k < K) atomicAdd(&ws[(long long)o * K + k], wacc[i][q]);
      }
    }
  }
}

__global__ void cg_rows_fold_kernel(CgRowsArgs a) {
  const CgRowsConv& t = a.t;
  const long long n = (long long)t.O * a.g.K;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < CG_ROWS_REPLICAS; ++r) s += t.ws[r * n + e];
    t.dW[e] = s;
  }
}

// ======================================================================================================================
// The same for the JOINT axis (reference: nn.Conv2d(C, O, (1, V)) of Map2Adj.joint_compress, CISTGCN.py:152-163):
//   y[b,o,t] = sum_{c,v} W[o,c,v] x[b,c,t,v]        per sample  Y_b (O x T) = W (O x K) X_b (K x T),  K = C * V
// x[b] lies in memory as [c][t][v]: row k = (c, v) of X_b is a column of stride V.  A lane's four consecutive k of an MFMA step are
// four consecutive v of one frame (a 16-byte run; a step that crosses from channel c to c + 1 splits), the 16 lanes of a row take 16
// frames.  Round 3 ran this through the generic contraction (split-K with atomics: 92 / 154 us per block next to the gate items).
//   forward   one 512-thread workgroup per sample, eight waves split K, operands straight from global memory, DEPTH groups in flight
//   backward  a workgroup owns 64 rows k for a slice of the samples (dx[b,k,t] from W^T dy, dW[o,k] in registers over the slice)
// T <= 64 (NT tiles of 16 frames), O <= 64, C * V % 4 == 0.
// ======================================================================================================================
__device__ __forceinline__ unsigned cg_cols_div(unsigned n, unsigned magic) { return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n; }

template <int OT, int NT>
__global__ __launch_bounds__(CG_ROWS_FWD_THREADS) void cg_cols_fwd_kernel(CgRowsArgs a, unsigned magicV) {
  const CgRowsConv& t = a.t; const CgRowsGeom& g = a.g;
  const int b = blockIdx.x, K = g.K, V = t.V, T = t.T, O = t.O, TV = T * V;
  constexpr int nw = CG_ROWS_FWD_THREADS / 64, DEPTH = 3, YS = 16 * NT + 1;
  float* sY = reinterpret_cast<float*>(cg_dyn_lds);              // [nw][16 * OT][YS] partial tiles of the eight waves
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  const float* xb = t.x + (long long)b * K * T;
  cg_f32x4 acc[OT][NT];
#pragma unroll
  for (int i = 0; i < OT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  const int ngroups = (K + 15) >> 4;
  const float* wrow[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) wrow[i] = t.W + (long long)min(16 * i + l15, O - 1) * K;
  int toff[NT];                                                   // frame of this lane in tile j (clamped: columns t >= T are never stored)
#pragma unroll
  for (int j = 0; j < NT; ++j) toff[j] = min(16 * j + l15, T - 1) * V;
  float* sTab = sY + nw * 16 * OT * YS;                           // [C][4] input transform: mean, gamma * rstd, beta, -
  const bool tr = t.in_on != 0;
  const float in_alpha = tr ? t.in_alpha[0] : 1.f;
  float* const tapb = (tr && t.in_tap) ? t.in_tap + (long long)b * K * T : nullptr;
  if (tr) {
    for (int c = tid; c < t.C; c += CG_ROWS_FWD_THREADS) {
      const CgAff af = cg_tail_aff(t.in_bn, c, t.C, (double)t.B * T * V, t.in_train, false, blockIdx.x == 0);
      sTab[4 * c] = af.mean; sTab[4 * c + 1] = af.gamma * af.rstd; sTab[4 * c + 2] = af.beta; sTab[4 * c + 3] = 0.f;
    }
    __syncthreads();
  }
  struct Frag { float4 w[OT]; float x[NT][4]; float mask; int kc; };
  auto load = [&](int gi, Frag& f) {
    const int k = 16 * gi + 4 * slot;
    const bool kin = k < K;                                       // K % 4 == 0
    const int kc = kin ? k : K - 4;
    f.mask = kin ? 1.f : 0.f; f.kc = kc;
#pragma unroll
    for (int i = 0; i < OT; ++i) f.w[i] = *reinterpret_cast<const float4*>(wrow[i] + kc);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int kk = kc + s, c = (int)cg_cols_div((unsigned)kk, magicV), v = kk - c * V;
      const float* xr = xb + c * TV + v;
#pragma unroll
      for (int j = 0; j < NT; ++j) f.x[j][s] = xr[toff[j]];
    }
  };
  auto mma = [&](Frag& f) {
    if (tr) {                                               // PReLU(BatchNorm(x)) of the rows k = (c, v) of this lane
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int c = (int)cg_cols_div((unsigned)(f.kc + s), magicV);
        const float4 k4 = *reinterpret_cast<const float4*>(sTab + 4 * c);
#pragma unroll
        for (int j = 0; j < NT; ++j) f.x[j][s] = cg_prelu((f.x[j][s] - k4.x) * k4.y + k4.z, in_alpha);
        if (tapb && f.mask != 0.f) {
          const int v = f.kc + s - c * V;
#pragma unroll
          for (int j = 0; j < NT; ++j) tapb[c * TV + toff[j] + v] = f.x[j][s];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      const float w4[4] = {f.w[i].x * f.mask, f.w[i].y * f.mask, f.w[i].z * f.mask, f.w[i].w * f.mask};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[s], f.x[j][s], acc[i][j], 0, 0, 0);
    }
  };
  Frag f[DEPTH];
#pragma unroll
  for (int j = 0; j < DEPTH; ++j)
    if (wave + j * nw < ngroups) load(wave + j * nw, f[j]);
  for (int gi = wave; gi < ngroups; gi += DEPTH * nw) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int gj = gi + j * nw;
      if (gj < ngroups) {
        mma(f[j]);
        if (gj + DEPTH * nw < ngroups) load(gj + DEPTH * nw, f[j]);
      }
    }
  }
  float* mine = sY + wave * 16 * OT * YS;
#pragma unroll
  for (int i = 0; i < OT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) mine[(16 * i + 4 * slot + q) * YS + 16 * j + l15] = acc[i][j][q];
  __syncthreads();
  for (int e = tid; e < O * T; e += CG_ROWS_FWD_THREADS) {
    const int o = e / T, tt = e - o * T;
    float y = 0.f;
#pragma unroll
    for (int w = 0; w < nw; ++w) y += sY[(w * 16 * OT + o) * YS + tt];
    t.y[(long long)b * O * T + e] = y;
    sY[o * YS + tt] = y;                                   // this thread's own element of slot 0: the channel sums below read it
  }
  if (t.stats) {
    __syncthreads();
    for (int o = tid; o < O; o += CG_ROWS_FWD_THREADS) {
      double s1 = 0.0, s2 = 0.0;
      for (int tt = 0; tt < T; ++tt) { const double y = (double)sY[o * YS + tt]; s1 += y; s2 += y * y; }
      double* rep = t.stats + ((long long)(b % CG_STAT_REPLICAS) * O + o) * 2;
      atomicAdd(&rep[0], s1); atomicAdd(&rep[1], s2);
    }
  }
}

template <int OT, int NT, bool RD>
__global__ __launch_bounds__(CG_ROWS_BWD_THREADS) void cg_cols_bwd_kernel(CgRowsArgs a, unsigned magicV) {
  const CgRowsConv& t = a.t; const CgRowsGeom& g = a.g;
  const int K = g.K, V = t.V, T = t.T, O = t.O, TV = T * V;
  const int kr = blockIdx.x, sl = blockIdx.y, k0 = kr * CG_ROWS_KB;
  const int b0 = sl * g.per, b1 = min(t.B, b0 + g.per);
  if (b0 >= t.B) return;
  constexpr int WS = CG_ROWS_KB + 4, DS = 16 * NT + 4;
  float* sW = reinterpret_cast<float*>(cg_dyn_lds);              // [16 * OT][KB + 4]  W[o][k0 ..]
  float* sDY = sW + 16 * OT * WS;                                 // [2][16 * OT][DS]   dy of the current / next sample, frames padded with zeros
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  for (int e0 = tid; e0 < 16 * OT * WS; e0 += 8 * CG_ROWS_BWD_THREADS) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = e0 + CG_ROWS_BWD_THREADS * j, o = e / WS, kk = e - o * WS;
      const bool in = o < O && kk < CG_ROWS_KB && k0 + kk < K;
      v[j] = t.W[in ? (long long)o * K + k0 + kk : (long long)k0];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = e0 + CG_ROWS_BWD_THREADS * j, o = e / WS, kk = e - o * WS;
      if (e < 16 * OT * WS) sW[e] = (o < O && kk < CG_ROWS_KB && k0 + kk < K) ? v[j] : 0.f;
    }
  }
  for (int e = tid; e < 2 * 16 * OT * DS; e += CG_ROWS_BWD_THREADS) sDY[e] = 0.f;
  __syncthreads();
  auto stage_dy = [&](int b, int buf) {
    const float* src = t.dy + (long long)b * O * T;
    float* dst = sDY + buf * 16 * OT * DS;
    for (int e = tid; e < O * T; e += CG_ROWS_BWD_THREADS) { const int o = e / T, tt = e - o * T; dst[o * DS + tt] = src[e]; }
  };
  stage_dy(b0, 0);
  // a wave owns the 16 rows k = k0 + 16 wave + .. of the range: dW tiles [o][k] in registers over the slice
  cg_f32x4 wacc[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) wacc[i] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  const int kw = k0 + 16 * wave;
  // row k = (c, v) of this lane as the B operand of the dW product (x[b][c][t][v], t = 16 tt + 4 slot + s), clamped: rows k >= K are never stored
  const int kl = min(kw + l15, K - 1), cl = (int)cg_cols_div((unsigned)kl, magicV);
  const int xoff = cl * TV + (kl - cl * V);
  int tx[NT][4];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt)
#pragma unroll
    for (int s = 0; s < 4; ++s) tx[tt][s] = min(16 * tt + 4 * slot + s, T - 1) * V;      // frames >= T meet dy = 0
  // rows k = kw + 4 slot + q of this lane's dx results
  int doff[4]; bool dok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = kw + 4 * slot + q, kc = min(k, K - 1), c = (int)cg_cols_div((unsigned)kc, magicV);
    doff[q] = c * TV + (kc - c * V); dok[q] = k < K;
  }
  auto load_x = [&](int b, float xv[NT][4]) {
    const float* xr = t.x + (long long)b * K * T + xoff;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int s = 0; s < 4; ++s) xv[tt][s] = xr[tx[tt][s]];
  };
  const bool tr = t.in_on != 0;                                   // optional input transform: this lane's row k = (c, v) has one channel
  float tm = 0.f, ts = 1.f, tb = 0.f, ta = 1.f;
  if (tr) {
    const CgAff af = cg_tail_aff(t.in_bn, cl, t.C, 0.0, t.in_train, true, false);
    tm = af.mean; ts = af.gamma * af.rstd; tb = af.beta; ta = t.in_alpha[0];
  }
  auto act = [&](float xv[NT][4]) {
    if (tr) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int s = 0; s < 4; ++s) xv[tt][s] = cg_prelu((xv[tt][s] - tm) * ts + tb, ta);
    }
  };
  // optional reduction of the BatchNorm / PReLU backward (in_red), as in cg_rows_bwd_kernel: raw x at the positions of this lane's dx results
  const bool rd = RD && tr && t.in_red != nullptr;
  float qm[4], qr[4], qs[4], qb[4];
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0}, sal = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    qm[q] = 0.f; qr[q] = 1.f; qs[q] = 1.f; qb[q] = 0.f;
    if (rd) {
      const int kc = min(kw + 4 * slot + q, K - 1);
      const CgAff af = cg_tail_aff(t.in_bn, (int)cg_cols_div((unsigned)kc, magicV), t.C, 0.0, t.in_train, true, false);
      qm[q] = af.mean; qr[q] = af.rstd; qs[q] = af.gamma * af.rstd; qb[q] = af.beta;
    }
  }
  int tcol[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) tcol[j] = min(16 * j + l15, T - 1) * V;
  auto load_r = [&](int b, float r[NT][4]) {
    const float* xb = t.x + (long long)b * K * T;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) r[j][q] = xb[doff[q] + tcol[j]];
  };
  float ra[NT][4], rn[NT][4];
  if (rd) load_r(b0, ra);
  float xa[NT][4], xn[NT][4];
  load_x(b0, xa);
  act(xa);
  for (int b = b0; b < b1; ++b) {
    const int buf = (b - b0) & 1;
    __syncthreads();                                              // dy[b] is in sDY[buf]; the other buffer is free
    if (b + 1 < b1) { stage_dy(b + 1, buf ^ 1); load_x(b + 1, xn); if (rd) load_r(b + 1, rn); }
    const float* dyb = sDY + buf * 16 * OT * DS;
    float* dxb = t.dx + (long long)b * K * T;
    // dx[k][t] = sum_o W[o][k] dy[o][t]: rows k of this wave, NT tiles of frames
    cg_f32x4 c[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) c[j] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int oc = 0; oc < 16 * OT; oc += 16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int o = oc + 4 * slot + s;
        const float w = sW[o * WS + 16 * wave + l15];              // A[i = k][kk = o]
#pragma unroll
        for (int j = 0; j < NT; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, dyb[o * DS + 16 * j + l15], c[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int tt = 16 * j + l15;
      if (tt < T) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (dok[q]) dxb[doff[q] + tt * V] = c[j][q];
      }
    }
    if (rd) {                                                     // rows k >= K and frames t >= T carry dx' = 0
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float dh = c[j][q], d = ra[j][q] - qm[q], u = d * qs[q] + qb[q];
          const float gh = u > 0.f ? dh : ta * dh;
          s1[q] += (double)gh; s2[q] += (double)gh * (double)(d * qr[q]);
          if (!(u > 0.f)) sal += (double)dh * (double)u;
        }
    }
    // dW[o][k] += sum_t dy[o][t] x[k][t]: A[i = o][kk = t] from LDS (float4 along t), B[kk = t][j = k] = this lane's x values
#pragma unroll
    for (int i = 0; i < OT; ++i) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const float4 d4 = *reinterpret_cast<const float4*>(dyb + (16 * i + l15) * DS + 16 * tt + 4 * slot);
        const float e4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) wacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(e4[s], xa[tt][s], wacc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int s = 0; s < 4; ++s) xa[tt][s] = xn[tt][s];
    if (b + 1 < b1) act(xa);
    if (rd) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) ra[j][q] = rn[j][q];
    }
  }
  if (rd) {
    __syncthreads();
    double* sR = reinterpret_cast<double*>(sDY);                  // dy images are dead: [channels of the range][2] f64 + [1]
    const int c_lo = (int)cg_cols_div((unsigned)min(k0, K - 1), magicV), c_hi = (int)cg_cols_div((unsigned)min(k0 + CG_ROWS_KB - 1, K - 1), magicV), nch = c_hi - c_lo + 1;
    for (int e = tid; e < 2 * nch + 1; e += CG_ROWS_BWD_THREADS) sR[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double a1 = cg_row16_sum(s1[q]), a2 = cg_row16_sum(s2[q]);
      if (l15 == 0) {
        const int cc = (int)cg_cols_div((unsigned)min(kw + 4 * slot + q, K - 1), magicV) - c_lo;
        atomicAdd(&sR[2 * cc], a1); atomicAdd(&sR[2 * cc + 1], a2);
      }
    }
    sal = cg_wave_sum(sal);
    if (lane == 0) atomicAdd(&sR[2 * nch], sal);
    __syncthreads();
    for (int e = tid; e < 2 * nch; e += CG_ROWS_BWD_THREADS) atomicAdd(&t.in_red[2 * c_lo + e], sR[e]);
    if (tid == 0) atomicAdd(&t.in_red[2 * t.C + ((kr + 5 * sl) & (CG_ALPHA_SLOTS - 1))], sR[2 * nch]);
  }
  float* ws = t.ws + (long long)(sl % CG_ROWS_REPLICAS) * O * K;
#pragma unroll
  for (int i = 0; i < OT; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = 16 * i + 4 * slot + q, k = kw + l15;
      if (o < O && k < K) atomicAdd(&ws[(long long)o * K + k], wacc[i][q]);
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_rows_geometry(const CgRowsConv* t, CgRowsGeom* g) {
  if (!t || !t->x || !t->W) return CG_EARG;
  if (t->B <= 0 || t->C <= 0 || t->T <= 0 || t->V <= 0 || t->V > 32 || t->O <= 0 || t->O > 64) return CG_ESHAPE;
  const long long K = (long long)t->C * t->T;
  if (K > (1 << 20) || (K & 3)) return CG_ESHAPE;
  g->K = (int)K; g->OT = (t->O + 15) / 16;
  g->kranges = (g->K + CG_ROWS_KB - 1) / CG_ROWS_KB;
  int slices = 768 / g->kranges;                        // ~768 backward workgroups (1024-1536: 53 us against 46 us on the tower tensors, round 4)
  slices = slices < 1 ? 1 : (slices > t->B ? t->B : slices);
  g->per = (t->B + slices - 1) / slices;
  g->slices = (t->B + g->per - 1) / g->per;
  return CG_OK;
}

extern "C" long long cg_collapse_rows_ws_floats(int C, int T, int O) { return (long long)CG_ROWS_REPLICAS * O * C * T; }

// include/cistgcn_hip.h : cg_collapse_rows_fwd / cg_collapse_rows_bwd
extern "C" int cg_collapse_rows_fwd(const CgRowsConv* t, void* stream_) {
  CgRowsArgs a;
  int st = cg_rows_geometry(t, &a.g);
  if (st != CG_OK) return st;
  if (!t->y) return CG_EARG;
  a.t = *t;
  if (t->in_on && (!t->in_alpha || !t->in_bn.gamma || !t->in_bn.beta || !t->in_bn.save || (t->in_train ? !t->in_bn.stats : !t->in_bn.running_mean))) return CG_EARG;
  const size_t lds = ((size_t)(CG_ROWS_FWD_THREADS / 64) * 16 * a.g.OT * 33 + (size_t)4 * t->C) * sizeof(float);
  const dim3 grid((unsigned)t->B), block(CG_ROWS_FWD_THREADS);
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned magicT = t->T > 1 ? (unsigned)((0x100000000ULL + t->T - 1) / t->T) : 0u;
#define CG_ROWS_FWD_LAUNCH(N)                                                                   \
  {                                                                                            \
    hipError_t e = cg_lds_limit((const void*)cg_rows_fwd_kernel<N>, lds);                       \
    if (e != hipSuccess) return (int)e;                                                        \
    hipLaunchKernelGGL((cg_rows_fwd_kernel<N>), grid, block, lds, stream, a, magicT);            \
  }
  switch (a.g.OT) {
    case 1: CG_ROWS_FWD_LAUNCH(1) break;
    case 2: CG_ROWS_FWD_LAUNCH(2) break;
    case 3: CG_ROWS_FWD_LAUNCH(3) break;
    default: CG_ROWS_FWD_LAUNCH(4) break;
  }
#undef CG_ROWS_FWD_LAUNCH
  return cg_launch_status();
}

extern "C" int cg_collapse_rows_bwd(const CgRowsConv* t, void* stream_) {
  CgRowsArgs a;
  int st = cg_rows_geometry(t, &a.g);
  if (st != CG_OK) return st;
  if (!t->dy || !t->dx || !t->dW || !t->ws) return CG_EARG;
  if (t->in_on && (!t->in_alpha || !t->in_bn.gamma || !t->in_bn.beta || !t->in_bn.save)) return CG_EARG;
  a.t = *t;
  const size_t lds = ((size_t)16 * a.g.OT * (CG_ROWS_KB + 4) + (size_t)2 * 16 * a.g.OT * 36) * sizeof(float);
  hipStream_t stream = (hipStream_t)stream_;
  if (t->in_on && t->in_red) hipLaunchKernelGGL((cg_rows_bwd_kernel<true>), dim3((unsigned)a.g.kranges, (unsigned)a.g.slices), dim3(CG_ROWS_BWD_THREADS), lds, stream, a);
  else hipLaunchKernelGGL((cg_rows_bwd_kernel<false>), dim3((unsigned)a.g.kranges, (unsigned)a.g.slices), dim3(CG_ROWS_BWD_THREADS), lds, stream, a);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  hipLaunchKernelGGL(cg_rows_fold_kernel, dim3(128), dim3(256), 0, stream, a);
  return cg_launch_status();
}

// ---- joint axis ----
static unsigned cg_cols_magic(int d) { return d > 1 ? (unsigned)((0x100000000ULL + d - 1) / d) : 0u; }
static int cg_cols_geometry(const CgRowsConv* t, CgRowsGeom* g) {
  if (!t || !t->x || !t->W) return CG_EARG;
  if (t->B <= 0 || t->C <= 0 || t->T <= 0 || t->V <= 0 || t->T > 64 || t->O <= 0 || t->O > 64) return CG_ESHAPE;
  const long long K = (long long)t->C * t->V;
  if (K > (1 << 20) || (K & 3) || K * t->T >= (1ll << 31)) return CG_ESHAPE;
  g->K = (int)K; g->OT = (t->O + 15) / 16;
  g->kranges = (g->K + CG_ROWS_KB - 1) / CG_ROWS_KB;
  int slices = 768 / g->kranges;
  slices = slices < 1 ? 1 : (slices > t->B ? t->B : slices);
  g->per = (t->B + slices - 1) / slices;
  g->slices = (t->B + g->per - 1) / g->per;
  return CG_OK;
}
static int cg_cols_nt(int T) { return T <= 16 ? 1 : T <= 32 ? 2 : 4; }

extern "C" int cg_collapse_cols_supported(int C, int T, int V, int O) { return T >= 1 && T <= 64 && O >= 1 && O <= 64 && (((long long)C * V) & 3) == 0 ? 1 : 0; }
extern "C" long long cg_collapse_cols_ws_floats(int C, int V, int O) { return (long long)CG_ROWS_REPLICAS * O * C * V; }

#define CG_COLS_DISPATCH(OT_, NT_, LAUNCH)                         \
  switch (4 * ((OT_) - 1) + ((NT_) == 1 ? 0 : (NT_) == 2 ? 1 : 2)) { \
    case 0: LAUNCH(1, 1) break; case 1: LAUNCH(1, 2) break; case 2: LAUNCH(1, 4) break;       \
    case 4: LAUNCH(2, 1) break; case 5: LAUNCH(2, 2) break; case 6: LAUNCH(2, 4) break;       \
    case 8: LAUNCH(3, 1) break; case 9: LAUNCH(3, 2) break; case 10: LAUNCH(3, 4) break;      \
    case 12: LAUNCH(4, 1) break; case 13: LAUNCH(4, 2) break; default: LAUNCH(4, 4) break;    \
  }

// include/cistgcn_hip.h : cg_collapse_cols_fwd / cg_collapse_cols_bwd
extern "C" int cg_collapse_cols_fwd(const CgRowsConv* t, void* stream_) {
  CgRowsArgs a;
  int st = cg_cols_geometry(t, &a.g);
  if (st != CG_OK) return st;
  if (!t->y) return CG_EARG;
  a.t = *t;
  const int nt = cg_cols_nt(t->T);
  if (t->in_on && (!t->in_alpha || !t->in_bn.gamma || !t->in_bn.beta || !t->in_bn.save || (t->in_train ? !t->in_bn.stats : !t->in_bn.running_mean))) return CG_EARG;
  const size_t lds = ((size_t)(CG_ROWS_FWD_THREADS / 64) * 16 * a.g.OT * (16 * nt + 1) + (size_t)4 * t->C + 4) * sizeof(float);
  const dim3 grid((unsigned)t->B), block(CG_ROWS_FWD_THREADS);
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned magic = cg_cols_magic(t->V);
#define CG_COLS_FWD_LAUNCH(N, M)                                                                \
  {                                                                                            \
    hipError_t e = cg_lds_limit((const void*)cg_cols_fwd_kernel<N, M>, lds);                    \
    if (e != hipSuccess) return (int)e;                                                        \
    hipLaunchKernelGGL((cg_cols_fwd_kernel<N, M>), grid, block, lds, stream, a, magic);          \
  }
  CG_COLS_DISPATCH(a.g.OT, nt, CG_COLS_FWD_LAUNCH)
#undef CG_COLS_FWD_LAUNCH
  return cg_launch_status();
}

extern "C" int cg_collapse_cols_bwd(const CgRowsConv* t, void* stream_) {
  CgRowsArgs a;
  int st = cg_cols_geometry(t, &a.g);
  if (st != CG_OK) return st;
  if (!t->dy || !t->dx || !t->dW || !t->ws) return CG_EARG;
  if (t->in_on && (!t->in_alpha || !t->in_bn.gamma || !t->in_bn.beta || !t->in_bn.save)) return CG_EARG;
  a.t = *t;
  const int nt = cg_cols_nt(t->T);
  const size_t lds = ((size_t)16 * a.g.OT * (CG_ROWS_KB + 4) + (size_t)2 * 16 * a.g.OT * (16 * nt + 4)) * sizeof(float);
  const dim3 grid((unsigned)a.g.kranges, (unsigned)a.g.slices), block(CG_ROWS_BWD_THREADS);
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned magic = cg_cols_magic(t->V);
#define CG_COLS_BWD_LAUNCH(N, M)                                                                \
  {                                                                                            \
    if (t->in_on && t->in_red) {                                                               \
      hipError_t e = cg_lds_limit((const void*)cg_cols_bwd_kernel<N, M, true>, lds);             \
      if (e != hipSuccess) return (int)e;                                                      \
      hipLaunchKernelGGL((cg_cols_bwd_kernel<N, M, true>), grid, block, lds, stream, a, magic);  \
    } else {                                                                                   \
      hipError_t e = cg_lds_limit((const void*)cg_cols_bwd_kernel<N, M, false>, lds);            \
      if (e != hipSuccess) return (int)e;                                                      \
      hipLaunchKernelGGL((cg_cols_bwd_kernel<N, M, false>), grid, block, lds, stream, a, magic); \
    }                                                                                          \
  }
  CG_COLS_DISPATCH(a.g.OT, nt, CG_COLS_BWD_LAUNCH)
#undef CG_COLS_BWD_LAUNCH
  st = cg_launch_status();
  if (st != CG_OK) return st;
  hipLaunchKernelGGL(cg_rows_fold_kernel, dim3(128), dim3(256), 0, stream, a);
  return cg_launch_status();
}
