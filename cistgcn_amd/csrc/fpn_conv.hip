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

// four consecutive positions p .. p + 3 (p % 4 == 0) of a row of P floats: one float4 when the rows are 16-byte aligned (P % 4 == 0),
// float2 pairs when P % 4 == 2 (25-joint skeletons: 10 x 25 planes; the last quad of a row is then half)
__device__ __forceinline__ void cg_fpn_store_quad(float* row, int p, int P, float a, float b, float c, float d) {
  if (p >= P) return;
  if ((P & 3) == 0) { *reinterpret_cast<float4*>(row + p) = make_float4(a, b, c, d); return; }
  *reinterpret_cast<float2*>(row + p) = make_float2(a, b);
  if (p + 2 < P) *reinterpret_cast<float2*>(row + p + 2) = make_float2(c, d);
}

#define CG_FPN_THREADS 512
#define CG_FPN_PAD 3

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
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
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
    // (round 4: three steps per iteration - vectors, then all gathers, then the MFMAs - changed nothing, 49.6 us either way: at 14 tasks of
    // 232 MFMAs per workgroup the k loop is ~6 us of a ~16 us workgroup; the rest is the prologue - 150 KB of LDS zeroed and filled, the
    // dilation's weights staged per SAMPLE.  Workgroups that keep their weights over a slice of samples, as the dW kernel does, are the fix.)
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
        cg_fpn_store_quad(yb + (long long)o * g.P, p, g.P, c[0] + bias, c[1] + bias, c[2] + bias, c[3] + bias);
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
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4, nw = CG_FPN_THREADS / 64;
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
    for (int u = 0; u < CG_FPN_DXT; u += 2) {                      // two tiles at a time: independent MFMA chains
      const int id0 = u * nw + wave, id1 = (u + 1) * nw + wave;
      if (id0 < PT * CT) {
        const bool two = id1 < PT * CT;
        const int pt0 = id0 / CT, ct0 = id0 - pt0 * CT, pt1 = two ? id1 / CT : pt0, ct1 = two ? id1 - pt1 * CT : ct0;
        const int pa = sPos[16 * pt0 + l15], pb = sPos[16 * pt1 + l15];
        const float* wp0 = sW + (16 * ct0 + l15) * g.KS2 + 4 * slot;
        const float* wp1 = sW + (16 * ct1 + l15) * g.KS2 + 4 * slot;
        cg_f32x4 c0 = acc[u], c1 = acc[u + 1];
#pragma unroll 3
        for (int k0 = 0; k0 < g.KP2; k0 += 16) {
          const float4 w0 = *reinterpret_cast<const float4*>(wp0 + k0), w1 = *reinterpret_cast<const float4*>(wp1 + k0);
          const int4 t4 = *reinterpret_cast<const int4*>(sTap + k0 + 4 * slot);
          const float a0[4] = {sD[pa + t4.x], sD[pa + t4.y], sD[pa + t4.z], sD[pa + t4.w]};
          const float a1[4] = {sD[pb + t4.x], sD[pb + t4.y], sD[pb + t4.z], sD[pb + t4.w]};
          const float v0[4] = {w0.x, w0.y, w0.z, w0.w}, v1[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s4], v0[s4], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s4], v1[s4], c1, 0, 0, 0);
          }
        }
        acc[u] = c0;
        if (two) acc[u + 1] = c1;
      }
    }
  }
  float* dxb = t.dx + (long long)b * t.C * g.P;
#pragma unroll
  for (int u = 0; u < CG_FPN_DXT; ++u) {
    const int id = u * nw + wave;
    if (id < PT * CT) {
      const int pt = id / CT, ct = id - pt * CT, c = 16 * ct + l15, p = 16 * pt + 4 * slot;
      if (c < t.C) cg_fpn_store_quad(dxb + (long long)c * g.P, p, g.P, acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
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
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4, nw = CG_FPN_THREADS / 64;
  const int OT = g.OM / 16, KT = g.KP / 16;
  cg_f32x4 acc[CG_FPN_DWT];
#pragma unroll
  for (int u = 0; u < CG_FPN_DWT; ++u) acc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f, bsum1 = 0.f;                  // bias gradient shares of outputs l15 and 16 + l15 (O <= 32)
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
    if ((g.P & 3) == 0) {
#pragma unroll 4
      for (int e = tid; e < t.O * (g.P / 4); e += CG_FPN_THREADS) {
        const int o = e / (g.P / 4), p = 4 * (e - o * (g.P / 4));
        *reinterpret_cast<float4*>(sD + o * DS + p) = *reinterpret_cast<const float4*>(dyb + (long long)o * g.P + p);
      }
    } else {                                                        // rows of 8-byte alignment: pairs
#pragma unroll 4
      for (int e = tid; e < t.O * (g.P / 2); e += CG_FPN_THREADS) {
        const int o = e / (g.P / 2), p = 2 * (e - o * (g.P / 2));
        *reinterpret_cast<float2*>(sD + o * DS + p) = *reinterpret_cast<const float2*>(dyb + (long long)o * g.P + p);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CG_FPN_DWT; u += 2) {                      // two tiles at a time: independent MFMA chains
      const int id0 = u * nw + wave, id1 = (u + 1) * nw + wave;
      if (id0 < OT * KT) {
        const bool two = id1 < OT * KT;
        const int ot0 = id0 / KT, kt0 = id0 - ot0 * KT, ot1 = two ? id1 / KT : ot0, kt1 = two ? id1 - ot1 * KT : kt0;
        const int tap0 = sTap[16 * kt0 + l15], tap1 = sTap[16 * kt1 + l15];
        const float* dp0 = sD + (16 * ot0 + l15) * DS + 4 * slot;
        const float* dp1 = sD + (16 * ot1 + l15) * DS + 4 * slot;
        cg_f32x4 c0 = acc[u], c1 = acc[u + 1];
        float bs0 = 0.f, bs1 = 0.f;
#pragma unroll 2
        for (int p0 = 0; p0 < g.PM; p0 += 16) {
          const float4 d0 = *reinterpret_cast<const float4*>(dp0 + p0), d1 = *reinterpret_cast<const float4*>(dp1 + p0);
          const int4 q4 = *reinterpret_cast<const int4*>(sPos + p0 + 4 * slot);
          // positions beyond P carry dy = 0 (their image offset is position 0's: any finite value)
          const float x0[4] = {sX[tap0 + q4.x], sX[tap0 + q4.y], sX[tap0 + q4.z], sX[tap0 + q4.w]};
          const float x1[4] = {sX[tap1 + q4.x], sX[tap1 + q4.y], sX[tap1 + q4.z], sX[tap1 + q4.w]};
          const float e0[4] = {d0.x, d0.y, d0.z, d0.w}, e1[4] = {d1.x, d1.y, d1.z, d1.w};
          bs0 += (d0.x + d0.y) + (d0.z + d0.w); bs1 += (d1.x + d1.y) + (d1.z + d1.w);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e0[s4], x0[s4], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e1[s4], x1[s4], c1, 0, 0, 0);
          }
        }
        acc[u] = c0;
        if (two) acc[u + 1] = c1;
        // the owner of tile (ot, 0) also holds, per lane, a quarter of sum_p dy[o = 16 ot + l15][p]: the bias gradient
        if (kt0 == 0) { if (ot0) bsum1 += bs0; else bsum += bs0; }
        if (two && kt1 == 0) { if (ot1) bsum1 += bs1; else bsum += bs1; }
      }
    }
  }
  // this workgroup's partial sums, plain stores: [workgroup][dilation][O * K + O]; cg_fpn_fold_kernel adds the workgroups up
  // (float atomics into shared replicas cost ~60 us here: 510 workgroups x 5650 words)
  float* ws = t.ws + ((long long)blockIdx.x * t.n + di) * (t.O * g.K + t.O);
#pragma unroll
  for (int u = 0; u < CG_FPN_DWT; ++u) {
    const int id = u * nw + wave;
    if (id < OT * KT) {
      const int ot = id / KT, kt = id - ot * KT, k = 16 * kt + l15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = 16 * ot + 4 * slot + q;
        if (o < t.O && k < g.K) ws[o * g.K + k] = acc[u][q];
      }
    }
  }
  // bias: four lanes (slots) of the owner wave hold quarters of an output's sum
  bsum += __shfl_xor(bsum, 16, 64); bsum += __shfl_xor(bsum, 32, 64);
  bsum1 += __shfl_xor(bsum1, 16, 64); bsum1 += __shfl_xor(bsum1, 32, 64);
  float* sB = reinterpret_cast<float*>(cg_dyn_lds);                // [2][nw][16] (the image is no longer needed)
  __syncthreads();
  if (slot == 0) { sB[wave * 16 + l15] = bsum; sB[(nw + wave) * 16 + l15] = bsum1; }
  __syncthreads();
  if (tid < 32 && tid < t.O) {
    float s0 = 0.f;
    for (int w = 0; w < nw; ++w) s0 += sB[((tid >> 4) * nw + w) * 16 + (tid & 15)];
    ws[t.O * g.K + tid] = s0;
  }
}

// dW / db of one dilation = sum of the workgroups' partials: 64 elements x 4 slices of the workgroup list per block
__global__ void cg_fpn_fold_kernel(CgFpnArgs a, int nwg) {
  __shared__ float red[4][64];
  const CgFpnConv& t = a.t; const CgFpnGeom& g = a.g;
  const int di = blockIdx.y, n = t.O * g.K + t.O, e = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float s = 0.f;
  if (e < n)
    for (int w = part; w < nwg; w += 4) s += t.ws[((long long)w * t.n + di) * n + e];
  red[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && e < n) {
    s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
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
  if ((g->P & 1) || g->PM > 256) return CG_ESHAPE;
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
// scratch of the backward (per-workgroup partial weight / bias gradients; no zeroing needed)
extern "C" long long cg_fpn_conv_ws_floats(int B, int C, int O, int n) { return (long long)(B < 170 ? B : 170) * n * (O * C * 9 + O); }

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
  hipError_t e = cg_lds_limit((const void*)cg_fpn_fwd_kernel, lds);
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
    hipError_t e = cg_lds_limit((const void*)cg_fpn_dx_kernel, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cg_fpn_dx_kernel, dim3((unsigned)t->B), dim3(CG_FPN_THREADS), lds, stream, a);
    st = cg_launch_status();
    if (st != CG_OK) return st;
  }
  const size_t lds = cg_fpn_lds_dw(t, a.g);
  if (lds > 160 * 1024) return CG_ESHAPE;
  hipError_t e = cg_lds_limit((const void*)cg_fpn_dw_kernel, lds);
  if (e != hipSuccess) return (int)e;
  const int nwg = (t->B + a.g.per - 1) / a.g.per;
  hipLaunchKernelGGL(cg_fpn_dw_kernel, dim3((unsigned)nwg, (unsigned)t->n), dim3(CG_FPN_THREADS), lds, stream, a);
  st = cg_launch_status();
  if (st != CG_OK) return st;
  const int nel = t->O * a.g.K + t->O;
  hipLaunchKernelGGL(cg_fpn_fold_kernel, dim3((unsigned)((nel + 63) / 64), (unsigned)t->n), dim3(256), 0, stream, a, nwg);
  return cg_launch_status();
}
