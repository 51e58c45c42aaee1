// Stage kernels of the CIST-GCN path that are not plain contractions or per-channel affines:
// feature lift (row A), block statistics (row C), squeeze-excite gate (row H), cumulative sum
// (row J), tail sum (row L) and the MPJPE loss (row L).  Citations are to the reference file
// human_motion_prediction/models/CISTGCN/CISTGCN.py unless stated otherwise.
#include "cg_common.h"

// ---------------------------------------------------------------------------------------------
// Row A — CISTGCN.py:568-577.  x (B,T,V,3) -> f (B,10,T,V), channels [x(3), acc(3), vel(3), |vel|]
//   vel[t] = x[t+1]-x[t] (t<T-1), vel[T-1] = x[T-1];  acc[t] = vel[t+1]-vel[t] (t<T-1), acc[T-1] = vel[T-1]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float cg_vel(const float* x, long long T, long long V, long long b, long long t, long long v, int d) {
  const float cur = x[((b * T + t) * V + v) * 3 + d];
  if (t == T - 1) return cur;
  return x[((b * T + t + 1) * V + v) * 3 + d] - cur;
}

__global__ void cg_feature_lift_fwd_kernel(const float* __restrict__ x, float* __restrict__ f, long long B, long long T, long long V) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T * V) return;
  const long long v = i % V, t = (i / V) % T, b = i / (V * T);
  const long long TV = T * V;
  float* o = f + b * 10 * TV + t * V + v;
  float n2 = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float ve = cg_vel(x, T, V, b, t, v, d);
    const float ac = (t == T - 1) ? ve : cg_vel(x, T, V, b, t + 1, v, d) - ve;
    o[(0 + d) * TV] = x[((b * T + t) * V + v) * 3 + d];
    o[(3 + d) * TV] = ac;
    o[(6 + d) * TV] = ve;
    n2 += ve * ve;
  }
  o[9 * TV] = sqrtf(n2);
}

extern "C" int cg_feature_lift_fwd(const float* x, float* f, long long B, long long T, long long V, void* stream_) {
  if (!x || !f) return CG_EARG;
  if (B <= 0 || T <= 0 || V <= 0) return CG_ESHAPE;
  const long long n = B * T * V;
  hipLaunchKernelGGL(cg_feature_lift_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, f, B, T, V);
  return cg_launch_status();
}

// total gradient reaching vel[t] (directly, through |vel| and through acc)
__device__ __forceinline__ float cg_dvel(const float* x, const float* df, long long T, long long V, long long TV,
                                         long long b, long long t, long long v, int d) {
  const float* g = df + b * 10 * TV + t * V + v;
  float r = g[(6 + d) * TV];
  float ve[3], n2 = 0.f;
#pragma unroll
  for (int e = 0; e < 3; ++e) { ve[e] = cg_vel(x, T, V, b, t, v, e); n2 += ve[e] * ve[e]; }
  if (n2 > 0.f) r += g[9 * TV] * ve[d] / sqrtf(n2);       // d|vel|/dvel, sub-gradient 0 at |vel| = 0
  if (t < T - 1) r -= g[(3 + d) * TV]; else r += g[(3 + d) * TV];   // acc[t]
  if (t >= 1) r += g[(3 + d) * TV - V];                    // acc[t-1] = vel[t] - vel[t-1]
  return r;
}

__global__ void cg_feature_lift_bwd_kernel(const float* __restrict__ x, const float* __restrict__ df, float* __restrict__ dx,
                                           long long B, long long T, long long V) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T * V) return;
  const long long v = i % V, t = (i / V) % T, b = i / (V * T);
  const long long TV = T * V;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float r = df[b * 10 * TV + d * TV + t * V + v];
    const float dv = cg_dvel(x, df, T, V, TV, b, t, v, d);
    if (t < T - 1) r -= dv; else r += dv;
    if (t >= 1) r += cg_dvel(x, df, T, V, TV, b, t - 1, v, d);
    dx[((b * T + t) * V + v) * 3 + d] = r;
  }
}

extern "C" int cg_feature_lift_bwd(const float* x, const float* df, float* dx, long long B, long long T, long long V, void* stream_) {
  if (!x || !df || !dx) return CG_EARG;
  if (B <= 0 || T <= 0 || V <= 0) return CG_ESHAPE;
  const long long n = B * T * V;
  hipLaunchKernelGGL(cg_feature_lift_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, df, dx, B, T, V);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Row C — DSTD_GC._get_stats_, CISTGCN.py:360-371, on a contiguous (B,C,T,V) tensor.
//   out[b] = [ mean_c mean_{t,v} | mean_c mean_v (T) | std_c std_{t,v} | std_c std_v (T) ], all std unbiased.
// One workgroup per sample.  LDS: per-(c,t) row mean and centred sum of squares, per-c mean and std.
// Dynamic LDS layout (floats): rm[C*T] | rq[C*T] | cm[C] | cs[C]
// ---------------------------------------------------------------------------------------------
HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

// lanes that share a row of V values: 16, 32 or 64 (a row wider than 64 is walked in steps of 64)
__device__ __forceinline__ int cg_stats_lpr(int V) { return V <= 16 ? 16 : V <= 32 ? 32 : 64; }
// sum over the `lpr` lanes of a row group, in every lane of the group
__device__ __forceinline__ float cg_stats_group_sum(float v, int lpr) {
  v = cg_row16_sum(v);
  if (lpr >= 32) v += __shfl_xor(v, 16, 64);
  if (lpr >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// per-(c,t) row mean and centred sum of squares (one coalesced read of the sample: a group of lanes per row, the row's
// values stay in registers between the two sums), then per-c mean and std
__device__ __forceinline__ void cg_stats_rows(const float* xb, int C, int T, int V, float* rm, float* rq, float* cm, float* cs) {
  const int lpr = cg_stats_lpr(V), gl = threadIdx.x & (lpr - 1), grp = threadIdx.x / lpr, ngrp = blockDim.x / lpr;
  const int rows = C * T, iters = (rows + 4 * ngrp - 1) / (4 * ngrp);
  for (int it = 0; it < iters; ++it) {                 // every lane runs every iteration: the group sums are wave-wide exchanges
    float x0[4], s[4], m[4], q[4];
    int rr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                      // four rows per group in flight
      rr[k] = (4 * it + k) * ngrp + grp;
      const bool ok = rr[k] < rows && gl < V;
      x0[k] = ok ? xb[(long long)rr[k] * V + gl] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s[k] = x0[k];
      if (V > 64 && rr[k] < rows) for (int v = gl + 64; v < V; v += 64) s[k] += xb[(long long)rr[k] * V + v];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = cg_stats_group_sum(s[k], lpr) / (float)V;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q[k] = (rr[k] < rows && gl < V) ? (x0[k] - m[k]) * (x0[k] - m[k]) : 0.f;
      if (V > 64 && rr[k] < rows) for (int v = gl + 64; v < V; v += 64) { const float d = xb[(long long)rr[k] * V + v] - m[k]; q[k] += d * d; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q[k] = cg_stats_group_sum(q[k], lpr);
      if (rr[k] < rows && gl == 0) { rm[rr[k]] = m[k]; rq[rr[k]] = q[k]; }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float m = 0.f;
    for (int t = 0; t < T; ++t) m += rm[c * T + t];
    m /= (float)T;
    float q = 0.f;
    for (int t = 0; t < T; ++t) { const float d = rm[c * T + t] - m; q += rq[c * T + t] + (float)V * d * d; }
    cm[c] = m;
    cs[c] = sqrtf(q / (float)(T * V - 1));
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void cg_dstd_stats_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int T, int V) {
  float* rm = (float*)cg_dyn_lds;
  float* rq = rm + C * T;
  float* cm = rq + C * T;
  float* cs = cm + C;
  const int b = blockIdx.x;
  const float* xb = x + (long long)b * C * T * V;
  cg_stats_rows(xb, C, T, V, rm, rq, cm, cs);
  float* o = out + (long long)b * (2 + 2 * T);
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
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float m = 0.f, sm = 0.f;
    for (int c = 0; c < C; ++c) { m += rm[c * T + t]; sm += sqrtf(rq[c * T + t] / (float)(V - 1)); }
    o[1 + t] = m / (float)C;
    sm /= (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float d = sqrtf(rq[c * T + t] / (float)(V - 1)) - sm; q += d * d; }
    o[2 + T + t] = sqrtf(q / (float)(C - 1));
  }
}

// backward: d x[c,t,v] = x * P[c,t] + Q[c,t] with row coefficients from the four statistics' gradients
__global__ __launch_bounds__(1024) void cg_dstd_stats_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, float* __restrict__ dx,
                                         int C, int T, int V) {
  float* rm = (float*)cg_dyn_lds;
  float* rq = rm + C * T;
  float* cm = rq + C * T;
  float* cs = cm + C;
  float* tS = cs + C;       // [T] std over c of row stds
  float* tM = tS + T;       // [T] mean over c of row stds
  float* rP = tM + T;       // [C*T] slope of the row
  float* rQ = rP + C * T;   // [C*T] offset of the row
  __shared__ float gS, gSm;
  const int b = blockIdx.x;
  const float* xb = x + (long long)b * C * T * V;
  const float* g = dout + (long long)b * (2 + 2 * T);
  cg_stats_rows(xb, C, T, V, rm, rq, cm, cs);
  if (threadIdx.x == 0) {
    float sm = 0.f;
    for (int c = 0; c < C; ++c) sm += cs[c];
    sm /= (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float d = cs[c] - sm; q += d * d; }
    gS = sqrtf(q / (float)(C - 1));
    gSm = sm;
  }
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float sm = 0.f;
    for (int c = 0; c < C; ++c) sm += sqrtf(rq[c * T + t] / (float)(V - 1));
    sm /= (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float d = sqrtf(rq[c * T + t] / (float)(V - 1)) - sm; q += d * d; }
    tS[t] = sqrtf(q / (float)(C - 1));
    tM[t] = sm;
  }
  __syncthreads();
  const float g0 = g[0] / (float)(C * T * V);
  const float gall = g[1 + T];
  for (int r = threadIdx.x; r < C * T; r += blockDim.x) {
    const int c = r / T, t = r - c * T;
    // std over c of s_c, s_c = std over (t,v);  std over c of s_ct, s_ct = std over v
    const float pc = gall * (cs[c] - gSm) / ((float)(C - 1) * gS) / ((float)(T * V - 1) * cs[c]);
    const float sct = sqrtf(rq[r] / (float)(V - 1));
    const float pr = g[2 + T + t] * (sct - tM[t]) / ((float)(C - 1) * tS[t]) / ((float)(V - 1) * sct);
    rP[r] = pc + pr;
    rQ[r] = g0 + g[1 + t] / (float)(C * V) - pc * cm[c] - pr * rm[r];
  }
  __syncthreads();
  float* dxb = dx + (long long)b * C * T * V;
  const int lpr = cg_stats_lpr(V), gl = threadIdx.x & (lpr - 1), grp = threadIdx.x / lpr, ngrp = blockDim.x / lpr;
#pragma unroll 4
  for (int r = grp; r < C * T; r += ngrp) {
    const float P = rP[r], Q = rQ[r];
    for (int v = gl; v < V; v += 64) dxb[(long long)r * V + v] = xb[(long long)r * V + v] * P + Q;
  }
}

static size_t cg_dstd_lds(int C, int T) { return (size_t)(4 * C * T + 2 * C + 2 * T) * sizeof(float); }

extern "C" int cg_dstd_stats_fwd(const float* x, float* out, int B, int C, int T, int V, void* stream_) {
  if (!x || !out) return CG_EARG;
  if (B <= 0 || C < 2 || T <= 0 || V < 2 || cg_dstd_lds(C, T) > 160 * 1024) return CG_ESHAPE;
  if (cg_lds_limit((const void*)cg_dstd_stats_fwd_kernel, cg_dstd_lds(C, T)) != hipSuccess) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_dstd_stats_fwd_kernel, dim3(B), dim3(1024), cg_dstd_lds(C, T), (hipStream_t)stream_, x, out, C, T, V);
  return cg_launch_status();
}

extern "C" int cg_dstd_stats_bwd(const float* x, const float* dout, float* dx, int B, int C, int T, int V, void* stream_) {
  if (!x || !dout || !dx) return CG_EARG;
  if (B <= 0 || C < 2 || T <= 0 || V < 2 || cg_dstd_lds(C, T) > 160 * 1024) return CG_ESHAPE;
  if (cg_lds_limit((const void*)cg_dstd_stats_bwd_kernel, cg_dstd_lds(C, T)) != hipSuccess) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_dstd_stats_bwd_kernel, dim3(B), dim3(1024), cg_dstd_lds(C, T), (hipStream_t)stream_, x, dout, dx, C, T, V);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Row H — SELayer gate, SE.py:9-14 / 30-35:  gate = sigmoid(W2 relu(W1 pooled)),  W1 (H,C), W2 (C,H)
// One workgroup per sample; LDS: pooled[C] | hidden[H] | scratch[C]
// ---------------------------------------------------------------------------------------------
__global__ void cg_se_gate_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ W1, const float* __restrict__ W2,
                                      float* __restrict__ gate, int C, int H) {
  float* sp = (float*)cg_dyn_lds;
  float* sh = sp + C;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) sp[c] = pooled[(long long)b * C + c];
  __syncthreads();
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += W1[j * C + c] * sp[c];
    sh[j] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < H; ++j) s += W2[c * H + j] * sh[j];
    gate[(long long)b * C + c] = 1.f / (1.f + expf(-s));
  }
}

__global__ void cg_se_gate_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ W1, const float* __restrict__ W2,
                                      const float* __restrict__ gate, const float* __restrict__ dgate,
                                      float* __restrict__ dpooled, float* __restrict__ dW1, float* __restrict__ dW2, int C, int H) {
  float* sp = (float*)cg_dyn_lds;
  float* sh = sp + C;     // hidden (post-ReLU)
  float* sz = sh + H;     // d z2 [C]
  float* sd = sz + C;     // d z1 [H]
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    sp[c] = pooled[(long long)b * C + c];
    const float gt = gate[(long long)b * C + c];
    sz[c] = dgate[(long long)b * C + c] * gt * (1.f - gt);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += W1[j * C + c] * sp[c];
    sh[j] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += W2[c * H + j] * sz[c];
    sd[j] = sh[j] > 0.f ? s : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * H; i += blockDim.x) {
    const int c = i / H, j = i % H;
    atomicAdd(&dW2[c * H + j], sz[c] * sh[j]);
    atomicAdd(&dW1[j * C + c], sd[j] * sp[c]);
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < H; ++j) s += W1[j * C + c] * sd[j];
    dpooled[(long long)b * C + c] = s;
  }
}

// The same gradients with the weight gradients kept in registers: a workgroup walks a slice of the batch and adds its dW1 / dW2
// partial sums once at the end.  One workgroup per sample sent B atomics to every one of the 2 C H addresses (B = 256: 24 us for a
// 64 x 8 gate, 80 us for the 3 x 1 gate of the output block, where all of them hit the same six words).
#define CG_SE_THREADS 256
#define CG_SE_MAXACC 8                         // C * H <= CG_SE_THREADS * CG_SE_MAXACC
__global__ __launch_bounds__(CG_SE_THREADS) void cg_se_gate_bwd_sliced_kernel(const float* __restrict__ pooled, const float* __restrict__ W1,
                                                                               const float* __restrict__ W2, const float* __restrict__ gate,
                                                                               const float* __restrict__ dgate, float* __restrict__ dpooled,
                                                                               float* __restrict__ dW1, float* __restrict__ dW2, int B, int C, int H) {
  float* sp = (float*)cg_dyn_lds;
  float* sh = sp + C;     // hidden (post-ReLU)
  float* sz = sh + H;     // d z2 [C]
  float* sd = sz + C;     // d z1 [H]
  const int tid = threadIdx.x, lane = tid & (CG_WAVE - 1), wave = tid / CG_WAVE, nw = CG_SE_THREADS / CG_WAVE;
  float acc1[CG_SE_MAXACC], acc2[CG_SE_MAXACC];
#pragma unroll
  for (int u = 0; u < CG_SE_MAXACC; ++u) { acc1[u] = 0.f; acc2[u] = 0.f; }
  // the next sample's row (channel tid; wider gates take the loop) travels while this one is worked on
  float np_ = 0.f, ng_ = 0.f, nd_ = 0.f;
  if (tid < C && (int)blockIdx.x < B) { const long long o = (long long)blockIdx.x * C + tid; np_ = pooled[o]; ng_ = gate[o]; nd_ = dgate[o]; }
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    if (tid < C) { sp[tid] = np_; sz[tid] = nd_ * ng_ * (1.f - ng_); }
    for (int c = tid + CG_SE_THREADS; c < C; c += CG_SE_THREADS) {
      sp[c] = pooled[(long long)b * C + c];
      const float gt = gate[(long long)b * C + c];
      sz[c] = dgate[(long long)b * C + c] * gt * (1.f - gt);
    }
    if (tid < C && b + (int)gridDim.x < B) { const long long o = (long long)(b + gridDim.x) * C + tid; np_ = pooled[o]; ng_ = gate[o]; nd_ = dgate[o]; }
    __syncthreads();
    for (int j = wave; j < H; j += nw) {                    // one wave per hidden unit, lanes over the channels
      float s = 0.f, d = 0.f;
      for (int c = lane; c < C; c += CG_WAVE) { s += W1[j * C + c] * sp[c]; d += W2[c * H + j] * sz[c]; }
      s = cg_wave_sum(s); d = cg_wave_sum(d);
      if (lane == 0) { sh[j] = s > 0.f ? s : 0.f; sd[j] = s > 0.f ? d : 0.f; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CG_SE_MAXACC; ++u) {
      const int i = tid + CG_SE_THREADS * u;
      if (i < C * H) { const int c = i / H, j = i - c * H; acc2[u] += sz[c] * sh[j]; acc1[u] += sd[j] * sp[c]; }
    }
    for (int c = tid; c < C; c += CG_SE_THREADS) {
      float s = 0.f;
      for (int j = 0; j < H; ++j) s += W1[j * C + c] * sd[j];
      dpooled[(long long)b * C + c] = s;
    }
  }
#pragma unroll
  for (int u = 0; u < CG_SE_MAXACC; ++u) {
    const int i = tid + CG_SE_THREADS * u;
    if (i < C * H) { const int c = i / H, j = i - c * H; atomicAdd(&dW2[c * H + j], acc2[u]); atomicAdd(&dW1[j * C + c], acc1[u]); }
  }
}

extern "C" int cg_se_gate_fwd(const float* pooled, const float* W1, const float* W2, float* gate, int B, int C, int H, void* stream_) {
  if (!pooled || !W1 || !W2 || !gate) return CG_EARG;
  if (B <= 0 || C <= 0 || H <= 0) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_se_gate_fwd_kernel, dim3(B), dim3(64), (size_t)(C + H) * 4, (hipStream_t)stream_, pooled, W1, W2, gate, C, H);
  return cg_launch_status();
}

// dW1 (H,C) and dW2 (C,H) are accumulated with f32 atomics over the batch; they are zeroed here unless the caller
// hands over zero-filled memory (prezeroed != 0: slices of a scratch pool cleared once per step).
extern "C" int cg_se_gate_bwd(const float* pooled, const float* W1, const float* W2, const float* gate, const float* dgate,
                              float* dpooled, float* dW1, float* dW2, int B, int C, int H, int prezeroed, void* stream_) {
  if (!pooled || !W1 || !W2 || !gate || !dgate || !dpooled || !dW1 || !dW2) return CG_EARG;
  if (B <= 0 || C <= 0 || H <= 0) return CG_ESHAPE;
  hipStream_t stream = (hipStream_t)stream_;
  if (!prezeroed) {
    int zs = cg_zero_fill(dW1, (long long)C * H * 4, stream);
    if (zs == CG_OK) zs = cg_zero_fill(dW2, (long long)C * H * 4, stream);
    if (zs != CG_OK) return zs;
  }
  if ((long long)C * H <= CG_SE_THREADS * CG_SE_MAXACC) {
    const int nwg = B < 64 ? B : 64;
    hipLaunchKernelGGL(cg_se_gate_bwd_sliced_kernel, dim3(nwg), dim3(CG_SE_THREADS), (size_t)(2 * C + 2 * H) * 4, stream, pooled, W1, W2, gate, dgate,
                       dpooled, dW1, dW2, B, C, H);
    return cg_launch_status();
  }
  hipLaunchKernelGGL(cg_se_gate_bwd_kernel, dim3(B), dim3(64), (size_t)(2 * C + 2 * H) * 4, stream, pooled, W1, W2, gate, dgate, dpooled, dW1, dW2, C, H);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Row J — cumulative sum over axis 1 of a strided 4-D view (B, L, R1, R2), CISTGCN.py:589.
// reverse = 1 gives the adjoint (suffix sums).
// ---------------------------------------------------------------------------------------------
__global__ void cg_cumsum_kernel(const float* __restrict__ x, CgView4 xv, float* __restrict__ y, CgView4 yv, int reverse) {
  const long long B = xv.n[0], L = xv.n[1], R1 = xv.n[2], R2 = xv.n[3];
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * R1 * R2) return;
  const long long r2 = i % R2, r1 = (i / R2) % R1, b = i / (R1 * R2);
  const long long bx = b * xv.s[0] + r1 * xv.s[2] + r2 * xv.s[3];
  const long long by = b * yv.s[0] + r1 * yv.s[2] + r2 * yv.s[3];
  float s = 0.f;
  for (long long l = 0; l < L; ++l) {
    const long long ll = reverse ? L - 1 - l : l;
    s += x[bx + ll * xv.s[1]];
    y[by + ll * yv.s[1]] = s;
  }
}

extern "C" int cg_cumsum(const float* x, const CgView4* xv, float* y, const CgView4* yv, int reverse, void* stream_) {
  if (!x || !y || !xv || !yv) return CG_EARG;
  const long long n = xv->n[0] * xv->n[2] * xv->n[3];
  if (n <= 0 || xv->n[1] <= 0) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_cumsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, *xv, y, *yv, reverse);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Row L — MPJPE, losses/losses.py:50-61 with reduce_axis=[]:  mean over (b,t,v) of ||pred - target||_2
// pred is a strided (N, 3) view (N = B*T*V rows with stride ps_n, coordinate stride ps_d).
// ---------------------------------------------------------------------------------------------
__global__ void cg_mpjpe_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, float* __restrict__ loss, long long N) {
  __shared__ double red[16];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
    const float a = pred[3 * i] - tgt[3 * i], b = pred[3 * i + 1] - tgt[3 * i + 1], c = pred[3 * i + 2] - tgt[3 * i + 2];
    s += (double)sqrtf(a * a + b * b + c * c);
  }
  s = cg_block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(loss, (float)(s / (double)N));
}

__global__ void cg_mpjpe_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ gloss,
                                    float* __restrict__ dpred, long long N) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float a = pred[3 * i] - tgt[3 * i], b = pred[3 * i + 1] - tgt[3 * i + 1], c = pred[3 * i + 2] - tgt[3 * i + 2];
  const float n = sqrtf(a * a + b * b + c * c);
  const float k = n > 0.f ? gloss[0] / ((float)N * n) : 0.f;
  dpred[3 * i] = a * k; dpred[3 * i + 1] = b * k; dpred[3 * i + 2] = c * k;
}

extern "C" int cg_mpjpe_fwd(const float* pred, const float* tgt, float* loss, long long N, void* stream_) {
  if (!pred || !tgt || !loss) return CG_EARG;
  if (N <= 0) return CG_ESHAPE;
  hipStream_t stream = (hipStream_t)stream_;
  const int zs = cg_zero_fill(loss, (long long)sizeof(float), stream);
  if (zs != CG_OK) return zs;
  long long blocks = (N + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(cg_mpjpe_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pred, tgt, loss, N);
  return cg_launch_status();
}

extern "C" int cg_mpjpe_bwd(const float* pred, const float* tgt, const float* gloss, float* dpred, long long N, void* stream_) {
  if (!pred || !tgt || !gloss || !dpred) return CG_EARG;
  if (N <= 0) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_mpjpe_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, pred, tgt, gloss, dpred, N);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// dropout-seed bookkeeping: one device word, bumped once per training step so that graph replays
// draw fresh masks while forward and backward of one step agree.
// ---------------------------------------------------------------------------------------------
__global__ void cg_seed_bump_kernel(unsigned long long* seed) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *seed = *seed * 6364136223846793005ull + 1442695040888963407ull;
}

extern "C" int cg_seed_bump(unsigned long long* seed, void* stream_) {
  if (!seed) return CG_EARG;
  hipLaunchKernelGGL(cg_seed_bump_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, seed);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Row G — rank-1 adjacency seed of Map2Adj (CISTGCN.py:183-189): the joint and time summaries s (B,V,T) and q (B,T,V)
// are expanded to one J x J slab per (sample, joint) or (sample, frame):
//   space (domain 0): o[b,v,t,u] = s[b,v,t] * q[b,u,v]      (B,V,T,T)
//   time  (domain 1): o[b,t,v,w] = s[b,v,t] * q[b,t,w]      (B,T,V,V)
// Forward is a pure write stream; backward reads each slab of d o ONCE and produces both d s (row sums against q)
// and d q (column sums against s) - as two generic contractions it was read twice through 64x64 tiles of which one
// column was used.  One workgroup per slab, both domains of a block in one launch (blockIdx.y).
// ---------------------------------------------------------------------------------------------
#define CG_R1_JMAX 64
struct CgRank1 { const float* s; const float* q; float* o; const float* dout; float* ds; float* dq; int domain; int pad; };
struct CgRank1Batch { int n, B, T, V; CgRank1 it[2]; };

// slab j of sample b: vectors sv[i], qv[k] and their strides
__device__ __forceinline__ void cg_rank1_geom(int domain, int b, int j, int T, int V, int& J, long long& s0, int& ss, long long& q0, int& qs) {
  if (domain == 0) { J = T; s0 = ((long long)b * V + j) * T; ss = 1; q0 = (long long)b * T * V + j; qs = V; }       // j = joint v
  else             { J = V; s0 = (long long)b * V * T + j; ss = T; q0 = ((long long)b * T + j) * V; qs = 1; }       // j = frame t
}

__global__ void cg_rank1_adj_fwd_kernel(CgRank1Batch batch) {
  __shared__ float sv[CG_R1_JMAX], qv[CG_R1_JMAX];
  const CgRank1& it = batch.it[blockIdx.y];
  const int NG = it.domain == 0 ? batch.V : batch.T;
  const int b = blockIdx.x / NG, j = blockIdx.x - b * NG;
  if (b >= batch.B) return;
  int J, ss, qs; long long s0, q0;
  cg_rank1_geom(it.domain, b, j, batch.T, batch.V, J, s0, ss, q0, qs);
  if ((int)threadIdx.x < J) sv[threadIdx.x] = it.s[s0 + (long long)threadIdx.x * ss];
  else if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + J) qv[threadIdx.x - 64] = it.q[q0 + (long long)(threadIdx.x - 64) * qs];
  __syncthreads();
  float* o = it.o + ((long long)b * NG + j) * J * J;
  const int JJ = J * J;
  if ((JJ & 3) == 0) {
    for (int e = 4 * threadIdx.x; e < JJ; e += 4 * blockDim.x) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int i = (e + r) / J, k = (e + r) - i * J; v[r] = sv[i] * qv[k]; }
      *reinterpret_cast<float4*>(o + e) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    for (int e = threadIdx.x; e < JJ; e += blockDim.x) { const int i = e / J, k = e - i * J; o[e] = sv[i] * qv[k]; }
  }
}

__global__ void cg_rank1_adj_bwd_kernel(CgRank1Batch batch) {
  __shared__ float sv[CG_R1_JMAX], qv[CG_R1_JMAX];
  __shared__ float slab[CG_R1_JMAX * (CG_R1_JMAX + 1)];
  const CgRank1& it = batch.it[blockIdx.y];
  const int NG = it.domain == 0 ? batch.V : batch.T;
  const int b = blockIdx.x / NG, j = blockIdx.x - b * NG;
  if (b >= batch.B) return;
  int J, ss, qs; long long s0, q0;
  cg_rank1_geom(it.domain, b, j, batch.T, batch.V, J, s0, ss, q0, qs);
  const int LD = J + 1;
  if ((int)threadIdx.x < J) sv[threadIdx.x] = it.s[s0 + (long long)threadIdx.x * ss];
  else if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + J) qv[threadIdx.x - 64] = it.q[q0 + (long long)(threadIdx.x - 64) * qs];
  const float* d = it.dout + ((long long)b * NG + j) * J * J;
  const int JJ = J * J;
  if ((JJ & 3) == 0) {
    for (int e = 4 * threadIdx.x; e < JJ; e += 4 * blockDim.x) {
      const float4 v = *reinterpret_cast<const float4*>(d + e);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int i = (e + r) / J, k = (e + r) - i * J; slab[i * LD + k] = vv[r]; }
    }
  } else {
    for (int e = threadIdx.x; e < JJ; e += blockDim.x) { const int i = e / J, k = e - i * J; slab[i * LD + k] = d[e]; }
  }
  __syncthreads();
  // threads 0..J-1: d s[i] = sum_k slab[i][k] q[k];  threads 64..64+J-1: d q[k] = sum_i slab[i][k] s[i]
  if ((int)threadIdx.x < J) {
    const int i = threadIdx.x;
    float acc = 0.f;
    for (int k = 0; k < J; ++k) acc = fmaf(slab[i * LD + k], qv[k], acc);
    it.ds[s0 + (long long)i * ss] = acc;
  } else if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + J) {
    const int k = threadIdx.x - 64;
    float acc = 0.f;
    for (int i = 0; i < J; ++i) acc = fmaf(slab[i * LD + k], sv[i], acc);
    it.dq[q0 + (long long)k * qs] = acc;
  }
}

static int cg_rank1_check(const CgRank1* items, int n, int B, int T, int V, bool bwd) {
  if (!items || n <= 0 || n > 2) return CG_EARG;
  if (B <= 0 || T <= 0 || V <= 0 || T > CG_R1_JMAX || V > CG_R1_JMAX) return CG_ESHAPE;
  for (int i = 0; i < n; ++i) {
    if (!items[i].s || !items[i].q || (items[i].domain != 0 && items[i].domain != 1)) return CG_EARG;
    if (!bwd && (!items[i].o || ((uintptr_t)items[i].o & 15))) return CG_EARG;
    if (bwd && (!items[i].dout || !items[i].ds || !items[i].dq || ((uintptr_t)items[i].dout & 15))) return CG_EARG;
  }
  return CG_OK;
}

// include/cistgcn_hip.h : cg_rank1_adj_fwd / cg_rank1_adj_bwd
extern "C" int cg_rank1_adj_fwd(const CgRank1* items, int n, int B, int T, int V, void* stream_) {
  int st = cg_rank1_check(items, n, B, T, V, false);
  if (st != CG_OK) return st;
  CgRank1Batch batch;
  batch.n = n; batch.B = B; batch.T = T; batch.V = V;
  for (int i = 0; i < 2; ++i) batch.it[i] = items[i < n ? i : 0];
  const int ng = T > V ? T : V;
  hipLaunchKernelGGL(cg_rank1_adj_fwd_kernel, dim3((unsigned)((long long)B * ng), (unsigned)n), dim3(256), 0, (hipStream_t)stream_, batch);
  return cg_launch_status();
}

extern "C" int cg_rank1_adj_bwd(const CgRank1* items, int n, int B, int T, int V, void* stream_) {
  int st = cg_rank1_check(items, n, B, T, V, true);
  if (st != CG_OK) return st;
  CgRank1Batch batch;
  batch.n = n; batch.B = B; batch.T = T; batch.V = V;
  for (int i = 0; i < 2; ++i) batch.it[i] = items[i < n ? i : 0];
  const int ng = T > V ? T : V;
  hipLaunchKernelGGL(cg_rank1_adj_bwd_kernel, dim3((unsigned)((long long)B * ng), (unsigned)n), dim3(256), 0, (hipStream_t)stream_, batch);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Evaluation harness counterpart, environment/test.py:97-132 + losses.py:50-61:
//   inputs[:, :, dim_used]                                     -> cg_gather_joints
//   mygt = target.clone(); mygt[:, :, dim_used] = outputs; mygt[:, :, dim_repeat_32] = outputs[:, :, dim_repeat_22]
//   mpjpe(mygt, target, reduce_axis=(0, 2))  (per predicted frame)   -> cg_eval_scatter_mpjpe (one pass, one launch)
// `src[j]` = joint of the prediction that full-skeleton joint j takes, or -1 (keeps the ground truth).
// ---------------------------------------------------------------------------------------------
__global__ void cg_gather_joints_kernel(const float* __restrict__ x, float* __restrict__ y, const int32_t* __restrict__ idx,
                                        long long rows, int Jin, int Jout) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * Jout) return;
  const long long r = i / Jout;
  const int k = (int)(i - r * Jout);
  const float* s = x + (r * Jin + idx[k]) * 3;
  float* d = y + i * 3;
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

extern "C" int cg_gather_joints(const float* x, float* y, const int32_t* idx, long long rows, int Jin, int Jout, void* stream_) {
  if (!x || !y || !idx) return CG_EARG;
  if (rows <= 0 || Jin <= 0 || Jout <= 0) return CG_ESHAPE;
  const long long n = rows * Jout;
  hipLaunchKernelGGL(cg_gather_joints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, y, idx, rows, Jin, Jout);
  return cg_launch_status();
}

// one workgroup per (sample, frame): writes the J32 joints of `out` and adds the frame's mean joint error / B
__global__ void cg_eval_scatter_mpjpe_kernel(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ out,
                                             float* __restrict__ frame_err, const int32_t* __restrict__ src, int B, int To, int J32, int J22) {
  __shared__ float red[16];
  const int b = blockIdx.x / To, t = blockIdx.x - b * To;
  const float* tg = target + ((long long)b * To + t) * J32 * 3;
  const float* pr = pred + ((long long)b * To + t) * J22 * 3;
  float* o = out + ((long long)b * To + t) * J32 * 3;
  float e = 0.f;
  for (int j = threadIdx.x; j < J32; j += blockDim.x) {
    const int k = src[j];
    const float gx = tg[3 * j], gy = tg[3 * j + 1], gz = tg[3 * j + 2];
    const float px = k >= 0 ? pr[3 * k] : gx, py = k >= 0 ? pr[3 * k + 1] : gy, pz = k >= 0 ? pr[3 * k + 2] : gz;
    o[3 * j] = px; o[3 * j + 1] = py; o[3 * j + 2] = pz;
    const float dx = px - gx, dy = py - gy, dz = pz - gz;
    e += sqrtf(dx * dx + dy * dy + dz * dz);
  }
  e = cg_block_sum(e, red);
  if (threadIdx.x == 0) atomicAdd(&frame_err[t], e / ((float)J32 * (float)B));
}

extern "C" int cg_eval_scatter_mpjpe(const float* pred, const float* target, float* out, float* frame_err, const int32_t* src,
                                     int B, int To, int J32, int J22, void* stream_) {
  if (!pred || !target || !out || !frame_err || !src) return CG_EARG;
  if (B <= 0 || To <= 0 || J32 <= 0 || J22 <= 0) return CG_ESHAPE;
  hipStream_t stream = (hipStream_t)stream_;
  const int zs = cg_zero_fill(frame_err, (long long)To * (long long)sizeof(float), stream);
  if (zs != CG_OK) return zs;
  hipLaunchKernelGGL(cg_eval_scatter_mpjpe_kernel, dim3((unsigned)((long long)B * To)), dim3(64), 0, stream, pred, target, out, frame_err,
                     src, B, To, J32, J22);
  return cg_launch_status();
}
