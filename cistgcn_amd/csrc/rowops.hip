// Row kernels: everything that is per-channel affine / reduction over a 4-D strided view
// (n0 = batch, n1 = channel, n2 x n3 = positions).  One workgroup owns one (channel, batch) row, so
// per-row results (gate gradients) need no atomics and per-channel sums need one f64 atomic per
// row.  Batch statistics are accumulated in f64 (sum, sum of squares): mm-scale pose coordinates
// square to ~1e6 and the E[x^2]-E[x]^2 form is not safe in f32.
//
// Reference semantics restated here: nn.BatchNorm{1,2}d (train: biased batch variance, running_var
// updated with the unbiased one, momentum 0.1, eps 1e-5; eval: running stats), nn.Dropout(inplace),
// nn.PReLU (shared or per-channel alpha) as used throughout CISTGCN.py (e.g. :138-153, :229-237,
// :305-358), SELayer scale (SE.py:20,41).
#include "cg_common.h"

// Row loop: p runs over the n2*n3 positions of one (batch, channel) row; (i2, i3) are derived once per
// element with one 32-bit division and shared by every view of the kernel (all views have the same dims).
#define CG_ROW_LOOP(P_, p_) for (int p_ = threadIdx.x; p_ < (int)(P_); p_ += blockDim.x)
#define CG_POS(v_, p_) const int i2_ = (p_) / (int)(v_).n[3]; const int i3_ = (p_) - i2_ * (int)(v_).n[3];
// Chunked rows: a workgroup owns channel c and the batch rows [b0, b0 + nb); e runs over nb * P elements.
#define CG_CHUNK_LOOP(nb_, P_, e_) for (int e_ = threadIdx.x; e_ < (nb_) * (int)(P_); e_ += blockDim.x)
#define CG_CHUNK_ROW(e_, P_, b0_) const int br_ = (e_) / (int)(P_); const int p = (e_) - br_ * (int)(P_); const int b = (b0_) + br_;
#define CG_OFF(v_) ((long long)i2_ * (v_).s[2] + (long long)i3_ * (v_).s[3])

__device__ __forceinline__ long long cg_row_base(const CgView4& v, int b, int c) {
  return (long long)b * v.s[0] + (long long)c * v.s[1];
}

// ---------------------------------------------------------------------------------------------
// per-channel sums  stats[c] = { sum_{b,p} v, sum v^2 },  v = x * pre[b,c]
// ---------------------------------------------------------------------------------------------
// rows of one channel handled per workgroup: ~4096 elements, so that the block reduction and the f64 atomics
// are amortised (a (B,C) BatchNorm1d input becomes one workgroup per channel)
static int cg_rows_per_block(const CgView4& v) {
  const long long P = v.n[2] * v.n[3];
  // large tensors: ~2048 workgroups (8 per CU), up to 32 K elements each, so that the per-workgroup prologue
  // (replica fold, affine) and epilogue (block sums, atomics) stay small next to the streaming part
  long long target = (v.n[0] * v.n[1] * P) / 2048;
  target = target < 4096 ? 4096 : (target > 32768 ? 32768 : target);
  long long rb = target / (P > 0 ? P : 1);
  if (rb < 1) rb = 1;
  if (rb > v.n[0]) rb = v.n[0];
  // small problems: keep at least ~512 workgroups in flight rather than amortising
  while (rb > 1 && v.n[1] * ((v.n[0] + rb - 1) / rb) < 512) rb = (rb + 1) / 2;
  return (int)rb;
}

#define CG_ROW_MAX_BATCH 6     // problems per launch of the row kernels (kernel-argument budget)
// Vector width of the contiguous-row paths: 4 (rows 16-byte aligned, P % 4 == 0) or 2 (rows 8-byte aligned, P % 2 == 0: the
// (T,V) planes of the 25-joint skeletons, P = 1250, and the (25, 66) planes of the output block).  The width is uniform per
// problem; lanes j >= vw of a group are never used.
__device__ __forceinline__ void cg_ldv(const float* p, int vw, float out[4]) {
  if (vw == 4) { const float4 v = *reinterpret_cast<const float4*>(p); out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; }
  else { const float2 v = *reinterpret_cast<const float2*>(p); out[0] = v.x; out[1] = v.y; out[2] = 0.f; out[3] = 0.f; }
}
__device__ __forceinline__ void cg_stv(float* p, int vw, const float v[4]) {
  if (vw == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
}

struct CgStatsItem { const float* x; CgView4 xv; const float* pre; double* stats; int rb; int pad; };
struct CgStatsBatch { int n; int pad; CgStatsItem it[CG_ROW_MAX_BATCH]; };

__global__ void cg_chan_stats_kernel(CgStatsBatch batch) {
  __shared__ double red[32];
  const CgStatsItem& it = batch.it[blockIdx.z];
  const CgView4& xv = it.xv;
  const int rb = it.rb;
  const int c = blockIdx.x, b0 = blockIdx.y * rb;
  if (c >= xv.n[1] || b0 >= xv.n[0]) return;
  const float* __restrict__ x = it.x; const float* __restrict__ pre = it.pre;
  const int P = (int)(xv.n[2] * xv.n[3]);
  const int nb = min(rb, (int)xv.n[0] - b0);
  double s = 0.0, q = 0.0;
  if (it.pad) {                                   // pad = 4 / 2: contiguous rows, 16- / 8-byte aligned (vector path)
    const int vw = it.pad, PV = P / vw;
    for (int e = threadIdx.x; e < nb * PV; e += blockDim.x) {
      const int br = e / PV, p = vw * (e - br * PV), b = b0 + br;
      const float w = pre ? pre[(long long)b * xv.n[1] + c] : 1.f;
      float v[4];
      cg_ldv(x + cg_row_base(xv, b, c) + p, vw, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= w;
      s += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
      q += ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
    }
  } else {
    CG_CHUNK_LOOP(nb, P, e) {
      CG_CHUNK_ROW(e, P, b0)
      CG_POS(xv, p)
      const float w = pre ? pre[(long long)b * xv.n[1] + c] : 1.f;
      const float v = x[cg_row_base(xv, b, c) + CG_OFF(xv)] * w;
      s += (double)v;
      q += (double)v * (double)v;
    }
  }
  s = cg_block_sum(s, red);
  q = cg_block_sum(q, red + 16);
  if (threadIdx.x == 0) {
    double* rep = it.stats + (long long)(blockIdx.y % CG_STAT_REPLICAS) * 2 * xv.n[1];
    atomicAdd(&rep[2 * c], s);
    atomicAdd(&rep[2 * c + 1], q);
  }
}

struct CgStatsArgs { const float* x; CgView4 xv; const float* pre; double* stats; };
static int cg_view_vec(const void* ptr, const CgView4& v);       // vector width of contiguous aligned rows: 4, 2 or 0 (defined with the row kernels below)

// include/cistgcn_hip.h : cg_chan_stats_many (up to CG_ROW_MAX_BATCH tensors per launch)
extern "C" int cg_chan_stats_many(const CgStatsArgs* items, int n, void* stream_) {
  if (!items || n <= 0 || n > CG_ROW_MAX_BATCH) return CG_EARG;
  CgStatsBatch batch;
  batch.n = n; batch.pad = 0;
  long long gx = 1, gy = 1, big = 0;
  for (int i = 0; i < n; ++i) {
    const CgStatsArgs& a = items[i];
    if (!a.x || !a.stats) return CG_EARG;
    const long long P = a.xv.n[2] * a.xv.n[3];
    if (a.xv.n[0] <= 0 || a.xv.n[1] <= 0 || P <= 0) return CG_ESHAPE;
    const int rb = cg_rows_per_block(a.xv);
    batch.it[i].x = a.x; batch.it[i].xv = a.xv; batch.it[i].pre = a.pre; batch.it[i].stats = a.stats; batch.it[i].rb = rb;
    batch.it[i].pad = cg_view_vec(a.x, a.xv);                  // vector width of the contiguous-row path (0: none)
    gx = gx > a.xv.n[1] ? gx : a.xv.n[1];
    const long long chunks = (a.xv.n[0] + rb - 1) / rb;
    gy = gy > chunks ? gy : chunks;
    if (P * rb > 256) big = 1;
  }
  if (gy > 65535 || gx > 2147483647LL) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_chan_stats_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)n), dim3(big ? 256 : 64), 0, (hipStream_t)stream_, batch);
  return cg_launch_status();
}

extern "C" int cg_chan_stats(const float* x, const CgView4* xv, const float* pre, double* stats, void* stream_) {
  if (!x || !xv || !stats) return CG_EARG;
  CgStatsArgs a;
  a.x = x; a.xv = *xv; a.pre = pre; a.stats = stats;
  return cg_chan_stats_many(&a, 1, stream_);
}

// per-channel plain sum (bias gradients): out[c] += sum_{b,p} x, `out` zero on entry.  A workgroup per (channel, batch slice):
// f64 inside the workgroup, one f32 atomic per workgroup.
__global__ void cg_chan_sum_kernel(const float* __restrict__ x, CgView4 xv, float* __restrict__ out, int rb) {
  __shared__ double red[16];
  const int c = blockIdx.x, b0 = blockIdx.y * rb, b1 = min((int)xv.n[0], b0 + rb);
  const long long P = xv.n[2] * xv.n[3];
  double s = 0.0;
  for (int b = b0; b < b1; ++b) {
    const long long base = cg_row_base(xv, b, c);
    CG_ROW_LOOP(P, p) { CG_POS(xv, p) s += (double)x[base + CG_OFF(xv)]; }
  }
  s = cg_block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(&out[c], (float)s);
}

extern "C" int cg_chan_sum(const float* x, const CgView4* xv, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !xv || !out) return CG_EARG;
  if (xv->n[1] <= 0 || xv->n[0] <= 0) return CG_ESHAPE;
  long long nb = 1024 / xv->n[1];                       // ~1024 workgroups
  nb = nb < 1 ? 1 : (nb > xv->n[0] ? xv->n[0] : nb);
  const int rb = (int)((xv->n[0] + nb - 1) / nb);
  hipLaunchKernelGGL(cg_chan_sum_kernel, dim3((unsigned)xv->n[1], (unsigned)((xv->n[0] + rb - 1) / rb)), dim3(256), 0, stream, x, *xv, out, rb);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// fused  y = PReLU( Dropout( (x*pre) * scale + shift ) [+ add] ) [+ add if add_post]
// ---------------------------------------------------------------------------------------------
struct CgNormAct {
  const float* x;  CgView4 xv;
  float* y;        CgView4 yv;
  const float* pre;              // (B,C) per-sample channel gate, or null
  const float* add; CgView4 av;  // addend (same logical shape) or null
  int add_post;                  // 0: added before the PReLU, 1: after it
  int bn_mode;                   // 0 none, 1 batch statistics (train), 2 running statistics (eval)
  const double* stats;           // [C][2] sums over B*P (bn_mode 1)
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; long long* num_batches_tracked;
  float momentum, eps;
  float* save_mean; float* save_rstd;    // [C]; written in fwd (both modes), read in bwd
  float drop_p; const unsigned long long* seed; unsigned int salt;
  const float* alpha; int alpha_n;       // PReLU slope(s) or null
  // backward only
  const float* dy; CgView4 dyv;
  float* dx; CgView4 dxv;
  float* dadd; CgView4 dav;     // gradient of a pre-activation addend (null: not needed)
  float* dpre;                  // (B,C) or null
  double* red;                  // [C][2] sum g, sum g*xhat  (+ alpha_n slots behind it for d alpha)
  float* dgamma; float* dbeta; float* dalpha;
  double* ystats;               // forward, optional: [C][2] f64 sums of y (zero on entry) for the BatchNorm that consumes y
};

// u = (v - mean) * scale + shift, scale = gamma * rstd, shift = beta: the mean is subtracted FIRST (as
// nn.BatchNorm does); the folded form v*scale + (beta - mean*scale) cancels catastrophically when
// |mean| >> std.
struct CgChanAffine { float scale, shift, mean, rstd; };
// vec[i] = 4 / 2: every view the kernel touches is a contiguous-row NCHW view (s[3] = 1, s[2] = n[3]) with P % vec == 0 and
// rows aligned to 4 * vec bytes -> vector path (float4 / float2), no per-element index arithmetic
struct CgNormActBatch { int n; int rb[CG_ROW_MAX_BATCH]; int vec[CG_ROW_MAX_BATCH]; int pad; CgNormAct a[CG_ROW_MAX_BATCH]; };

static int cg_view_vec(const void* ptr, const CgView4& v) {
  if (!ptr) return 4;
  const long long P = v.n[2] * v.n[3];
  const bool rows = v.s[3] == 1 && (v.s[2] == v.n[3] || v.n[2] == 1);
  if (!rows) return 0;
  if ((P & 3) == 0 && (v.s[0] & 3) == 0 && (v.s[1] & 3) == 0 && ((uintptr_t)ptr & 15) == 0) return 4;
  if ((P & 1) == 0 && (v.s[0] & 1) == 0 && (v.s[1] & 1) == 0 && ((uintptr_t)ptr & 7) == 0) return 2;
  return 0;
}
// kind: 0 forward, 1 backward reduce, 2 backward apply; returns the common vector width of every view the kernel touches
static int cg_norm_act_vec(const CgNormAct& a, int kind) {
  int vw = cg_view_vec(a.x, a.xv);
  auto also = [&](const void* p, const CgView4& v) { const int w = cg_view_vec(p, v); vw = w < vw ? w : vw; };
  also(a.add, a.av);
  if (kind == 0) also(a.y, a.yv);
  else also(a.dy, a.dyv);
  if (kind == 2) { also(a.dx, a.dxv); also(a.dadd, a.dav); }
  return vw;
}

__device__ __forceinline__ CgChanAffine cg_chan_affine(const CgNormAct& a, int c, bool backward) {
  CgChanAffine r;
  r.scale = 1.f; r.shift = 0.f; r.mean = 0.f; r.rstd = 1.f;
  if (a.bn_mode == 0) return r;
  if (backward) {
    r.mean = a.save_mean[c]; r.rstd = a.save_rstd[c];
  } else if (a.bn_mode == 1) {
    const double cnt = (double)a.xv.n[0] * (double)(a.xv.n[2] * a.xv.n[3]);
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < CG_STAT_REPLICAS; ++r) {      // replicated accumulators (cg_common.h)
      s1 += a.stats[((long long)r * a.xv.n[1] + c) * 2];
      s2 += a.stats[((long long)r * a.xv.n[1] + c) * 2 + 1];
    }
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    r.mean = (float)mean;
    r.rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    if (blockIdx.y == 0 && threadIdx.x == 0) {
      a.save_mean[c] = r.mean; a.save_rstd[c] = r.rstd;
      if (a.running_mean) {
        const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
        a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unb;
        if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
      }
    }
  } else {
    r.mean = a.running_mean[c];
    r.rstd = 1.0f / sqrtf(a.running_var[c] + a.eps);
    if (blockIdx.y == 0 && threadIdx.x == 0) { a.save_mean[c] = r.mean; a.save_rstd[c] = r.rstd; }
  }
  r.scale = a.gamma[c] * r.rstd;
  r.shift = a.beta[c];
  return r;
}

__device__ __forceinline__ float cg_norm_act_y(const CgNormAct& a, const CgChanAffine& af, float alpha, float xval, float w,
                                               float keep, float ad) {
  float u = ((xval * w - af.mean) * af.scale + af.shift) * keep;
  if (a.add && !a.add_post) u += ad;
  if (a.alpha) u = u > 0.f ? u : alpha * u;
  if (a.add && a.add_post) u += ad;
  return u;
}

__global__ void cg_norm_act_fwd_kernel(CgNormActBatch batch) {
  const CgNormAct& a = batch.a[blockIdx.z];
  const int rb = batch.rb[blockIdx.z];
  const int c = blockIdx.x, b0 = blockIdx.y * rb;
  if (c >= a.xv.n[1] || b0 >= a.xv.n[0]) return;
  const long long C = a.xv.n[1];
  const int P = (int)(a.xv.n[2] * a.xv.n[3]);
  const int nb = min(rb, (int)a.xv.n[0] - b0);
  const CgChanAffine af = cg_chan_affine(a, c, false);
  const float alpha = a.alpha ? a.alpha[a.alpha_n == 1 ? 0 : c] : 1.f;
  const bool drop = a.drop_p > 0.f;
  const unsigned long long seed = drop ? *a.seed : 0ull;
  __shared__ double red[32];
  double ys = 0.0, yq = 0.0;
  if (batch.vec[blockIdx.z]) {
    const int vw = batch.vec[blockIdx.z], PV = P / vw;
    for (int e = threadIdx.x; e < nb * PV; e += blockDim.x) {
      const int br = e / PV, p = vw * (e - br * PV), b = b0 + br;
      const float w = a.pre ? a.pre[(long long)b * C + c] : 1.f;
      float xq[4], aq[4] = {0.f, 0.f, 0.f, 0.f}, yv[4];
      cg_ldv(a.x + cg_row_base(a.xv, b, c) + p, vw, xq);
      if (a.add) cg_ldv(a.add + cg_row_base(a.av, b, c) + p, vw, aq);
      const unsigned long long idx = ((unsigned long long)b * C + c) * P + p;      // element index: its group of four shares 64 random bits
      const unsigned long long bits = drop ? cg_drop_bits(seed, a.salt, idx >> 2) : 0ull;
      const int j0 = (int)(idx & 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) yv[j] = j < vw ? cg_norm_act_y(a, af, alpha, xq[j], w, drop ? cg_drop_pick(bits, j0 + j, a.drop_p) : 1.f, aq[j]) : 0.f;
      cg_stv(a.y + cg_row_base(a.yv, b, c) + p, vw, yv);
      if (a.ystats) {
        ys += ((double)yv[0] + (double)yv[1]) + ((double)yv[2] + (double)yv[3]);
        yq += ((double)yv[0] * (double)yv[0] + (double)yv[1] * (double)yv[1]) + ((double)yv[2] * (double)yv[2] + (double)yv[3] * (double)yv[3]);
      }
    }
  } else {
    CG_CHUNK_LOOP(nb, P, e) {
      CG_CHUNK_ROW(e, P, b0)
      CG_POS(a.xv, p)
      const float w = a.pre ? a.pre[(long long)b * C + c] : 1.f;
      const float keep = drop ? cg_drop_scale(a.drop_p, seed, a.salt, ((unsigned long long)b * C + c) * P + p) : 1.f;
      const float ad = a.add ? a.add[cg_row_base(a.av, b, c) + CG_OFF(a.av)] : 0.f;
      const float u = cg_norm_act_y(a, af, alpha, a.x[cg_row_base(a.xv, b, c) + CG_OFF(a.xv)], w, keep, ad);
      a.y[cg_row_base(a.yv, b, c) + CG_OFF(a.yv)] = u;
      ys += (double)u; yq += (double)u * (double)u;
    }
  }
  if (a.ystats) {      // uniform per problem: every thread of the workgroup takes this branch
    ys = cg_block_sum(ys, red);
    yq = cg_block_sum(yq, red + 16);
    if (threadIdx.x == 0) {
      double* rep = a.ystats + (long long)(blockIdx.y % CG_STAT_REPLICAS) * 2 * C;
      atomicAdd(&rep[2 * c], ys); atomicAdd(&rep[2 * c + 1], yq);
    }
  }
}

// gradient at the affine output (after PReLU and dropout are undone) from loaded values; also returns the gated
// input v, the pre-activation u and the gradient gu in front of the PReLU
__device__ __forceinline__ float cg_norm_act_gh_val(const CgNormAct& a, const CgChanAffine& af, float alpha, float xval, float w,
                                                    float keep, float ad, float g, float& v, float& u, float& gu) {
  v = xval * w;
  u = ((v - af.mean) * af.scale + af.shift) * keep;
  if (a.add && !a.add_post) u += ad;
  gu = a.alpha ? (u > 0.f ? g : alpha * g) : g;
  return gu * keep;
}

__device__ __forceinline__ float cg_norm_act_gh(const CgNormAct& a, const CgChanAffine& af, float alpha,
                                                unsigned long long seed, int b, int c, long long C, int P,
                                                int p, int i2_, int i3_, float& w, float& v, float& u, float& gu, float& g) {
  w = a.pre ? a.pre[(long long)b * C + c] : 1.f;
  const float keep = a.drop_p > 0.f ? cg_drop_scale(a.drop_p, seed, a.salt, ((unsigned long long)b * C + c) * P + p) : 1.f;
  const float ad = (a.add && !a.add_post) ? a.add[cg_row_base(a.av, b, c) + CG_OFF(a.av)] : 0.f;
  g = a.dy[cg_row_base(a.dyv, b, c) + CG_OFF(a.dyv)];
  return cg_norm_act_gh_val(a, af, alpha, a.x[cg_row_base(a.xv, b, c) + CG_OFF(a.xv)], w, keep, ad, g, v, u, gu);
}

// a group of vw (4 or 2) consecutive elements: loads for the vector path of both backward passes
struct CgNaQuad { float x[4], ad[4], g[4], keep[4]; float w; };
__device__ __forceinline__ CgNaQuad cg_norm_act_quad(const CgNormAct& a, unsigned long long seed, int b, int c, long long C, int P, int p, int vw) {
  CgNaQuad q;
  q.w = a.pre ? a.pre[(long long)b * C + c] : 1.f;
  cg_ldv(a.x + cg_row_base(a.xv, b, c) + p, vw, q.x);
  cg_ldv(a.dy + cg_row_base(a.dyv, b, c) + p, vw, q.g);
#pragma unroll
  for (int j = 0; j < 4; ++j) q.ad[j] = 0.f;
  if (a.add && !a.add_post) cg_ldv(a.add + cg_row_base(a.av, b, c) + p, vw, q.ad);
  if (a.drop_p > 0.f) {
    const unsigned long long idx = ((unsigned long long)b * C + c) * P + p;
    const unsigned long long bits = cg_drop_bits(seed, a.salt, idx >> 2);
    const int j0 = (int)(idx & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) q.keep[j] = j < vw ? cg_drop_pick(bits, j0 + j, a.drop_p) : 0.f;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) q.keep[j] = 1.f;
  }
  return q;
}

// pass 1 of backward: per-channel sums of g and g*xhat (f64), d alpha
__global__ void cg_norm_act_bwd_reduce_kernel(CgNormActBatch batch) {
  __shared__ double red[48];
  const CgNormAct& a = batch.a[blockIdx.z];
  const int rb = batch.rb[blockIdx.z];
  const int c = blockIdx.x, b0 = blockIdx.y * rb;
  if (c >= a.xv.n[1] || b0 >= a.xv.n[0]) return;
  const long long C = a.xv.n[1];
  const int P = (int)(a.xv.n[2] * a.xv.n[3]);
  const int nb = min(rb, (int)a.xv.n[0] - b0);
  const CgChanAffine af = cg_chan_affine(a, c, true);
  const float alpha = a.alpha ? a.alpha[a.alpha_n == 1 ? 0 : c] : 1.f;
  const unsigned long long seed = (a.drop_p > 0.f) ? *a.seed : 0ull;
  double s1 = 0.0, s2 = 0.0, sa = 0.0;
  if (batch.vec[blockIdx.z]) {
    const int vw = batch.vec[blockIdx.z], PV = P / vw;
    for (int e = threadIdx.x; e < nb * PV; e += blockDim.x) {
      const int br = e / PV, p = vw * (e - br * PV), b = b0 + br;
      const CgNaQuad q = cg_norm_act_quad(a, seed, b, c, C, P, p, vw);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j >= vw) break;
        float v, u, gu;
        const float gh = cg_norm_act_gh_val(a, af, alpha, q.x[j], q.w, q.keep[j], q.ad[j], q.g[j], v, u, gu);
        s1 += (double)gh;
        s2 += (double)gh * (double)((v - af.mean) * af.rstd);
        if (a.alpha && !(u > 0.f)) sa += (double)q.g[j] * (double)u;
      }
    }
  } else {
    CG_CHUNK_LOOP(nb, P, e) {
      CG_CHUNK_ROW(e, P, b0)
      CG_POS(a.xv, p)
      float w, v, u, gu, g;
      const float gh = cg_norm_act_gh(a, af, alpha, seed, b, c, C, P, p, i2_, i3_, w, v, u, gu, g);
      s1 += (double)gh;
      s2 += (double)gh * (double)((v - af.mean) * af.rstd);
      if (a.alpha && !(u > 0.f)) sa += (double)g * (double)u;
    }
  }
  s1 = cg_block_sum(s1, red);
  s2 = cg_block_sum(s2, red + 16);
  sa = cg_block_sum(sa, red + 32);
  if (threadIdx.x == 0) {
    atomicAdd(&a.red[2 * c], s1);
    atomicAdd(&a.red[2 * c + 1], s2);
    if (a.alpha) atomicAdd(&a.red[2 * C + (a.alpha_n == 1 ? ((c + 5 * (int)blockIdx.y) & (CG_ALPHA_SLOTS - 1)) : c)], sa);
  }
}

// pass 2 of backward: dx, d add, d pre (row sums), and the per-channel parameter gradients
__global__ void cg_norm_act_bwd_apply_kernel(CgNormActBatch batch) {
  __shared__ double red[16];
  const CgNormAct& a = batch.a[blockIdx.z];
  const int rb = batch.rb[blockIdx.z];                // rb == 1 whenever the per-row gate gradient is requested
  const int c = blockIdx.x, b0 = blockIdx.y * rb;
  if (c >= a.xv.n[1] || b0 >= a.xv.n[0]) return;
  const long long C = a.xv.n[1];
  const int P = (int)(a.xv.n[2] * a.xv.n[3]);
  const int nb = min(rb, (int)a.xv.n[0] - b0);
  const CgChanAffine af = cg_chan_affine(a, c, true);
  const float alpha = a.alpha ? a.alpha[a.alpha_n == 1 ? 0 : c] : 1.f;
  const unsigned long long seed = (a.drop_p > 0.f) ? *a.seed : 0ull;
  float m1 = 0.f, m2 = 0.f;
  if (a.bn_mode == 1) {
    const double cnt = (double)a.xv.n[0] * (double)P;
    m1 = (float)(a.red[2 * c] / cnt);
    m2 = (float)(a.red[2 * c + 1] / cnt);
  }
  double sp = 0.0;
  if (batch.vec[blockIdx.z]) {
    const int vw = batch.vec[blockIdx.z], PV = P / vw;
    for (int e = threadIdx.x; e < nb * PV; e += blockDim.x) {
      const int br = e / PV, p = vw * (e - br * PV), b = b0 + br;
      const CgNaQuad q = cg_norm_act_quad(a, seed, b, c, C, P, p, vw);
      float dxq[4] = {0.f, 0.f, 0.f, 0.f}, daq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j >= vw) break;
        float v, u, gu;
        const float gh = cg_norm_act_gh_val(a, af, alpha, q.x[j], q.w, q.keep[j], q.ad[j], q.g[j], v, u, gu);
        float gv;
        if (a.bn_mode == 1) gv = af.scale * (gh - m1 - (v - af.mean) * af.rstd * m2);
        else gv = gh * af.scale;
        dxq[j] = gv * q.w; daq[j] = gu;
        if (a.dpre) sp += (double)gv * (double)q.x[j];
      }
      if (a.dx) cg_stv(a.dx + cg_row_base(a.dxv, b, c) + p, vw, dxq);
      if (a.dadd) cg_stv(a.dadd + cg_row_base(a.dav, b, c) + p, vw, daq);
    }
  } else {
    CG_CHUNK_LOOP(nb, P, e) {
      CG_CHUNK_ROW(e, P, b0)
      CG_POS(a.xv, p)
      float w, v, u, gu, g;
      const float gh = cg_norm_act_gh(a, af, alpha, seed, b, c, C, P, p, i2_, i3_, w, v, u, gu, g);
      float gv;
      if (a.bn_mode == 1) gv = af.scale * (gh - m1 - (v - af.mean) * af.rstd * m2);
      else gv = gh * af.scale;
      if (a.dx) a.dx[cg_row_base(a.dxv, b, c) + CG_OFF(a.dxv)] = gv * w;
      if (a.dadd) a.dadd[cg_row_base(a.dav, b, c) + CG_OFF(a.dav)] = gu;
      if (a.dpre) sp += (double)gv * (double)a.x[cg_row_base(a.xv, b, c) + CG_OFF(a.xv)];
    }
  }
  if (a.dpre) {
    sp = cg_block_sum(sp, red);
    if (threadIdx.x == 0) a.dpre[(long long)b0 * C + c] = (float)sp;
  }
  if (b0 == 0 && threadIdx.x == 0) {
    if (a.bn_mode != 0) {
      if (a.dgamma) a.dgamma[c] = (float)a.red[2 * c + 1];
      if (a.dbeta) a.dbeta[c] = (float)a.red[2 * c];
    }
    if (a.alpha && a.dalpha) {
      if (a.alpha_n == 1) { if (c == 0) a.dalpha[0] = (float)cg_alpha_sum(a.red + 2 * C); }
      else a.dalpha[c] = (float)a.red[2 * C + c];
    }
  }
}

static int cg_norm_act_check(const CgNormAct* a, bool fwd) {
  if (!a || !a->x) return CG_EARG;
  const long long P = a->xv.n[2] * a->xv.n[3];
  if (a->xv.n[0] <= 0 || a->xv.n[1] <= 0 || P <= 0 || a->xv.n[0] > 65535) return CG_ESHAPE;
  if (a->bn_mode != 0 && (!a->gamma || !a->beta || !a->save_mean || !a->save_rstd)) return CG_EARG;
  if (fwd && a->bn_mode == 1 && !a->stats) return CG_EARG;
  if (fwd && a->bn_mode == 2 && (!a->running_mean || !a->running_var)) return CG_EARG;
  if (a->drop_p > 0.f && !a->seed) return CG_EARG;
  if (a->drop_p < 0.f || a->drop_p >= 1.f) return CG_EARG;
  return CG_OK;
}

static dim3 cg_row_block(const CgView4& v) { return dim3(v.n[2] * v.n[3] <= 256 ? 64 : 256); }
static dim3 cg_chunk_block(const CgView4& v, int rb) { return dim3(v.n[2] * v.n[3] * rb <= 256 ? 64 : 256); }

// kind: 0 forward, 1 backward reduce, 2 backward apply
static int cg_norm_act_launch(const CgNormAct* arr, const int* sel, int n, int kind, hipStream_t stream) {
  CgNormActBatch batch;
  batch.n = n; batch.pad = 0;
  long long gx = 1, gy = 1, big = 0;
  for (int i = 0; i < n; ++i) {
    const CgNormAct& a = arr[sel ? sel[i] : i];
    int rb = cg_rows_per_block(a.xv);
    if (kind == 2 && a.dpre) rb = 1;      // the gate gradient is one sum per (batch, channel) row
    batch.a[i] = a; batch.rb[i] = rb; batch.vec[i] = cg_norm_act_vec(a, kind);
    gx = gx > a.xv.n[1] ? gx : a.xv.n[1];
    const long long chunks = (a.xv.n[0] + rb - 1) / rb;
    gy = gy > chunks ? gy : chunks;
    if (a.xv.n[2] * a.xv.n[3] * rb > 256) big = 1;
  }
  if (gy > 65535) return CG_ESHAPE;
  dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)n), block(big ? 256 : 64);
  if (kind == 0) hipLaunchKernelGGL(cg_norm_act_fwd_kernel, grid, block, 0, stream, batch);
  else if (kind == 1) hipLaunchKernelGGL(cg_norm_act_bwd_reduce_kernel, grid, block, 0, stream, batch);
  else hipLaunchKernelGGL(cg_norm_act_bwd_apply_kernel, grid, block, 0, stream, batch);
  return cg_launch_status();
}

// include/cistgcn_hip.h : cg_norm_act_fwd_many / cg_norm_act_bwd_many (up to CG_ROW_MAX_BATCH row problems per launch)
extern "C" int cg_norm_act_fwd_many(const CgNormAct* arr, int n, void* stream_) {
  if (!arr || n <= 0 || n > CG_ROW_MAX_BATCH) return CG_EARG;
  for (int i = 0; i < n; ++i) {
    int st = cg_norm_act_check(&arr[i], true);
    if (st != CG_OK) return st;
    if (!arr[i].y) return CG_EARG;
  }
  return cg_norm_act_launch(arr, nullptr, n, 0, (hipStream_t)stream_);
}

// `red` of every problem must be zero on entry (slices of the per-step zeroed arena).
extern "C" int cg_norm_act_bwd_many(const CgNormAct* arr, const int* need_reduce, int n, void* stream_) {
  if (!arr || !need_reduce || n <= 0 || n > CG_ROW_MAX_BATCH) return CG_EARG;
  int sel[CG_ROW_MAX_BATCH], nr = 0;
  for (int i = 0; i < n; ++i) {
    int st = cg_norm_act_check(&arr[i], false);
    if (st != CG_OK) return st;
    if (!arr[i].dy) return CG_EARG;
    if ((arr[i].bn_mode != 0 || arr[i].alpha) && !arr[i].red) return CG_EARG;
    if (need_reduce[i]) sel[nr++] = i;
  }
  if (nr > 0) {
    int st = cg_norm_act_launch(arr, sel, nr, 1, (hipStream_t)stream_);
    if (st != CG_OK) return st;
  }
  return cg_norm_act_launch(arr, nullptr, n, 2, (hipStream_t)stream_);
}

// per-channel parameter gradients from `red` alone (the apply pass writes them otherwise)
__global__ void cg_norm_act_params_kernel(CgNormActBatch batch) {
  const CgNormAct& a = batch.a[blockIdx.x];
  const long long C = a.xv.n[1];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    if (a.bn_mode != 0) {
      if (a.dgamma) a.dgamma[c] = (float)a.red[2 * c + 1];
      if (a.dbeta) a.dbeta[c] = (float)a.red[2 * c];
    }
    if (a.alpha && a.dalpha) {
      if (a.alpha_n == 1) { if (c == 0) a.dalpha[0] = (float)cg_alpha_sum(a.red + 2 * C); }
      else a.dalpha[c] = (float)a.red[2 * C + c];
    }
  }
}

// include/cistgcn_hip.h : cg_norm_act_bwd_reduce_many - pass 1 of the backward only (sums of g and g * xhat, slope gradient) plus the
// parameter gradients; for a consumer that applies the BatchNorm / PReLU backward itself while it loads the gradient
// (cg_pointwise_maps_bwd with `yraw`).  dx / dadd / dpre are not written.
extern "C" int cg_norm_act_bwd_reduce_many(const CgNormAct* arr, int n, void* stream_) {
  if (!arr || n <= 0 || n > CG_ROW_MAX_BATCH) return CG_EARG;
  for (int i = 0; i < n; ++i) {
    int st = cg_norm_act_check(&arr[i], false);
    if (st != CG_OK) return st;
    if (!arr[i].dy || !arr[i].red) return CG_EARG;
  }
  int st = cg_norm_act_launch(arr, nullptr, n, 1, (hipStream_t)stream_);
  if (st != CG_OK) return st;
  CgNormActBatch batch;
  batch.n = n; batch.pad = 0;
  for (int i = 0; i < n; ++i) { batch.a[i] = arr[i]; batch.rb[i] = 1; batch.vec[i] = 0; }
  hipLaunchKernelGGL(cg_norm_act_params_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream_, batch);
  return cg_launch_status();
}

// include/cistgcn_hip.h : cg_norm_act_params_many - dgamma / dbeta / dalpha from `red` alone, for a consumer whose producer-side kernel
// has filled `red` already (cg_collapse_rows_bwd / cg_collapse_cols_bwd with in_red): no pass over the tensors at all
extern "C" int cg_norm_act_params_many(const CgNormAct* arr, int n, void* stream_) {
  if (!arr || n <= 0 || n > CG_ROW_MAX_BATCH) return CG_EARG;
  CgNormActBatch batch;
  batch.n = n; batch.pad = 0;
  for (int i = 0; i < n; ++i) {
    if (!arr[i].red) return CG_EARG;
    batch.a[i] = arr[i]; batch.rb[i] = 1; batch.vec[i] = 0;
  }
  hipLaunchKernelGGL(cg_norm_act_params_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream_, batch);
  return cg_launch_status();
}

extern "C" int cg_norm_act_fwd(const CgNormAct* a, void* stream_) { return cg_norm_act_fwd_many(a, 1, stream_); }

extern "C" int cg_norm_act_bwd(const CgNormAct* a, int need_reduce, void* stream_) {
  return cg_norm_act_bwd_many(a, &need_reduce, 1, stream_);
}

// ---------------------------------------------------------------------------------------------
// per-(b,c) reductions over the positions: mean (0), max with first-arg-max (1), sum (2) (ContextLayer :465-467,
// SE squeeze SE.py:17,38, FPN action context CISTGCN.py:76)
// ---------------------------------------------------------------------------------------------
__global__ void cg_reduce_bc_kernel(const float* __restrict__ x, CgView4 xv, int kind, float* __restrict__ out,
                                    int32_t* __restrict__ arg) {
  __shared__ float sv[256];
  __shared__ int si[256];
  __shared__ double red[16];
  const int c = blockIdx.x, b = blockIdx.y;
  const long long C = xv.n[1], P = xv.n[2] * xv.n[3];
  const long long base = cg_row_base(xv, b, c);
  if (kind != 1) {
    double s = 0.0;
    CG_ROW_LOOP(P, p) { CG_POS(xv, p) s += (double)x[base + CG_OFF(xv)]; }
    s = cg_block_sum(s, red);
    if (threadIdx.x == 0) out[(long long)b * C + c] = kind == 0 ? (float)(s / (double)P) : (float)s;
  } else {
    float best = -INFINITY; int bi = 0x7fffffff;
    CG_ROW_LOOP(P, p) {
      CG_POS(xv, p)
      const float v = x[base + CG_OFF(xv)];
      if (v > best || (v == best && (int)p < bi) || bi == 0x7fffffff) { best = v; bi = (int)p; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        const float ov = sv[threadIdx.x + s]; const int oi = si[threadIdx.x + s];
        if (oi != 0x7fffffff && (si[threadIdx.x] == 0x7fffffff || ov > sv[threadIdx.x] ||
                                 (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))) {
          sv[threadIdx.x] = ov; si[threadIdx.x] = oi;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) { out[(long long)b * C + c] = sv[0]; arg[(long long)b * C + c] = si[0]; }
  }
}

extern "C" int cg_reduce_bc(const float* x, const CgView4* xv, int kind, float* out, int32_t* arg, void* stream_) {
  if (!x || !xv || !out || (kind == 1 && !arg)) return CG_EARG;
  if (xv->n[0] <= 0 || xv->n[0] > 65535 || xv->n[1] <= 0 || xv->n[2] * xv->n[3] <= 0) return CG_ESHAPE;
  dim3 grid((unsigned)xv->n[1], (unsigned)xv->n[0]);
  hipLaunchKernelGGL(cg_reduce_bc_kernel, grid, cg_row_block(*xv), 0, (hipStream_t)stream_, x, *xv, kind, out, arg);
  return cg_launch_status();
}

__global__ void cg_reduce_bc_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg, int kind,
                                        float* __restrict__ dx, CgView4 dxv) {
  const int c = blockIdx.x, b = blockIdx.y;
  const long long C = dxv.n[1], P = dxv.n[2] * dxv.n[3];
  const long long base = cg_row_base(dxv, b, c);
  const float g = dout[(long long)b * C + c];
  const int hit = kind == 1 ? arg[(long long)b * C + c] : -1;
  const float gm = g / (float)P;
  CG_ROW_LOOP(P, p) { CG_POS(dxv, p) dx[base + CG_OFF(dxv)] = kind == 0 ? gm : (p == hit ? g : 0.f); }
}

extern "C" int cg_reduce_bc_bwd(const float* dout, const int32_t* arg, int kind, float* dx, const CgView4* dxv, void* stream_) {
  if (!dout || !dx || !dxv || (kind == 1 && !arg)) return CG_EARG;
  if (dxv->n[0] <= 0 || dxv->n[0] > 65535 || dxv->n[1] <= 0) return CG_ESHAPE;
  dim3 grid((unsigned)dxv->n[1], (unsigned)dxv->n[0]);
  hipLaunchKernelGGL(cg_reduce_bc_bwd_kernel, grid, cg_row_block(*dxv), 0, (hipStream_t)stream_, dout, arg, kind, dx, *dxv);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// strided copy / sum of up to three views (cat, halo padding, residual sums, broadcast)
// ---------------------------------------------------------------------------------------------
__global__ void cg_add3_kernel(float* __restrict__ y, CgView4 yv, const float* __restrict__ a, CgView4 av,
                               const float* __restrict__ b_, CgView4 bv, const float* __restrict__ c_, CgView4 cv) {
  const int c = blockIdx.x, b = blockIdx.y;
  const long long P = yv.n[2] * yv.n[3];
  const long long by = cg_row_base(yv, b, c), ba = cg_row_base(av, b, c);
  const long long bb = b_ ? cg_row_base(bv, b, c) : 0, bc = c_ ? cg_row_base(cv, b, c) : 0;
  CG_ROW_LOOP(P, p) {
    CG_POS(yv, p)
    float v = a[ba + CG_OFF(av)];
    if (b_) v += b_[bb + CG_OFF(bv)];
    if (c_) v += c_[bc + CG_OFF(cv)];
    y[by + CG_OFF(yv)] = v;
  }
}

extern "C" int cg_add3(float* y, const CgView4* yv, const float* a, const CgView4* av, const float* b,
                       const CgView4* bv, const float* c, const CgView4* cv, void* stream_) {
  if (!y || !yv || !a || !av) return CG_EARG;
  if (yv->n[0] <= 0 || yv->n[0] > 65535 || yv->n[1] <= 0 || yv->n[2] * yv->n[3] <= 0) return CG_ESHAPE;
  CgView4 zero = *av;
  dim3 grid((unsigned)yv->n[1], (unsigned)yv->n[0]);
  hipLaunchKernelGGL(cg_add3_kernel, grid, cg_row_block(*yv), 0, (hipStream_t)stream_, y, *yv, a, *av,
                     b, b ? *bv : zero, c, c ? *cv : zero);
  return cg_launch_status();
}

// y = sum of up to CG_SUM_MAX same-shape strided tensors in one launch: the gradient of a tensor that feeds several
// consumers (a block input goes to the statistics, the gate/tower maps, both graph stages and the residuals) is
// summed once instead of by one autograd accumulation kernel per extra consumer.
#define CG_SUM_MAX 8
struct CgSumArgs { float* y; CgView4 yv; int n; int vec; const float* a[CG_SUM_MAX]; CgView4 av[CG_SUM_MAX]; };

__global__ void cg_sum_many_kernel(CgSumArgs s) {
  const int c = blockIdx.x, b = blockIdx.y;
  const int P = (int)(s.yv.n[2] * s.yv.n[3]);
  if (s.vec) {                                   // contiguous rows everywhere, aligned to 4 * vec bytes (vec = 4 or 2)
    const int vw = s.vec;
    float* yr = s.y + cg_row_base(s.yv, b, c);
    for (int p = vw * threadIdx.x; p < P; p += vw * blockDim.x) {
      float v[4], w[4];
      cg_ldv(s.a[0] + cg_row_base(s.av[0], b, c) + p, vw, v);
      for (int i = 1; i < s.n; ++i) {
        cg_ldv(s.a[i] + cg_row_base(s.av[i], b, c) + p, vw, w);
        v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
      }
      cg_stv(yr + p, vw, v);
    }
    return;
  }
  CG_ROW_LOOP(P, p) {
    CG_POS(s.yv, p)
    float v = s.a[0][cg_row_base(s.av[0], b, c) + CG_OFF(s.av[0])];
    for (int i = 1; i < s.n; ++i) v += s.a[i][cg_row_base(s.av[i], b, c) + CG_OFF(s.av[i])];
    s.y[cg_row_base(s.yv, b, c) + CG_OFF(s.yv)] = v;
  }
}

struct CgSumItem { const float* a; CgView4 av; };
// include/cistgcn_hip.h : cg_sum_many
extern "C" int cg_sum_many(float* y, const CgView4* yv, const CgSumItem* items, int n, void* stream_) {
  if (!y || !yv || !items || n <= 0 || n > CG_SUM_MAX) return CG_EARG;
  if (yv->n[0] <= 0 || yv->n[0] > 65535 || yv->n[1] <= 0 || yv->n[2] * yv->n[3] <= 0) return CG_ESHAPE;
  CgSumArgs s;
  s.y = y; s.yv = *yv; s.n = n;
  int vec = cg_view_vec(y, *yv);
  for (int i = 0; i < CG_SUM_MAX; ++i) {
    s.a[i] = i < n ? items[i].a : nullptr;
    s.av[i] = i < n ? items[i].av : *yv;
    if (i < n) {
      if (!items[i].a) return CG_EARG;
      for (int k = 0; k < 4; ++k)
        if (items[i].av.n[k] != yv->n[k]) return CG_ESHAPE;
      { const int w = cg_view_vec(items[i].a, items[i].av); vec = w < vec ? w : vec; }
    }
  }
  s.vec = vec;
  dim3 grid((unsigned)yv->n[1], (unsigned)yv->n[0]);
  hipLaunchKernelGGL(cg_sum_many_kernel, grid, cg_row_block(*yv), 0, (hipStream_t)stream_, s);
  return cg_launch_status();
}

// up to four strided copies in one launch (the slices of a channel concatenation)
struct CgCopyItem { float* y; CgView4 yv; const float* a; CgView4 av; };
struct CgCopyBatch { int n; int pad; CgCopyItem it[4]; };

__global__ void cg_copy_many_kernel(CgCopyBatch batch) {
  const CgCopyItem& it = batch.it[blockIdx.z];
  const int c = blockIdx.x, b = blockIdx.y;
  if (c >= it.yv.n[1] || b >= it.yv.n[0]) return;
  const int P = (int)(it.yv.n[2] * it.yv.n[3]);
  const long long by = cg_row_base(it.yv, b, c), ba = cg_row_base(it.av, b, c);
  CG_ROW_LOOP(P, p) {
    CG_POS(it.yv, p)
    it.y[by + CG_OFF(it.yv)] = it.a[ba + CG_OFF(it.av)];
  }
}

extern "C" int cg_copy_many(const CgCopyItem* items, int n, void* stream_) {
  if (!items || n <= 0 || n > 4) return CG_EARG;
  CgCopyBatch batch;
  batch.n = n; batch.pad = 0;
  long long gx = 1, gy = 1, big = 0;
  for (int i = 0; i < n; ++i) {
    if (!items[i].y || !items[i].a) return CG_EARG;
    const CgView4& v = items[i].yv;
    if (v.n[0] <= 0 || v.n[1] <= 0 || v.n[2] * v.n[3] <= 0 || v.n[0] > 65535) return CG_ESHAPE;
    batch.it[i] = items[i];
    gx = gx > v.n[1] ? gx : v.n[1];
    gy = gy > v.n[0] ? gy : v.n[0];
    if (v.n[2] * v.n[3] > 256) big = 1;
  }
  hipLaunchKernelGGL(cg_copy_many_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)n), dim3(big ? 256 : 64), 0, (hipStream_t)stream_, batch);
  return cg_launch_status();
}

// Zero fill as a KERNEL, not hipMemsetAsync: inside a captured HIP graph the memset node was observed not to be ordered
// reliably against the kernels around it once the process had allocated and freed other device memory between replays
// (zero halos of the dilated convolutions read NaN left by unrelated tensors; tools/diag_replay_stability.py, round 1).
__global__ void cg_zero_kernel(uint4* __restrict__ p16, long long n16, unsigned char* __restrict__ tail, int ntail) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long k = i; k < n16; k += stride) p16[k] = make_uint4(0u, 0u, 0u, 0u);
  if (i < ntail) tail[i] = 0;
}

int cg_zero_fill(void* p, long long bytes, hipStream_t stream) {
  if (!p) return CG_EARG;
  if (bytes < 0) return CG_ESHAPE;
  if (bytes == 0) return CG_OK;
  unsigned char* b = static_cast<unsigned char*>(p);
  // unaligned head bytes, 16-byte aligned body, tail bytes
  const uintptr_t addr = reinterpret_cast<uintptr_t>(b);
  long long head = (16 - (long long)(addr & 15)) & 15;
  if (head > bytes) head = bytes;
  const long long body = ((bytes - head) / 16) * 16;
  const long long tail = bytes - head - body;
  if (head > 0) hipLaunchKernelGGL(cg_zero_kernel, dim3(1), dim3(64), 0, stream, (uint4*)nullptr, 0LL, b, (int)head);
  if (body > 0 || tail > 0) {
    const long long n16 = body / 16;
    long long blocks = (n16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cg_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<uint4*>(b + head), n16,
                       b + head + body, (int)tail);
  }
  return cg_launch_status();
}

extern "C" int cg_zero(void* p, long long bytes, void* stream_) { return cg_zero_fill(p, bytes, (hipStream_t)stream_); }
