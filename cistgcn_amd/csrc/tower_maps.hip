// Stacked pointwise maps of one input: the first convolutions of the four Map2Adj towers of a DSTD_GC block (reference:
// Map2Adj.time_compress[0] / joint_compress[0], CISTGCN.py:138-163, applied to the normalised block input by :183-186) read the
// same (B,C,T,V) tensor.  As separate contractions the input travelled once per map forward and once per map and gradient
// backward (the K-reduction weight gradient alone read it four times); here
//   forward   y_i = W_i x for all maps from ONE staged tile of x (+ f64 channel sums of every y_i for the BatchNorm behind it)
//   backward  dx = sum_i W_i^T dy_i and dW_i += dy_i x^T from ONE staged tile of x and of every dy_i
// Persistent 512-thread workgroups walk [rows][PT positions] tiles (rows * PT = 8192: sixteen staging registers per thread and
// tensor, next tile's loads in flight during the matrix work); products on v_mfma_f32_16x16x4_f32 with positions as the rows of
// the result tile, so a lane ends with four consecutive positions of one channel (float4 stores); the maps' weights are stacked in
// LDS, every map padded to a multiple of 16 rows.
#include "cg_common.h"
#include "cg_phase.h"
#include "tower_maps.h"
#include <type_traits>
#include <cstdlib>

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

#define CG_PWM_THREADS 512     // eight waves per workgroup: a tile is rows * PT = 8192 elements, sixteen staging registers per thread
#define CG_PWM_REPLICAS 16

struct CgPwGeom {
  int CinM, MM, WS, NT, CT;            // padded input channels, stacked rows, weight row stride, 16-row tiles of the stack / of Cin
  int PT, PS, lgq, vw;                 // tile width, LDS row stride, log2(PT / 4), vector width of the global rows (4, or 2 when P % 4 == 2)
  int tps, total, per;                 // tiles per sample, tiles, tiles per workgroup
  int tile_map[CG_PWM_MAXROWS / 16], tile_row0[CG_PWM_MAXROWS / 16], row_base[CG_PWM_MAXN];
};
struct CgPwArgs { CgPwMaps t; CgPwGeom g; };

// The tile geometry as the kernels see it.  NT_ / CT_ > 0: 16-row tiles of the stacked maps / of the input channels known at compile time
// (one instantiation per shape of the shipped configurations): every size below is then a constant, the matrix-core loops unroll with
// constant LDS offsets and independent accumulator chains.  NT_ == 0: the sizes the host computed (any shape the entry point accepts).
// BWD: the backward kernel sizes its tile by the larger of the two stacks.
template <int NT_, int CT_, bool BWD>
struct CgPwK {
  const CgPwGeom& g;
  static constexpr int kBig = BWD ? (NT_ > CT_ ? NT_ : CT_) : CT_;
  static_assert((kBig & (kBig - 1)) == 0, "the host's tile width (cg_pwm_geometry) equals 8192 / rows only for power-of-two stacks");
  static constexpr int kPT = NT_ ? (16 * CG_PWM_THREADS / (16 * kBig) > 256 ? 256 : 16 * CG_PWM_THREADS / (16 * kBig)) : 0;
  static constexpr int kLgq = kPT == 256 ? 6 : kPT == 128 ? 5 : kPT == 64 ? 4 : kPT == 32 ? 3 : kPT == 16 ? 2 : 0;
  __device__ __forceinline__ int CinM() const { return NT_ ? 16 * CT_ : g.CinM; }
  __device__ __forceinline__ int MM() const { return NT_ ? 16 * NT_ : g.MM; }
  __device__ __forceinline__ int WS() const { return NT_ ? 16 * CT_ + 4 : g.WS; }
  __device__ __forceinline__ int NT() const { return NT_ ? NT_ : g.NT; }
  __device__ __forceinline__ int CT() const { return NT_ ? CT_ : g.CT; }
  __device__ __forceinline__ int PT() const { return NT_ ? kPT : g.PT; }
  __device__ __forceinline__ int PS() const { return NT_ ? kPT + 4 : g.PS; }
  __device__ __forceinline__ int lgq() const { return NT_ ? kLgq : g.lgq; }
};

// stacked weights -> sW [MM][WS]; rows beyond a map's M_i and columns beyond Cin are zero
// stacked weights [MM][WS] into LDS, zero in the padding.  Map by map (the map index is a scalar): looking the map of a stacked row up
// per element - kernel-argument tables indexed by a lane value, three dependent loads - made this prologue 30 k cycles
template <typename KT>
__device__ __forceinline__ void cg_pwm_weights(const CgPwArgs& a, const KT& G, float* sW) {
  const CgPwMaps& t = a.t; const CgPwGeom& g = a.g;
  for (int e = threadIdx.x; e < G.MM() * G.WS(); e += CG_PWM_THREADS) sW[e] = 0.f;
  __syncthreads();
  for (int i = 0; i < t.n; ++i) {
    const float* __restrict__ W = t.W[i];
    const int n = t.M[i] * t.Cin, r0 = g.row_base[i];
#pragma unroll 4
    for (int e = threadIdx.x; e < n; e += CG_PWM_THREADS) {
      const int m = e / t.Cin, c = e - m * t.Cin;
      sW[(r0 + m) * G.WS() + c] = W[e];
    }
  }
}

template <typename KT, typename SRC>
__device__ __forceinline__ void cg_pwm_fetch(const CgPwGeom& g, const KT& G, int np, float buf[16], SRC src) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = (int)threadIdx.x + CG_PWM_THREADS * r, row = e >> G.lgq(), pp = 4 * (e & ((1 << G.lgq()) - 1));
    const float* p = pp < np ? src(row) : nullptr;         // nullptr: row outside the tensor
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p != nullptr) {
      if (g.vw == 4) v = *reinterpret_cast<const float4*>(p + pp);
      else {                                               // rows aligned to 8 bytes (P % 4 == 2): pairs; the last quad of a row may be half
        const float2 lo = *reinterpret_cast<const float2*>(p + pp);
        v.x = lo.x; v.y = lo.y;
        if (pp + 2 < np) { const float2 hi = *reinterpret_cast<const float2*>(p + pp + 2); v.z = hi.x; v.w = hi.y; }
      }
    }
    buf[4 * r] = v.x; buf[4 * r + 1] = v.y; buf[4 * r + 2] = v.z; buf[4 * r + 3] = v.w;
  }
}
// four consecutive result positions pq .. pq + 3 (pq % 4 == 0) of a row with np valid positions
__device__ __forceinline__ void cg_pwm_store_quad(const CgPwGeom& g, float* row, int pq, int np, const cg_f32x4& c) {
  if (pq >= np) return;
  if (g.vw == 4) { *reinterpret_cast<float4*>(row + pq) = make_float4(c[0], c[1], c[2], c[3]); return; }
  *reinterpret_cast<float2*>(row + pq) = make_float2(c[0], c[1]);
  if (pq + 2 < np) *reinterpret_cast<float2*>(row + pq + 2) = make_float2(c[2], c[3]);
}
// the same fetch from row pointers resolved ONCE per kernel (slot r of a thread is the same tile row in every tile): base[r] is the row
// at sample 0 / position 0 or nullptr, off[r] the element offset of this tile's sample and first position
template <typename KT>
__device__ __forceinline__ void cg_pwm_fetch_rows(const CgPwGeom& g, const KT& G, int np, float buf[16], const float* const base[4], const long long off[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = (int)threadIdx.x + CG_PWM_THREADS * r, pp = 4 * (e & ((1 << G.lgq()) - 1));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    // (a branch-free form - every lane loads, rows outside the tensor from one stand-in address - was 60 us SLOWER per launch at the
    // headline shape: half of the x slots are such rows, and all workgroups then read the same 16 bytes)
    if (base[r] != nullptr && pp < np) {
      const float* p = base[r] + off[r] + pp;
      if (g.vw == 4) v = *reinterpret_cast<const float4*>(p);
      else {
        const float2 lo = *reinterpret_cast<const float2*>(p);
        v.x = lo.x; v.y = lo.y;
        if (pp + 2 < np) { const float2 hi = *reinterpret_cast<const float2*>(p + 2); v.z = hi.x; v.w = hi.y; }
      }
    }
    buf[4 * r] = v.x; buf[4 * r + 1] = v.y; buf[4 * r + 2] = v.z; buf[4 * r + 3] = v.w;
  }
}
template <typename KT>
__device__ __forceinline__ void cg_pwm_commit(const KT& G, int rows, const float buf[16], float* img) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = (int)threadIdx.x + CG_PWM_THREADS * r, row = e >> G.lgq(), pp = 4 * (e & ((1 << G.lgq()) - 1));
    if (row < rows) *reinterpret_cast<float4*>(img + row * G.PS() + pp) = make_float4(buf[4 * r], buf[4 * r + 1], buf[4 * r + 2], buf[4 * r + 3]);
  }
}

// ======================================================================================================================
// forward
// ======================================================================================================================
template <int NT_, int CT_>
__global__ __launch_bounds__(CG_PWM_THREADS) void cg_pwm_fwd_kernel(CgPwArgs a) {
  const CgPwMaps& t = a.t; const CgPwGeom& g = a.g;
  const CgPwK<NT_, CT_, false> G{g};
  float* sX = reinterpret_cast<float*>(cg_dyn_lds);              // [CinM][PS]
  float* sW = sX + G.CinM() * G.PS();                                 // [MM][WS]
  double* sStat = reinterpret_cast<double*>(sW + G.MM() * G.WS() + ((G.CinM() * G.PS() + G.MM() * G.WS()) & 1));      // [MM][2]
  float* sBias = reinterpret_cast<float*>(sStat + 2 * G.MM());                                                // [MM]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4;
  const int lid0 = blockIdx.x * g.per, lid1 = min(g.total, lid0 + g.per);
  if (lid0 >= g.total) return;
  float xbuf[16];
  auto xsrc = [&](int lid) {
    const int b = lid / g.tps, p0 = (lid - b * g.tps) * G.PT();
    const float* base = t.x + (long long)b * t.Cin * t.P + p0;
    cg_pwm_fetch(g, G, min(G.PT(), t.P - p0), xbuf, [&](int row) { return row < t.Cin ? base + (long long)row * t.P : nullptr; });
  };
  xsrc(lid0);
  cg_pwm_weights(a, G, sW);
  for (int e = tid; e < G.CinM() * G.PS(); e += CG_PWM_THREADS) sX[e] = 0.f;
  const bool stats = t.stats[0] != nullptr;
  if (stats) for (int e = tid; e < 2 * G.MM(); e += CG_PWM_THREADS) sStat[e] = 0.0;
  for (int r = tid; r < G.MM(); r += CG_PWM_THREADS) {
    const int tile = r >> 4, i = g.tile_map[tile], m = g.tile_row0[tile] + (r & 15);
    sBias[r] = (m < t.M[i] && t.bias[i]) ? t.bias[i][m] : 0.f;
  }
  // Known geometry with 8 % NT == 0: every task of a wave has the same row tile nt = wave % NT, so a lane meets ONE output channel in all
  // its tasks and tiles: its channel sums stay in two f64 registers and reach LDS once (an LDS atomic costs ~1000 cycles under load,
  // the run-time form pays two per task and tile).
  constexpr bool kKeepSums = NT_ > 0 && (CG_PWM_THREADS / 64) % (NT_ ? NT_ : 1) == 0;
  double keep1 = 0.0, keep2 = 0.0;
  for (int lid = lid0; lid < lid1; ++lid) {
    const int b = lid / g.tps, p0 = (lid - b * g.tps) * G.PT(), np = min(G.PT(), t.P - p0);
    __syncthreads();
    cg_pwm_commit(G, t.Cin, xbuf, sX);
    __syncthreads();
    if (lid + 1 < lid1) xsrc(lid + 1);
    for (int w = wave; w < (G.PT() / 32) * G.NT(); w += CG_PWM_THREADS / 64) {
      const int pg = w / G.NT(), nt = w - pg * G.NT(), n0 = 32 * pg, n1 = n0 + 16;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      const float* wp = cg_tfrag_ptr<0>(sW + 16 * nt * G.WS(), G.WS(), l15, slot);
      const float* xp0 = cg_tfrag_ptr<1>(sX + n0, G.PS(), l15, slot);
      const float* xp1 = cg_tfrag_ptr<1>(sX + n1, G.PS(), l15, slot);
      for (int k0 = 0; k0 < G.CinM(); k0 += 16) {
        float wv[4], x0v[4], x1v[4];
        cg_tfrag<0>(wp, G.WS(), k0, wv); cg_tfrag<1>(xp0, G.PS(), k0, x0v); cg_tfrag<1>(xp1, G.PS(), k0, x1v);
#pragma unroll
        for (int s = 0; s < 4; ++s) {                     // C[position][output channel]
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0v[s], wv[s], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1v[s], wv[s], c1, 0, 0, 0);
        }
      }
      const int i = g.tile_map[nt], m = g.tile_row0[nt] + l15;
      if (m < t.M[i]) {
        float s1 = 0.f, s2 = 0.f;
        const float bv = sBias[16 * nt + l15];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pq = (h ? n1 : n0) + 4 * slot;
          cg_f32x4 c = h ? c1 : c0;
          c[0] += bv; c[1] += bv; c[2] += bv; c[3] += bv;
          if (pq < np) {
            cg_pwm_store_quad(g, t.y[i] + ((long long)b * t.M[i] + m) * t.P + p0, pq, np, c);
            if (pq + 2 >= np) { c[2] = 0.f; c[3] = 0.f; }        // half quad at the end of a row (P % 4 == 2)
            s1 += (c[0] + c[1]) + (c[2] + c[3]); s2 += (c[0] * c[0] + c[1] * c[1]) + (c[2] * c[2] + c[3] * c[3]);
          }
        }
        if (stats) {
          if (kKeepSums) { keep1 += (double)s1; keep2 += (double)s2; }       // one channel per lane over every task and tile of this wave
          else { atomicAdd(&sStat[2 * (16 * nt + l15)], (double)s1); atomicAdd(&sStat[2 * (16 * nt + l15) + 1], (double)s2); }
        }
      }
    }
  }
  if (stats && kKeepSums) {
    const int nt = wave % (NT_ ? NT_ : 1), i = g.tile_map[nt], m = g.tile_row0[nt] + l15;
    if (m < t.M[i]) { atomicAdd(&sStat[2 * (16 * nt + l15)], keep1); atomicAdd(&sStat[2 * (16 * nt + l15) + 1], keep2); }
  }
  if (stats) {
    __syncthreads();
    for (int e = tid; e < 2 * G.MM(); e += CG_PWM_THREADS) {
      const int r = e >> 1, tile = r >> 4, i = g.tile_map[tile], m = g.tile_row0[tile] + (r & 15);
      if (m < t.M[i]) atomicAdd(&t.stats[i][((long long)(blockIdx.x % CG_STAT_REPLICAS) * t.M[i] + m) * 2 + (e & 1)], sStat[e]);
    }
  }
}

// ======================================================================================================================
// backward
// ======================================================================================================================
// Diagnostic build only (-DCG_TAIL_STAMPS, tools/stamps_planes.py --build): shader-clock stamps of thread 0 at the phase boundaries of
// the backward kernel, read by tools/stamps_pwm.py; the shipped library has no stamp.
#ifdef CG_TAIL_STAMPS
__device__ unsigned long long* cg_pwm_stamp_buf = nullptr;
extern "C" int cg_pwm_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(cg_pwm_stamp_buf), &p, sizeof(p)); }
#define CG_PSTAMP()                                                                                     \
  do {                                                                                                  \
    if (threadIdx.x == 0 && cg_pwm_stamp_buf && nst < 255) cg_pwm_stamp_buf[blockIdx.x * 256 + (++nst)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define CG_PSTAMP_END() do { if (threadIdx.x == 0 && cg_pwm_stamp_buf) cg_pwm_stamp_buf[blockIdx.x * 256] = nst; } while (0)
#else
#define CG_PSTAMP() do { } while (0)
#define CG_PSTAMP_END() do { } while (0)
#endif

#define CG_PWM_MAXW 4        // weight-gradient register tiles per wave: (stacked rows / 16) * (Cin / 16) <= 32 over 8 waves (128 x 64, 64 x 128, 32 x 112 ...)

template <int NT_, int CT_>
__global__ __launch_bounds__(CG_PWM_THREADS) void cg_pwm_bwd_kernel(CgPwArgs a) {
  const CgPwMaps& t = a.t; const CgPwGeom& g = a.g;
  const CgPwK<NT_, CT_, true> G{g};
  float* sD = reinterpret_cast<float*>(cg_dyn_lds);              // [MM][PS]   dy of every map, stacked
  float* sX = sD + G.MM() * G.PS();                                   // [CinM][PS]
  float* sW = sX + G.CinM() * G.PS();                                 // [MM][WS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, slot = lane >> 4, nw = CG_PWM_THREADS / 64;
  const int lid0 = blockIdx.x * g.per, lid1 = min(g.total, lid0 + g.per);
  if (lid0 >= g.total) return;
  int nst = 0; (void)nst;
  CG_PSTAMP();
  float xbuf[16], dbuf[16];
  // Row pointers of the four staging slots, resolved once: looking the map of a stacked row up in the kernel arguments (three
  // dependent loads indexed by a lane value) per tile kept the prefetch ISSUE at 9 k cycles of a 24 k-cycle tile (tools/stamps_pwm.py)
  const float* xrow[4]; const float* drow[4]; const float* yrow[4];
  long long dstr[4];
  // BatchNorm + PReLU behind the maps (optional, see CgPwMaps.yraw): constants of the four staging rows of this thread
  const bool undo = t.yraw[0] != nullptr;
  float k_mean[4], k_rstd[4], k_scale[4], k_beta[4], k_m1[4], k_m2[4], k_alpha[4];
  float ybuf[16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ((int)threadIdx.x + CG_PWM_THREADS * r) >> G.lgq();
    xrow[r] = row < t.Cin ? t.x + (long long)row * t.P : nullptr;
    drow[r] = nullptr; yrow[r] = nullptr; dstr[r] = 0;
    k_mean[r] = 0.f; k_rstd[r] = 0.f; k_scale[r] = 0.f; k_beta[r] = 0.f; k_m1[r] = 0.f; k_m2[r] = 0.f; k_alpha[r] = 1.f;
    if (row < G.MM()) {
      const int tile = row >> 4, i = g.tile_map[tile], m = g.tile_row0[tile] + (row & 15);
      if (m < t.M[i]) {
        drow[r] = t.dy[i] + (long long)m * t.P; dstr[r] = (long long)t.M[i] * t.P;
        if (undo) {
          yrow[r] = t.yraw[i] + (long long)m * t.P;
          k_mean[r] = t.bn_save[i][m]; k_rstd[r] = t.bn_save[i][t.M[i] + m];
          k_scale[r] = t.bn_gamma[i][m] * k_rstd[r]; k_beta[r] = t.bn_beta[i][m];
          k_alpha[r] = t.prelu[i][0];
          if (t.bn_train) {
            const double cnt = (double)t.B * (double)t.P;
            k_m1[r] = (float)(t.bn_red[i][2 * m] / cnt); k_m2[r] = (float)(t.bn_red[i][2 * m + 1] / cnt);
          }
        }
      }
    }
  }
  auto fetch = [&](int lid) {
    const int b = lid / g.tps, p0 = (lid - b * g.tps) * G.PT(), np = min(G.PT(), t.P - p0);
    long long xo[4], doff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { xo[r] = (long long)b * t.Cin * t.P + p0; doff[r] = (long long)b * dstr[r] + p0; }
    cg_pwm_fetch_rows(g, G, np, xbuf, xrow, xo);
    cg_pwm_fetch_rows(g, G, np, dbuf, drow, doff);
    if (undo) cg_pwm_fetch_rows(g, G, np, ybuf, yrow, doff);
  };
  // gradient in front of the BatchNorm from the one behind the PReLU (cg_norm_act's backward, applied to the staged values; zeros of the
  // padding stay zeros only where dy AND the sums' terms vanish: rows / positions outside the tensor are masked by their null row pointer)
  auto undo_rows = [&](int np) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = (int)threadIdx.x + CG_PWM_THREADS * r, pp = 4 * (e & ((1 << G.lgq()) - 1));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = ybuf[4 * r + j] - k_mean[r];
        const float u = d * k_scale[r] + k_beta[r];
        const float gu = u > 0.f ? dbuf[4 * r + j] : k_alpha[r] * dbuf[4 * r + j];
        const float v = k_scale[r] * (gu - k_m1[r] - d * k_rstd[r] * k_m2[r]);
        dbuf[4 * r + j] = (yrow[r] != nullptr && pp + j < np) ? v : 0.f;
      }
    }
  };
  fetch(lid0);
  cg_pwm_weights(a, G, sW);
  for (int e = tid; e < (G.MM() + G.CinM()) * G.PS(); e += CG_PWM_THREADS) sD[e] = 0.f;
  cg_f32x4 wacc[CG_PWM_MAXW], bacc[CG_PWM_MAXW];      // bacc: row sums of dy (bias gradient) on the tiles of the first channel column
#pragma unroll
  for (int u = 0; u < CG_PWM_MAXW; ++u) { wacc[u] = cg_f32x4{0.f, 0.f, 0.f, 0.f}; bacc[u] = wacc[u]; }
  bool want_db = false;
  for (int i = 0; i < t.n; ++i) want_db = want_db || t.db[i] != nullptr;
  CG_PSTAMP();
  for (int lid = lid0; lid < lid1; ++lid) {
    const int b = lid / g.tps, p0 = (lid - b * g.tps) * G.PT(), np = min(G.PT(), t.P - p0);
    __syncthreads();
    CG_PSTAMP();
    if (undo) undo_rows(np);
    cg_pwm_commit(G, G.MM(), dbuf, sD);                              // rows of the padding and positions beyond the tensor arrive as zeros
    cg_pwm_commit(G, t.Cin, xbuf, sX);
    __syncthreads();
    CG_PSTAMP();
    if (lid + 1 < lid1) fetch(lid + 1);
    CG_PSTAMP();
    // dW[m][c] += sum_p dy[m][p] x[c][p]
    if (NT_) {
      // known geometry: tile id = u * nw + wave = mt * CT + ct with ct = wave % CT (nw % CT == 0): the NU tiles of a wave share their x
      // fragment and advance through the positions together - 1 + NU LDS reads for 4 NU MFMAs in NU independent chains
      constexpr int CTc = CT_ ? CT_ : 1, NU = NT_ * CTc / (CG_PWM_THREADS / 64) ? NT_ * CTc / (CG_PWM_THREADS / 64) : 1;
      static_assert(NT_ == 0 || ((NT_ * CTc) % (CG_PWM_THREADS / 64) == 0 && (CG_PWM_THREADS / 64) % CTc == 0 && NU <= CG_PWM_MAXW), "tile split of the eight waves");
      const int ct = wave % CTc, mtb = wave / CTc;
      const bool dbw = want_db && ct == 0;                 // wave-uniform
      auto product = [&](auto with_db) {
#pragma unroll
        for (int k0 = 0; k0 < G.PT(); k0 += 16) {
          float bv[4];
          cg_tfrag<0>(cg_tfrag_ptr<0>(sX + 16 * ct * G.PS(), G.PS(), l15, slot), G.PS(), k0, bv);
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            float av[4];
            cg_tfrag<0>(cg_tfrag_ptr<0>(sD + 16 * (mtb + u * ((CG_PWM_THREADS / 64) / CTc)) * G.PS(), G.PS(), l15, slot), G.PS(), k0, av);
#pragma unroll
            for (int s = 0; s < 4; ++s) wacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], wacc[u], 0, 0, 0);
            if (decltype(with_db)::value) {                // dy . 1: every column of the tile = the row sums
#pragma unroll
              for (int s = 0; s < 4; ++s) bacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], 1.f, bacc[u], 0, 0, 0);
            }
          }
        }
      };
      if (dbw) product(std::true_type{}); else product(std::false_type{});
    } else {
    // two register tiles at a time (independent MFMA chains)
#pragma unroll
    for (int u = 0; u < CG_PWM_MAXW; u += 2) {      const int id0 = u * nw + wave, id1 = (u + 1) * nw + wave;
      if (id0 < G.NT() * G.CT()) {
        const bool two = id1 < G.NT() * G.CT();
        const int mt0 = id0 / G.CT(), ct0 = id0 - mt0 * G.CT(), mt1 = two ? id1 / G.CT() : mt0, ct1 = two ? id1 - mt1 * G.CT() : ct0;
        const float* ap0 = cg_tfrag_ptr<0>(sD + 16 * mt0 * G.PS(), G.PS(), l15, slot);
        const float* bp0 = cg_tfrag_ptr<0>(sX + 16 * ct0 * G.PS(), G.PS(), l15, slot);
        const float* ap1 = cg_tfrag_ptr<0>(sD + 16 * mt1 * G.PS(), G.PS(), l15, slot);
        const float* bp1 = cg_tfrag_ptr<0>(sX + 16 * ct1 * G.PS(), G.PS(), l15, slot);
        cg_f32x4 w0 = wacc[u], w1 = wacc[u + 1];
        const bool db0 = want_db && ct0 == 0, db1 = want_db && two && ct1 == 0;      // wave-uniform
        cg_f32x4 s0 = bacc[u], s1 = bacc[u + 1];
#pragma unroll 2
        for (int k0 = 0; k0 < G.PT(); k0 += 16) {
          float a0[4], b0[4], a1[4], b1[4];
          cg_tfrag<0>(ap0, G.PS(), k0, a0); cg_tfrag<0>(bp0, G.PS(), k0, b0); cg_tfrag<0>(ap1, G.PS(), k0, a1); cg_tfrag<0>(bp1, G.PS(), k0, b1);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            w0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b0[s], w0, 0, 0, 0);
            w1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b1[s], w1, 0, 0, 0);
          }
          if (db0) {                                     // dy . 1: every column of the tile = the row sums
#pragma unroll
            for (int s = 0; s < 4; ++s) s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], 1.f, s0, 0, 0, 0);
          }
          if (db1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], 1.f, s1, 0, 0, 0);
          }
        }
        wacc[u] = w0; bacc[u] = s0;
        if (two) { wacc[u + 1] = w1; bacc[u + 1] = s1; }
      }
    }
    }
    CG_PSTAMP();
    // dx[p][c] = sum_m dy[m][p] W[m][c]
    for (int w = wave; w < (G.PT() / 32) * G.CT(); w += nw) {
      const int pg = w / G.CT(), ct = w - pg * G.CT(), n0 = 32 * pg, n1 = n0 + 16;
      cg_f32x4 c0 = cg_f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
      const float* wp = cg_tfrag_ptr<1>(sW + 16 * ct, G.WS(), l15, slot);
      const float* dp0 = cg_tfrag_ptr<1>(sD + n0, G.PS(), l15, slot);
      const float* dp1 = cg_tfrag_ptr<1>(sD + n1, G.PS(), l15, slot);
      for (int k0 = 0; k0 < G.MM(); k0 += 16) {
        float wv[4], d0v[4], d1v[4];
        cg_tfrag<1>(wp, G.WS(), k0, wv); cg_tfrag<1>(dp0, G.PS(), k0, d0v); cg_tfrag<1>(dp1, G.PS(), k0, d1v);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d0v[s], wv[s], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d1v[s], wv[s], c1, 0, 0, 0);
        }
      }
      const int c = 16 * ct + l15;
      if (c < t.Cin) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pq = (h ? n1 : n0) + 4 * slot;
          const cg_f32x4 cc = h ? c1 : c0;
          cg_pwm_store_quad(g, t.dx + ((long long)b * t.Cin + c) * t.P + p0, pq, np, cc);
        }
      }
    }
  }
  CG_PSTAMP();
  float* ws = t.dW_ws + (long long)(blockIdx.x % CG_PWM_REPLICAS) * CG_PWM_MAXROWS * t.Cin;
#pragma unroll
  for (int u = 0; u < CG_PWM_MAXW; ++u) {
    const int id = u * nw + wave;
    if (id < G.NT() * G.CT()) {
      const int mt = id / G.CT(), ct = id - mt * G.CT();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * mt + 4 * slot + q, c = 16 * ct + l15;
        if (c < t.Cin) atomicAdd(&ws[r * t.Cin + c], wacc[u][q]);
      }
      if (want_db && ct == 0 && l15 == 0) {
        float* wsb = t.dW_ws + (long long)CG_PWM_REPLICAS * CG_PWM_MAXROWS * t.Cin + (long long)(blockIdx.x % CG_PWM_REPLICAS) * CG_PWM_MAXROWS;
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(&wsb[16 * mt + 4 * slot + q], bacc[u][q]);
      }
    }
  }
  CG_PSTAMP();
  CG_PSTAMP_END();
}

__global__ void cg_pwm_fold_kernel(CgPwArgs a) {
  const CgPwMaps& t = a.t; const CgPwGeom& g = a.g;
  const int i = blockIdx.y;
  if (i >= t.n) return;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < t.M[i] * t.Cin; e += gridDim.x * blockDim.x) {
    const int m = e / t.Cin, c = e - m * t.Cin;
    float s = 0.f;
    for (int r = 0; r < CG_PWM_REPLICAS; ++r) s += t.dW_ws[((long long)r * CG_PWM_MAXROWS + g.row_base[i] + m) * t.Cin + c];
    t.dW[i][e] = s;
  }
  if (t.db[i]) {
    const float* wsb = t.dW_ws + (long long)CG_PWM_REPLICAS * CG_PWM_MAXROWS * t.Cin;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < t.M[i]; m += gridDim.x * blockDim.x) {
      float s = 0.f;
      for (int r = 0; r < CG_PWM_REPLICAS; ++r) s += wsb[(long long)r * CG_PWM_MAXROWS + g.row_base[i] + m];
      t.db[i][m] = s;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int cg_pwm_geometry(const CgPwMaps* t, bool bwd, CgPwGeom* g) {
  if (!t || t->n <= 0 || t->n > CG_PWM_MAXN) return CG_EARG;
  if (t->B <= 0 || t->Cin <= 0 || t->Cin > 128 || t->P <= 0 || (t->P & 1)) return CG_ESHAPE;
  g->vw = (t->P & 3) == 0 ? 4 : 2;
  if (!t->x) return CG_EARG;
  int rows = 0, tile = 0;
  for (int i = 0; i < t->n; ++i) {
    if (t->M[i] <= 0 || t->M[i] > 64) return CG_ESHAPE;
    if (!t->W[i]) return CG_EARG;
    g->row_base[i] = rows;
    for (int r0 = 0; r0 < t->M[i]; r0 += 16) { if (tile >= CG_PWM_MAXROWS / 16) return CG_ESHAPE; g->tile_map[tile] = i; g->tile_row0[tile] = r0; ++tile; }
    rows = 16 * tile;
  }
  for (int k = tile; k < CG_PWM_MAXROWS / 16; ++k) { g->tile_map[k] = 0; g->tile_row0[k] = 1 << 20; }
  for (int i = t->n; i < CG_PWM_MAXN; ++i) g->row_base[i] = 0;
  g->CinM = (t->Cin + 15) & ~15; g->MM = rows; g->WS = g->CinM + 4; g->NT = rows / 16; g->CT = g->CinM / 16;
  if (g->NT * g->CT > CG_PWM_MAXW * (CG_PWM_THREADS / 64)) return CG_ESHAPE;      // weight-gradient register tiles: NT * CT over eight waves
  const int big = bwd ? (g->MM > g->CinM ? g->MM : g->CinM) : g->CinM;
  int pt = 256;                                   // largest power of two with big * pt <= 16 * threads (the staging registers), at most 256:
  while (big * pt > 16 * CG_PWM_THREADS) pt >>= 1; // a 48-row stack (three 10-channel maps, or 40 input channels) gets 128, not 170
  g->PT = pt; g->PS = pt + 4;
  g->lgq = 0;
  while ((4 << g->lgq) < pt) ++g->lgq;
  g->tps = (t->P + pt - 1) / pt;
  g->total = t->B * g->tps;
  const int nwg = g->total < 512 ? g->total : 512;
  g->per = (g->total + nwg - 1) / nwg;
  return CG_OK;
}

// A/B aid (read once, only in a process started with CISTGCN_ABLATION=1): CG_PWM_GENERIC=1 sends every shape to the run-time form
static bool cg_pwm_generic() { static const bool v = getenv("CISTGCN_ABLATION") && getenv("CG_PWM_GENERIC"); return v; }
// the instantiations with compile-time geometry: (stacked 16-row tiles, input 16-row tiles) of the shipped configurations; anything else
// takes the run-time form
#define CG_PWM_DISPATCH(G_, LAUNCH)                                   \
  if (cg_pwm_generic()) LAUNCH(0, 0)                                  \
  else if ((G_).NT == 8 && (G_).CT == 4) LAUNCH(8, 4)        /* four 32-channel towers of a 64-channel block (and 2 x 64 residual maps) */ \
  else if ((G_).NT == 8 && (G_).CT == 1) LAUNCH(8, 1)   /* 128 stacked rows of the 10-channel input block */                         \
  else if ((G_).NT == 4 && (G_).CT == 2) LAUNCH(4, 2)   /* four 16-channel towers of a 32-channel block */                           \
  else if ((G_).NT == 4 && (G_).CT == 4) LAUNCH(4, 4)                                                                                \
  else LAUNCH(0, 0)

extern "C" long long cg_pointwise_maps_ws_floats(int Cin) { return (long long)CG_PWM_REPLICAS * CG_PWM_MAXROWS * (Cin + 1); }      // + the bias-gradient rows

// include/cistgcn_hip.h : cg_pointwise_maps_fwd / cg_pointwise_maps_bwd
extern "C" int cg_pointwise_maps_fwd(const CgPwMaps* t, void* stream_) {
  CgPwArgs a;
  int st = cg_pwm_geometry(t, false, &a.g);
  if (st != CG_OK) return st;
  a.t = *t;
  for (int i = 0; i < t->n; ++i) {
    if (!t->y[i]) return CG_EARG;
    if ((t->stats[i] != nullptr) != (t->stats[0] != nullptr)) return CG_EARG;
  }
  const size_t lds = ((size_t)a.g.CinM * a.g.PS + (size_t)a.g.MM * a.g.WS + 2 + (size_t)5 * a.g.MM) * sizeof(float);
  const int nwg = (a.g.total + a.g.per - 1) / a.g.per;
#define CG_PWM_FWD_LAUNCH(N, C)                                                                                         \
  {                                                                                                                    \
    hipError_t e = cg_lds_limit((const void*)cg_pwm_fwd_kernel<N, C>, lds);                                            \
    if (e != hipSuccess) return (int)e;                                                                                \
    hipLaunchKernelGGL((cg_pwm_fwd_kernel<N, C>), dim3((unsigned)nwg), dim3(CG_PWM_THREADS), lds, (hipStream_t)stream_, a); \
  }
  CG_PWM_DISPATCH(a.g, CG_PWM_FWD_LAUNCH)
#undef CG_PWM_FWD_LAUNCH
  return cg_launch_status();
}

extern "C" int cg_pointwise_maps_bwd(const CgPwMaps* t, void* stream_) {
  CgPwArgs a;
  int st = cg_pwm_geometry(t, true, &a.g);
  if (st != CG_OK) return st;
  a.t = *t;
  if (!t->dx || !t->dW_ws) return CG_EARG;
  for (int i = 0; i < t->n; ++i) {
    if (!t->dy[i] || !t->dW[i]) return CG_EARG;
    if ((t->yraw[i] != nullptr) != (t->yraw[0] != nullptr)) return CG_EARG;
    if (t->yraw[i] && (!t->bn_save[i] || !t->bn_gamma[i] || !t->bn_beta[i] || !t->prelu[i] || (t->bn_train && !t->bn_red[i]))) return CG_EARG;
  }
  const size_t lds = ((size_t)(a.g.MM + a.g.CinM) * a.g.PS + (size_t)a.g.MM * a.g.WS) * sizeof(float);
  const int nwg = (a.g.total + a.g.per - 1) / a.g.per;
  hipStream_t stream = (hipStream_t)stream_;
#define CG_PWM_BWD_LAUNCH(N, C)                                                                                         \
  {                                                                                                                    \
    hipError_t e = cg_lds_limit((const void*)cg_pwm_bwd_kernel<N, C>, lds);                                            \
    if (e != hipSuccess) return (int)e;                                                                                \
    hipLaunchKernelGGL((cg_pwm_bwd_kernel<N, C>), dim3((unsigned)nwg), dim3(CG_PWM_THREADS), lds, stream, a);            \
  }
  CG_PWM_DISPATCH(a.g, CG_PWM_BWD_LAUNCH)
#undef CG_PWM_BWD_LAUNCH
  st = cg_launch_status();
  if (st != CG_OK) return st;
  hipLaunchKernelGGL(cg_pwm_fold_kernel, dim3(8, (unsigned)t->n), dim3(256), 0, stream, a);
  return cg_launch_status();
}
