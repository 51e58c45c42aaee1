// Generic strided tensor contraction  Y[g,m,n] (+)= sum_k A[g,m,k] * X[g,k,n] (+ bias[m])
//
// Every linear map of the CIST-GCN path (1x1 / (T,1) / (1,V) / dilated 3x3 convolutions, Linear
// layers, the per-sample adjacency products, rank-1 outer products) and every one of their
// gradients is an instance of this form once each logical index (g, m, n, k) is allowed to be a
// composite of tensor axes.  The host flattens the composites into int32 element-offset tables,
// so the kernel is layout-agnostic: NCTV / NTCV / (N,3,V,T) views, zero-padded halos and dilation
// all reduce to table contents.  64x64 output tile per 256-thread workgroup on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32, exact f32), or 16x64 on the VALU for thin outputs; K staged through LDS in
// steps of 16, optional split-K with fp32 atomics.
//
// This is the scaffold kernel: the hot ST-GCN stage has its own fused kernel (stgcn_domain.hip).
#include "cg_common.h"

#define CG_BK 16      // K step of the 64x64 matrix-core tile
#define CG_BK_THIN 16 // K step of the thin (M <= 16) tile (64 was measured slower: fewer resident workgroups)
#define CG_KT 2048   // k-offset table entries staged in LDS per workgroup (two tables)

// Per K-step the kernel needs one dependent global load per operand element (offset tables are in
// LDS / registers), and the loads of step s+1 are issued before the FMAs of step s so that their
// latency overlaps the compute (register prefetch, single LDS tile).
//
// One launch runs up to CG_MAX_BATCH independent contractions ("horizontal fusion"): the descriptors
// travel by value in the kernel arguments and every workgroup looks up the problem its block id
// falls into.  The model issues the same-depth maps of its parallel branches (gate s/t, the four
// Map2Adj towers, residual maps, and in backward every dA / dX / bias sum of a stage) as one launch.
struct CgContractDesc {
  const float* A; const float* X; float* Y; const float* bias; double* stats; const int32_t* tab;
  int G, M, N, K, splitk, kchunk, a_kfast, x_kfast;
  int accumulate;            // 1: fp32 atomic adds into Y (several problems sum into one zeroed output)
  int x_vec;                 // 1: X is contiguous and 16-byte aligned along n in groups of four -> float4 loads
  int stat_ch;               // number of channels of `stats` (replica stride = 2 * stat_ch doubles)
  int mode;                  // 0: tiled kernel below; 1: streaming kernel (cg_stream_body); 2: K-reduction (cg_kred_body)
  long long block0;          // first block id of this problem inside the launch
  float* ws;                 // mode 2: zeroed scratch of cg_contract_kred_ws_floats(G, M, N) floats
  int chain;                 // mode 1: 1 + index (in the caller's array) of a problem whose product is ADDED to this one's
  int pad2;                  //         output in registers (same G, M, N; it writes nothing itself), 0 = none
};
#define CG_MAX_BATCH 16
struct CgContractBatch { int n; int pad; CgContractDesc d[CG_MAX_BATCH]; };

// fp32-input MFMA (v_mfma_f32_16x16x4_f32): exact f32 fma chain in k order, one A and one B value per lane
// (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]); D: col = lane & 15, row = 4 * (lane >> 4) + reg.
typedef float cg_f32x4 __attribute__((vector_size(16)));
#define CG_LDT 80   // LDS row stride of the 64-wide tiles: 80 mod 32 = 16 keeps the four k-rows of a fragment read on distinct banks

template <int BM, int TM, int BK>
__device__ __forceinline__ void cg_contract_body(const CgContractDesc& d, long long bid, float* As_, float* Xs_,
                                                 int32_t* sKA, int32_t* sKX, double* sStat_) {
  constexpr int BN = 64, TN = 4;
  const float* __restrict__ A = d.A; const float* __restrict__ X = d.X; float* __restrict__ Y = d.Y;
  const float* __restrict__ bias = d.bias; double* __restrict__ stats = d.stats;
  const int G = d.G, M = d.M, N = d.N, K = d.K, splitk = d.splitk, kchunk = d.kchunk;
  const int a_kfast = d.a_kfast, x_kfast = d.x_kfast;
  const bool atomic_out = splitk > 1 || d.accumulate != 0;
  const int32_t* gA = d.tab;
  const int32_t* gX = gA + G;
  const int32_t* gY = gX + G;
  const int32_t* mA = gY + G;
  const int32_t* mY = mA + M;
  const int32_t* mB = mY + M;
  const int32_t* nX = mB + M;
  const int32_t* nY = nX + N;
  const int32_t* kA = nY + N;
  const int32_t* kX = kA + K;
  constexpr int LDA = BM == 64 ? CG_LDT : BM + 1;
  constexpr int LDB = BM == 64 ? CG_LDT : BN + 1;
  float (*As)[LDA] = reinterpret_cast<float (*)[LDA]>(As_);
  float (*Xs)[LDB] = reinterpret_cast<float (*)[LDB]>(Xs_);
  double (*sStat)[2] = reinterpret_cast<double (*)[2]>(sStat_);

  const int tiles_n = (N + BN - 1) / BN;
  const int tiles_m = (M + BM - 1) / BM;
  const int tn = (int)(bid % tiles_n); bid /= tiles_n;
  const int tm = (int)(bid % tiles_m); bid /= tiles_m;
  const int sk = (int)(bid % splitk);  bid /= splitk;
  const int g = (int)bid;

  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = sk * kchunk;
  const int kend = min(K, kbeg + kchunk);
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const bool lds_tab = (kend - kbeg) <= CG_KT;

  if (lds_tab) {
    for (int k = kbeg + tid; k < kend; k += 256) { sKA[k - kbeg] = kA[k]; sKX[k - kbeg] = kX[k]; }
  }
  if (stats != nullptr && tid < 2 * BM) sStat[tid >> 1][tid & 1] = 0.0;

  constexpr int A_PER = (BM * BK) / 256;   // 4 (BM=64) or 1 (BM=16)
  constexpr int X_PER = (BN * BK) / 256;   // 4
  int a_mm[A_PER], a_kk[A_PER], x_nn[X_PER], x_kk[X_PER];
  long long a_off[A_PER], x_off[X_PER];
  const long long baseA = gA[g], baseX = gX[g];
#pragma unroll
  for (int r = 0; r < A_PER; ++r) {
    const int e = tid + 256 * r;
    if (a_kfast) { a_kk[r] = e % BK; a_mm[r] = e / BK; } else { a_mm[r] = e % BM; a_kk[r] = e / BM; }
    const int m = m0 + a_mm[r];
    a_off[r] = m < M ? baseA + mA[m] : -1;
  }
  const bool x_vec = d.x_vec != 0 && !x_kfast;     // element r = 4*q + j of this thread is column 4*(e4 % 16) + j of k-row e4 / 16
#pragma unroll
  for (int r = 0; r < X_PER; ++r) {
    if (x_vec) {
      const int e4 = tid + 256 * (r >> 2);
      x_kk[r] = e4 / (BN / 4); x_nn[r] = (e4 % (BN / 4)) * 4 + (r & 3);
    } else {
      const int e = tid + 256 * r;
      if (x_kfast) { x_kk[r] = e % BK; x_nn[r] = e / BK; } else { x_nn[r] = e % BN; x_kk[r] = e / BN; }
    }
    const int n = n0 + x_nn[r];
    x_off[r] = n < N ? baseX + nX[n] : -1;
  }
  __syncthreads();   // k tables visible

  float ra[A_PER], rx[X_PER];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int r = 0; r < A_PER; ++r) {
      const int k = k0 + a_kk[r];
      float v = 0.f;
      if (a_off[r] >= 0 && k < kend) v = A[a_off[r] + (lds_tab ? sKA[k - kbeg] : kA[k])];
      ra[r] = v;
    }
    if (x_vec) {
#pragma unroll
      for (int q = 0; q < X_PER / 4; ++q) {
        const int k = k0 + x_kk[4 * q];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x_off[4 * q] >= 0 && k < kend) v = *reinterpret_cast<const float4*>(X + x_off[4 * q] + (lds_tab ? sKX[k - kbeg] : kX[k]));
        rx[4 * q] = v.x; rx[4 * q + 1] = v.y; rx[4 * q + 2] = v.z; rx[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < X_PER; ++r) {
        const int k = k0 + x_kk[r];
        float v = 0.f;
        if (x_off[r] >= 0 && k < kend) v = X[x_off[r] + (lds_tab ? sKX[k - kbeg] : kX[k])];
        rx[r] = v;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int r = 0; r < A_PER; ++r) As[a_kk[r]][a_mm[r]] = ra[r];
#pragma unroll
    for (int r = 0; r < X_PER; ++r) Xs[x_kk[r]][x_nn[r]] = rx[r];
  };
  const long long baseY = gY[g];

  if constexpr (BM == 64) {
    // ---- matrix-core path: each wave owns a 32x32 quadrant of the 64x64 tile = 2x2 MFMA tiles of 16x16 ----
    const int lane = tid & 63, wv = tid >> 6;
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    cg_f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      stage();
      __syncthreads();
      if (k0 + BK < kend) fetch(k0 + BK);      // in flight while the MFMAs below run
      const int nks = (min(BK, kend - k0) + 3) / 4;     // short reductions (outer products, K = 3..10) stop early
#pragma unroll 4
      for (int ks = 0; ks < nks; ++ks) {
        const int k = 4 * ks + l4;
        const float a0 = As[k][wm + l15], a1 = As[k][wm + 16 + l15];
        const float b0 = Xs[k][wn + l15], b1 = Xs[k][wn + 16 + l15];
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mt = wm + 16 * i + 4 * l4 + r;          // row inside the tile
        const int m = m0 + mt;
        const bool row_ok = m < M;
        const float bv = (row_ok && bias != nullptr && sk == 0) ? bias[mB[m]] : 0.f;
        const long long rowY = row_ok ? baseY + mY[m] : 0;
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn + 16 * j + l15;
          if (!row_ok || n >= N) continue;
          const float v = acc[i][j][r] + bv;
          if (atomic_out) atomicAdd(&Y[rowY + nY[n]], v);
          else Y[rowY + nY[n]] = v;
          s += (double)v; q += (double)v * (double)v;
        }
        if (stats != nullptr) {      // the 16 lanes of a row reduce in registers, one LDS atomic per row and wave
          s = cg_row16_sum(s); q = cg_row16_sum(q);
          if (l15 == 0 && row_ok) { atomicAdd(&sStat[mt][0], s); atomicAdd(&sStat[mt][1], q); }
        }
      }
    }
  } else {
    // ---- VALU path for thin outputs (M <= 16): one row x four columns per thread ----
    float acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      stage();
      __syncthreads();
      if (k0 + BK < kend) fetch(k0 + BK);      // in flight while the FMAs below run
      const int klim = min(BK, kend - k0);              // short reductions stop early
#pragma unroll 4
      for (int kk = 0; kk < klim; ++kk) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[kk][ty + 16 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Xs[kk][tx + 16 * j];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + ty + 16 * i;
      const bool row_ok = m < M;
      const float bv = (row_ok && bias != nullptr && sk == 0) ? bias[mB[m]] : 0.f;
      const long long rowY = row_ok ? baseY + mY[m] : 0;
      double s = 0.0, q = 0.0;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + tx + 16 * j;
        if (!row_ok || n >= N) continue;
        const float v = acc[i][j] + bv;
        if (atomic_out) atomicAdd(&Y[rowY + nY[n]], v);
        else Y[rowY + nY[n]] = v;
        s += (double)v; q += (double)v * (double)v;
      }
      if (stats != nullptr) {        // tx = 0..15 share the row
        s = cg_row16_sum(s); q = cg_row16_sum(q);
        if (tx == 0 && row_ok) { atomicAdd(&sStat[ty + 16 * i][0], s); atomicAdd(&sStat[ty + 16 * i][1], q); }
      }
    }
  }
  if (stats != nullptr) {           // channel of row m = mB[m] (the bias / statistics index)
    __syncthreads();
    double* rep = stats + (long long)(blockIdx.x % CG_STAT_REPLICAS) * 2 * d.stat_ch;
    if (tid < 2 * BM && m0 + (tid >> 1) < M) atomicAdd(&rep[2 * mB[m0 + (tid >> 1)] + (tid & 1)], sStat[tid >> 1][tid & 1]);
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming variant (mode 1): pointwise maps over a long position axis, Y[m,n] = sum_k A[m,k] X[k,n] with
// K <= 128, N >= 1e5 and X contiguous along n.  The tile kernel above spends its time in dependent round trips
// (tables -> k-step -> barrier -> k-step ... -> stores) with ~8 KB per workgroup in flight.  Here the whole A panel
// sits in LDS once, every wave owns a 64-column strip and feeds the matrix cores straight from global memory:
// lane (l15, l4) loads the float4 X[k = 4 ks + l4][n = strip + 4 l15 .. +3] - a wave instruction reads four full
// 256-byte rows - and component q of that float4 is the B operand of MFMA tile q, whose 16 columns are therefore the
// strided set {strip + 4 j + q}.  All k-steps of a chunk of 32 k are issued before the first MFMA, nothing in the
// K loop waits on a barrier, and the lane ends up with four consecutive columns per output row (float4 stores).
// ---------------------------------------------------------------------------------------------
#define CG_ST_KMAX 128
#define CG_ST_LDA 68                  // A panel row stride (floats): k-rows l4 = 0..3 of a fragment read fall on distinct banks

template <int MI>
__device__ __forceinline__ void cg_stream_body(const CgContractBatch& batch, const int pi, long long bid, float* As,
                                               int32_t* sKX, double* sStat_) {
  const CgContractDesc& d = batch.d[pi];
  float* __restrict__ Y = d.Y;
  const float* __restrict__ bias = d.bias; double* __restrict__ stats = d.stats;
  const int G = d.G, M = d.M, N = d.N;
  const int32_t* gY = d.tab + 2 * G;
  const int32_t* mY = gY + G + M;
  const int32_t* mB = mY + M;
  const int32_t* nY = mB + M + N;
  double (*sStat)[2] = reinterpret_cast<double (*)[2]>(sStat_);
  constexpr int BM = 16 * MI;
  const int tiles_n = (N + 255) / 256, tiles_m = (M + BM - 1) / BM;
  const int tn = (int)(bid % tiles_n); bid /= tiles_n;
  const int tm = (int)(bid % tiles_m); bid /= tiles_m;
  const int g = (int)bid;
  const int m0 = tm * BM;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  if (stats != nullptr && tid < 2 * BM) sStat[tid >> 1][tid & 1] = 0.0;
  const int n = tn * 256 + 64 * wv + 4 * l15;          // first of this lane's four columns (N % 4 == 0)
  const bool col_ok = n < N;

  cg_f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[i][q] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int KU = 8;                                // k-steps (of four k) per chunk
  // the head problem and every problem chained to it (sum of several maps into one output: the input gradient of a
  // tensor that feeds several pointwise maps) run through the same accumulators; only the head writes
  // (the links are walked by INDEX into the kernel-argument struct: a loop-carried pointer into it would force the
  // whole batch into scratch memory)
  for (int li = pi; li >= 0; li = batch.d[li].chain - 1) {
  const float* __restrict__ A = batch.d[li].A; const float* __restrict__ X = batch.d[li].X;
  const int K = batch.d[li].K;
  const int32_t* gA = batch.d[li].tab;
  const int32_t* gX = gA + G;
  const int32_t* mA = gX + 2 * G;
  const int32_t* nX = mA + 3 * M;
  const int32_t* kA = nX + 2 * N;
  const int32_t* kX = kA + K;
  const int Kp = (K + 3) & ~3;
  if (li != pi) __syncthreads();                       // every wave is done with the previous A panel
  // A panel [Kp][BM] and the k offsets of X into LDS
  const long long baseA = gA[g];
  for (int e = tid; e < Kp * BM; e += 256) {
    const int k = e / BM, mm = e - k * BM, m = m0 + mm;
    As[k * CG_ST_LDA + mm] = (k < K && m < M) ? A[baseA + mA[m] + kA[k]] : 0.f;
  }
  for (int k = tid; k < Kp; k += 256) sKX[k] = k < K ? kX[k] : 0;
  const long long xoff = col_ok ? (long long)gX[g] + nX[n] : 0;
  __syncthreads();
  const int nsteps = Kp >> 2;
  for (int s0 = 0; s0 < nsteps; s0 += KU) {            // uniform over the workgroup
    float4 xv[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k = 4 * (s0 + u) + l4;
      xv[u] = (col_ok && s0 + u < nsteps && k < K) ? *reinterpret_cast<const float4*>(X + xoff + sKX[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      if (s0 + u < nsteps) {
        const int k = 4 * (s0 + u) + l4;
        const float x4[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const float a = As[k * CG_ST_LDA + 16 * i + l15];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x4[q], acc[i][q], 0, 0, 0);
        }
      }
    }
  }
  }   // chain

  const long long baseY = gY[g];
  const bool atomic_out = d.accumulate != 0;
  const long long ycol = col_ok ? nY[n] : 0;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mt = 16 * i + 4 * l4 + r, m = m0 + mt;
      const bool ok = m < M && col_ok;
      const float bv = (ok && bias != nullptr) ? bias[mB[m]] : 0.f;
      float4 v = make_float4(acc[i][0][r] + bv, acc[i][1][r] + bv, acc[i][2][r] + bv, acc[i][3][r] + bv);
      if (ok) {
        float* yp = Y + baseY + mY[m] + ycol;
        if (atomic_out) { atomicAdd(yp, v.x); atomicAdd(yp + 1, v.y); atomicAdd(yp + 2, v.z); atomicAdd(yp + 3, v.w); }
        else *reinterpret_cast<float4*>(yp) = v;
      }
      if (stats != nullptr) {
        double s1 = ok ? ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w) : 0.0;
        double s2 = ok ? ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w) : 0.0;
        s1 = cg_row16_sum(s1); s2 = cg_row16_sum(s2);
        if (l15 == 0 && m < M) { atomicAdd(&sStat[mt][0], s1); atomicAdd(&sStat[mt][1], s2); }
      }
    }
  }
  if (stats != nullptr) {
    __syncthreads();
    double* rep = stats + (long long)(blockIdx.x % CG_STAT_REPLICAS) * 2 * d.stat_ch;
    if (tid < 2 * BM && m0 + (tid >> 1) < M) atomicAdd(&rep[2 * mB[m0 + (tid >> 1)] + (tid & 1)], sStat[tid >> 1][tid & 1]);
  }
}

static inline int cg_stream_mi(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : 4); }

__global__ __launch_bounds__(256, 3) void cg_contract_stream_kernel(CgContractBatch batch) {
  __shared__ float As[CG_ST_KMAX * CG_ST_LDA];
  __shared__ int32_t sKX[CG_ST_KMAX];
  __shared__ double sStat[64 * 2];
  long long bid = blockIdx.x;
  int pi = 0;
  for (int i = 0; i < batch.n; ++i)                     // chained members own no blocks (block0 < 0)
    if (batch.d[i].block0 >= 0 && bid >= batch.d[i].block0) pi = i;
  bid -= batch.d[pi].block0;
  const int M = batch.d[pi].M;
  if (M <= 16) cg_stream_body<1>(batch, pi, bid, As, sKX, sStat);
  else if (M <= 32) cg_stream_body<2>(batch, pi, bid, As, sKX, sStat);
  else cg_stream_body<4>(batch, pi, bid, As, sKX, sStat);
}

// ---------------------------------------------------------------------------------------------
// K-reduction variant (mode 2): weight gradients of the pointwise maps, Y[m,n] = sum_k A[m,k] X[n,k] with a
// handful of outputs and K = batch x positions (1e4 .. 1e6).  Both operands are contiguous along k in aligned
// groups of four, so every lane feeds the matrix cores straight from global memory with float4 loads: lane
// (l15, l4) of a wave reads A[row 16i + l15][k0 + 4*l4 .. +3]; MFMA step s of a 16-k group then takes component s,
// i.e. the four k-slots of one v_mfma_f32_16x16x4_f32 are k0 + 4*l4 + s for l4 = 0..3 (A and X agree on the
// assignment, and a sum does not care about the order).  No LDS staging and no barrier inside the K loop; the
// four waves of a workgroup take alternate 16-k groups.  Partial tiles are combined in LDS, added to one of
// CG_KR_REPL zero-initialised replicas in `ws` (fp32 atomics, split / replicas contenders per address) and the
// last workgroup of a tile folds the replicas into Y.  Output tiles are 16x16 or 32x32 (one or four MFMA tiles per
// wave: at 64x64 the f32 matrix pipe, not HBM, would bound the loop).  Runs inside cg_contract_many_kernel, so a
// batch that mixes both kinds of problem stays one launch.
// ---------------------------------------------------------------------------------------------
#define CG_KR_REPL 16
#define CG_KR_QT 1024         // k quads staged per workgroup: kchunk <= 4096

static inline int cg_kred_ti(int M, int N) { return (M > N ? M : N) <= 16 ? 1 : 2; }   // 16x16 or 32x32 output tiles

template <int TI>
__device__ __forceinline__ void cg_kred_body(const CgContractDesc& d, long long bid, float* red, int32_t* sQA, int32_t* sQX,
                                             int* sLast) {
  constexpr int BT = 16 * TI, LDR = BT + 1;
  constexpr int U = 4;                               // 16-k groups in flight per wave (64 consecutive k)
  const float* __restrict__ A = d.A; const float* __restrict__ X = d.X; float* __restrict__ Y = d.Y;
  const int G = d.G, M = d.M, N = d.N, K = d.K, splitk = d.splitk, kchunk = d.kchunk;
  const int32_t* gA = d.tab;
  const int32_t* gX = gA + G;
  const int32_t* gY = gX + G;
  const int32_t* mA = gY + G;
  const int32_t* mY = mA + M;
  const int32_t* mB = mY + M;
  const int32_t* nX = mB + M;
  const int32_t* nY = nX + N;
  const int32_t* kA = nY + N;
  const int32_t* kX = kA + K;
  const int tiles_n = (N + BT - 1) / BT, tiles_m = (M + BT - 1) / BT;
  const int tn = (int)(bid % tiles_n); bid /= tiles_n;
  const int tm = (int)(bid % tiles_m); bid /= tiles_m;
  const int sk = (int)(bid % splitk);  bid /= splitk;
  const int g = (int)bid;
  const int m0 = tm * BT, n0 = tn * BT;
  const int kbeg = sk * kchunk, kend = min(K, kbeg + kchunk);
  const int nq = (kend - kbeg) >> 2;                  // K % 4 == 0 and kchunk % 16 == 0
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;

  for (int q = tid; q < nq; q += 256) { sQA[q] = kA[kbeg + 4 * q]; sQX[q] = kX[kbeg + 4 * q]; }
  for (int e = tid; e < BT * LDR; e += 256) red[e] = 0.f;
  long long aoff[TI], xoff[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = m0 + 16 * i + l15, n = n0 + 16 * i + l15;
    aoff[i] = m < M ? (long long)gA[g] + mA[m] : -1;
    xoff[i] = n < N ? (long long)gX[g] + nX[n] : -1;
  }
  __syncthreads();

  cg_f32x4 acc[TI][TI];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j) acc[i][j] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
  const int ngroups = (nq + 3) >> 2;
  for (int t0 = wv * U; t0 < ngroups; t0 += 4 * U) {  // wave-uniform trip count
    float4 av[U][TI], xv[U][TI];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = 4 * (t0 + u) + l4;
      const bool ok = q < nq;
      const int ka = ok ? sQA[q] : 0, kx = ok ? sQX[q] : 0;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        av[u][i] = (ok && aoff[i] >= 0) ? *reinterpret_cast<const float4*>(A + aoff[i] + ka) : make_float4(0.f, 0.f, 0.f, 0.f);
        xv[u][i] = (ok && xoff[i] >= 0) ? *reinterpret_cast<const float4*>(X + xoff[i] + kx) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const float a4[4] = {av[u][i].x, av[u][i].y, av[u][i].z, av[u][i].w};
#pragma unroll
        for (int j = 0; j < TI; ++j) {
          const float x4[4] = {xv[u][j].x, xv[u][j].y, xv[u][j].z, xv[u][j].w};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s4], x4[s4], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  // four partial tiles -> one in LDS
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[(16 * i + 4 * l4 + r) * LDR + 16 * j + l15], acc[i][j][r]);
  __syncthreads();
  const long long gmn = (long long)G * M * N;
  float* rep = d.ws + (long long)(sk % CG_KR_REPL) * gmn + (long long)g * M * N;
  for (int e = tid; e < BT * BT; e += 256) {
    const int mt = e / BT, nt = e % BT, m = m0 + mt, n = n0 + nt;
    if (m < M && n < N) atomicAdd(&rep[(long long)m * N + n], red[mt * LDR + nt]);
  }
  __threadfence();                                    // this workgroup's adds are visible before it is counted
  __syncthreads();
  unsigned int* cnt = reinterpret_cast<unsigned int*>(d.ws + (long long)CG_KR_REPL * gmn) + ((long long)g * tiles_m + tm) * tiles_n + tn;
  if (tid == 0) *sLast = atomicAdd(cnt, 1u) == (unsigned int)(splitk - 1);
  __syncthreads();
  if (*sLast) {                                       // every split of this tile has been added: fold the replicas
    __threadfence();
    const volatile float* wsv = d.ws + (long long)g * M * N;
    const int nrep = splitk < CG_KR_REPL ? splitk : CG_KR_REPL;
    const long long baseY = gY[g];
    for (int e = tid; e < BT * BT; e += 256) {
      const int mt = e / BT, nt = e % BT, m = m0 + mt, n = n0 + nt;
      if (m >= M || n >= N) continue;
      float sum = d.bias != nullptr ? d.bias[mB[m]] : 0.f;
      for (int r = 0; r < nrep; ++r) sum += wsv[r * gmn + (long long)m * N + n];
      if (d.accumulate) atomicAdd(&Y[baseY + mY[m] + nY[n]], sum);
      else Y[baseY + mY[m] + nY[n]] = sum;
    }
  }
}

__global__ __launch_bounds__(256, 5) void cg_contract_many_kernel(CgContractBatch batch) {
  constexpr int kTile = CG_BK * CG_LDT > CG_BK_THIN * 65 ? CG_BK * CG_LDT : CG_BK_THIN * 65;
  __shared__ float As[kTile];                          // wide tile: BK x 80; thin tile: BK_THIN x 17
  __shared__ float Xs[kTile];                          // wide tile: BK x 80; thin tile: BK_THIN x 65
  __shared__ int32_t sKA[CG_KT];
  __shared__ int32_t sKX[CG_KT];
  __shared__ double sStat[64 * 2];
  long long bid = blockIdx.x;
  int pi = 0;
  for (int i = 1; i < batch.n; ++i)
    if (bid >= batch.d[i].block0) pi = i;
  const CgContractDesc& d = batch.d[pi];
  bid -= d.block0;
  if (d.mode == 2) {                                   // K-reduction problems reuse the tile / table storage
    static_assert(CG_KT >= CG_KR_QT && kTile >= 32 * 33, "LDS reuse");
    if ((d.M > d.N ? d.M : d.N) <= 16) cg_kred_body<1>(d, bid, As, sKA, sKX, reinterpret_cast<int*>(sStat));
    else cg_kred_body<2>(d, bid, As, sKA, sKX, reinterpret_cast<int*>(sStat));
  }
  else if (d.M <= 16) cg_contract_body<16, 1, CG_BK_THIN>(d, bid, As, Xs, sKA, sKX, sStat);
  else cg_contract_body<64, 4, CG_BK>(d, bid, As, Xs, sKA, sKX, sStat);
}

// include/cistgcn_hip.h : floats of zeroed scratch a mode-2 problem needs (replicas + per-tile arrival counters)
extern "C" long long cg_contract_kred_ws_floats(int G, int M, int N) {
  if (G <= 0 || M <= 0 || N <= 0) return 0;
  const int bt = 16 * cg_kred_ti(M, N);
  return (long long)CG_KR_REPL * G * M * N + (long long)G * ((M + bt - 1) / bt) * ((N + bt - 1) / bt);
}

static long long cg_contract_blocks(CgContractDesc& d) {
  if (d.mode == 1) {
    const int bm = 16 * cg_stream_mi(d.M);
    d.kchunk = d.K;
    return (long long)((d.N + 255) / 256) * ((d.M + bm - 1) / bm) * d.G;
  }
  if (d.mode == 2) {
    const int bt = 16 * cg_kred_ti(d.M, d.N);
    const int kchunk = (d.K + d.splitk - 1) / d.splitk;
    d.kchunk = ((kchunk + 15) / 16) * 16;
    return (long long)((d.N + bt - 1) / bt) * ((d.M + bt - 1) / bt) * d.splitk * d.G;
  }
  int kchunk = (d.K + d.splitk - 1) / d.splitk;
  const int bk = d.M <= 16 ? CG_BK_THIN : CG_BK;
  d.kchunk = ((kchunk + bk - 1) / bk) * bk;
  const int BM = d.M <= 16 ? 16 : 64;
  return (long long)((d.N + 63) / 64) * ((d.M + BM - 1) / BM) * d.splitk * d.G;
}

// include/cistgcn_hip.h : cg_contract_many.  Split-K outputs must be zero on entry (the host zeroes the one
// buffer they are carved from).
extern "C" int cg_contract_many(const CgContractDesc* descs, int n, void* stream_) {
  if (!descs || n <= 0 || n > CG_MAX_BATCH) return CG_EARG;
  CgContractBatch batch;                    // tiled + K-reduction problems: one launch
  CgContractBatch strm;                     // streaming problems: their own kernel (register budget), one launch
  batch.n = 0; batch.pad = 0; strm.n = 0; strm.pad = 0;
  long long total = 0, stotal = 0;
  int smap[CG_MAX_BATCH];                   // caller index -> index in `strm`, -1 if not a streaming problem
  for (int i = 0; i < CG_MAX_BATCH; ++i) smap[i] = -1;
  for (int i = 0; i < n; ++i) {
    CgContractDesc d = descs[i];
    if (!d.A || !d.X || !d.Y || !d.tab) return CG_EARG;
    if (d.G <= 0 || d.M <= 0 || d.N <= 0 || d.K <= 0 || d.splitk <= 0) return CG_ESHAPE;
    if (d.stats && (d.splitk > 1 || d.accumulate)) return CG_EARG;
    if (d.x_vec && ((d.N & 3) || ((uintptr_t)d.X & 15))) return CG_EARG;
    if (d.stats && d.stat_ch <= 0) return CG_EARG;
    if (d.mode < 0 || d.mode > 2) return CG_EARG;
    if (d.mode == 2) {
      if (!d.ws || d.stats || (d.K & 3) || (((uintptr_t)d.A | (uintptr_t)d.X) & 15)) return CG_EARG;
      if ((d.K + d.splitk - 1) / d.splitk > 4 * CG_KR_QT - 16) return CG_ESHAPE;
    }
    if (d.mode != 1 && d.chain != 0) return CG_EARG;
    if (d.mode == 1) {
      if (!d.x_vec || d.x_kfast || d.splitk != 1 || ((uintptr_t)d.Y & 15)) return CG_EARG;
      if (d.K > CG_ST_KMAX) return CG_ESHAPE;
      smap[i] = strm.n;
      strm.d[strm.n++] = d;
    } else {
      batch.d[batch.n++] = d;
    }
  }
  // chains: `chain` indexes the caller's array; members (targets of a chain link) own no workgroups
  bool member[CG_MAX_BATCH] = {false};
  for (int i = 0; i < strm.n; ++i) {
    const int c = strm.d[i].chain;
    if (c == 0) continue;
    if (c < 0 || c > n || smap[c - 1] < 0 || smap[c - 1] == i) return CG_EARG;
    const CgContractDesc& t = strm.d[smap[c - 1]];
    if (t.G != strm.d[i].G || t.M != strm.d[i].M || t.N != strm.d[i].N || member[smap[c - 1]]) return CG_ESHAPE;
    member[smap[c - 1]] = true;
    strm.d[i].chain = smap[c - 1] + 1;
  }
  for (int i = 0; i < strm.n; ++i) {
    if (member[i]) { strm.d[i].block0 = -1; if (strm.d[i].stats || strm.d[i].bias) return CG_EARG; continue; }
    strm.d[i].block0 = stotal;
    stotal += cg_contract_blocks(strm.d[i]);
  }
  if (strm.n > 0 && stotal == 0) return CG_EARG;            // a chain without a head
  // K-reduction problems have few, long-running workgroups: give them the lowest block ids so that they start first
  // and the short tile workgroups of the other problems fill in around them (the kernel finds a problem by scanning
  // for the largest block0 <= block id, so block0 must ascend with the position in the batch: reorder the batch).
  CgContractBatch sorted;
  sorted.n = batch.n; sorted.pad = 0;
  int w = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < batch.n; ++i)
      if ((batch.d[i].mode == 2) == (pass == 0)) {
        sorted.d[w] = batch.d[i];
        sorted.d[w].block0 = total;
        total += cg_contract_blocks(sorted.d[w]);
        ++w;
      }
  if (total > 2147483647LL || stotal > 2147483647LL) return CG_ESHAPE;
  if (strm.n) hipLaunchKernelGGL(cg_contract_stream_kernel, dim3((unsigned)stotal), dim3(256), 0, (hipStream_t)stream_, strm);
  if (sorted.n) hipLaunchKernelGGL(cg_contract_many_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream_, sorted);
  return cg_launch_status();
}

// include/cistgcn_hip.h : cg_contract (single problem; zeroes a split-K output itself)
extern "C" int cg_contract(const float* A, const float* X, float* Y, const float* bias, double* stats, const int32_t* tables,
                           int G, int M, int N, int K, int splitk, int a_kfast, int x_kfast,
                           long long y_dense_numel, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !X || !Y || !tables) return CG_EARG;
  if (G <= 0 || M <= 0 || N <= 0 || K <= 0 || splitk <= 0) return CG_ESHAPE;
  if (stats && splitk > 1) return CG_EARG;   // channel sums need final values
  if (splitk > 1) {
    if (y_dense_numel != (long long)G * M * N) return CG_ESHAPE;
    const int zs = cg_zero_fill(Y, y_dense_numel * (long long)sizeof(float), stream);
    if (zs != CG_OK) return zs;
  }
  CgContractDesc d;
  d.A = A; d.X = X; d.Y = Y; d.bias = bias; d.stats = stats; d.tab = tables;
  d.G = G; d.M = M; d.N = N; d.K = K; d.splitk = splitk; d.kchunk = 0; d.a_kfast = a_kfast; d.x_kfast = x_kfast; d.accumulate = 0; d.x_vec = 0; d.stat_ch = M; d.mode = 0; d.block0 = 0; d.ws = nullptr; d.chain = 0; d.pad2 = 0;
  return cg_contract_many(&d, 1, stream_);
}
