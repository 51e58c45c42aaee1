// Generic strided tensor contraction  Y[g,m,n] (+)= sum_k A[g,m,k] * X[g,k,n] (+ bias[m])
//
// Every linear map of the CIST-GCN path (1x1 / (T,1) / (1,V) / dilated 3x3 convolutions, Linear
// layers, the per-sample adjacency products, rank-1 outer products) and every one of their
// gradients is an instance of this form once each logical index (g, m, n, k) is allowed to be a
// composite of tensor axes.  The host flattens the composites into int32 element-offset tables,
// so the kernel is layout-agnostic: NCTV / NTCV / (N,3,V,T) views, zero-padded halos and dilation
// all reduce to table contents.  fp32 FMA on the VALU; 64x64 (or 16x64) output tile per 256-thread
// workgroup, K staged through LDS in steps of 16, optional split-K with fp32 atomics.
//
// This is the scaffold kernel: the hot ST-GCN stage has its own fused kernel (stgcn_domain.hip).
#include "cg_common.h"

#define CG_BK 16
#define CG_KT 2048   // k-offset table entries staged in LDS per workgroup (two tables)

// Per K-step the kernel needs one dependent global load per operand element (offset tables are in
// LDS / registers), and the loads of step s+1 are issued before the FMAs of step s so that their
// latency overlaps the compute (register prefetch, single LDS tile).
template <int BM, int TM>
__global__ __launch_bounds__(256) void cg_contract_kernel(
    const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ Y,
    const float* __restrict__ bias, double* __restrict__ stats, const int32_t* __restrict__ tab,
    int G, int M, int N, int K, int splitk, int kchunk, int a_kfast, int x_kfast) {
  constexpr int BN = 64, TN = 4;
  const int32_t* gA = tab;
  const int32_t* gX = gA + G;
  const int32_t* gY = gX + G;
  const int32_t* mA = gY + G;
  const int32_t* mY = mA + M;
  const int32_t* mB = mY + M;
  const int32_t* nX = mB + M;
  const int32_t* nY = nX + N;
  const int32_t* kA = nY + N;
  const int32_t* kX = kA + K;

  __shared__ float As[CG_BK][BM + 1];
  __shared__ float Xs[CG_BK][BN + 1];
  __shared__ int32_t sKA[CG_KT];
  __shared__ int32_t sKX[CG_KT];
  __shared__ double sStat[BM][2];   // per-row sum / sum of squares of this tile (train-mode BN epilogue)

  const int tiles_n = (N + BN - 1) / BN;
  const int tiles_m = (M + BM - 1) / BM;
  long long bid = blockIdx.x;
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

  constexpr int A_PER = (BM * CG_BK) / 256;   // 4 (BM=64) or 1 (BM=16)
  constexpr int X_PER = (BN * CG_BK) / 256;   // 4
  // fixed (row, k-lane) / (col, k-lane) of the elements this thread stages, and their row/col offsets
  int a_mm[A_PER], a_kk[A_PER], x_nn[X_PER], x_kk[X_PER];
  long long a_off[A_PER], x_off[X_PER];
  const long long baseA = gA[g], baseX = gX[g];
#pragma unroll
  for (int r = 0; r < A_PER; ++r) {
    const int e = tid + 256 * r;
    if (a_kfast) { a_kk[r] = e % CG_BK; a_mm[r] = e / CG_BK; } else { a_mm[r] = e % BM; a_kk[r] = e / BM; }
    const int m = m0 + a_mm[r];
    a_off[r] = m < M ? baseA + mA[m] : -1;
  }
#pragma unroll
  for (int r = 0; r < X_PER; ++r) {
    const int e = tid + 256 * r;
    if (x_kfast) { x_kk[r] = e % CG_BK; x_nn[r] = e / CG_BK; } else { x_nn[r] = e % BN; x_kk[r] = e / BN; }
    const int n = n0 + x_nn[r];
    x_off[r] = n < N ? baseX + nX[n] : -1;
  }
  __syncthreads();   // k tables visible

  float acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;

  float ra[A_PER], rx[X_PER];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int r = 0; r < A_PER; ++r) {
      const int k = k0 + a_kk[r];
      float v = 0.f;
      if (a_off[r] >= 0 && k < kend) v = A[a_off[r] + (lds_tab ? sKA[k - kbeg] : kA[k])];
      ra[r] = v;
    }
#pragma unroll
    for (int r = 0; r < X_PER; ++r) {
      const int k = k0 + x_kk[r];
      float v = 0.f;
      if (x_off[r] >= 0 && k < kend) v = X[x_off[r] + (lds_tab ? sKX[k - kbeg] : kX[k])];
      rx[r] = v;
    }
  };

  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += CG_BK) {
#pragma unroll
    for (int r = 0; r < A_PER; ++r) As[a_kk[r]][a_mm[r]] = ra[r];
#pragma unroll
    for (int r = 0; r < X_PER; ++r) Xs[x_kk[r]][x_nn[r]] = rx[r];
    __syncthreads();
    if (k0 + CG_BK < kend) fetch(k0 + CG_BK);      // in flight while the FMAs below run
#pragma unroll
    for (int kk = 0; kk < CG_BK; ++kk) {
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

  const long long baseY = gY[g];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + ty + 16 * i;
    if (m >= M) continue;
    const float bv = (bias != nullptr && sk == 0) ? bias[mB[m]] : 0.f;
    const long long rowY = baseY + mY[m];
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + tx + 16 * j;
      if (n >= N) continue;
      const float v = acc[i][j] + bv;
      if (splitk > 1) atomicAdd(&Y[rowY + nY[n]], v);
      else Y[rowY + nY[n]] = v;
      s += (double)v; q += (double)v * (double)v;
    }
    if (stats != nullptr) { atomicAdd(&sStat[ty + 16 * i][0], s); atomicAdd(&sStat[ty + 16 * i][1], q); }
  }
  if (stats != nullptr) {           // channel of row m = mB[m] (the bias / statistics index)
    __syncthreads();
    if (tid < 2 * BM && m0 + (tid >> 1) < M) atomicAdd(&stats[2 * mB[m0 + (tid >> 1)] + (tid & 1)], sStat[tid >> 1][tid & 1]);
  }
}

// include/cistgcn_hip.h : cg_contract
extern "C" int cg_contract(const float* A, const float* X, float* Y, const float* bias, double* stats, const int32_t* tables,
                           int G, int M, int N, int K, int splitk, int a_kfast, int x_kfast,
                           long long y_dense_numel, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!A || !X || !Y || !tables) return CG_EARG;
  if (G <= 0 || M <= 0 || N <= 0 || K <= 0 || splitk <= 0) return CG_ESHAPE;
  if (stats && splitk > 1) return CG_EARG;   // channel sums need final values
  if (splitk > 1) {
    // split-K accumulates with atomics into a dense, zero-initialised output
    if (y_dense_numel != (long long)G * M * N) return CG_ESHAPE;
    hipError_t e = hipMemsetAsync(Y, 0, (size_t)y_dense_numel * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
  }
  int kchunk = (K + splitk - 1) / splitk;
  kchunk = ((kchunk + CG_BK - 1) / CG_BK) * CG_BK;
  const bool small_m = M <= 16;
  const int BM = small_m ? 16 : 64;
  const long long tiles = (long long)((N + 63) / 64) * ((M + BM - 1) / BM);
  const long long blocks = tiles * splitk * G;
  if (blocks > 2147483647LL) return CG_ESHAPE;
  dim3 grid((unsigned)blocks), block(256);
  if (small_m)
    hipLaunchKernelGGL((cg_contract_kernel<16, 1>), grid, block, 0, stream, A, X, Y, bias, stats, tables, G, M, N, K, splitk, kchunk, a_kfast, x_kfast);
  else
    hipLaunchKernelGGL((cg_contract_kernel<64, 4>), grid, block, 0, stream, A, X, Y, bias, stats, tables, G, M, N, K, splitk, kchunk, a_kfast, x_kfast);
  return cg_launch_status();
}
