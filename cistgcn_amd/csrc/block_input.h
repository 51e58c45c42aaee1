// Argument block of the block-input kernels (block_input.hip); mirrored by include/cistgcn_hip.h and cistgcn_amd/_lib.py.
#pragma once
#include "dstd_tail.h"        // CgTailBN

// Head of DSTD_GC.forward (reference CISTGCN.py:375-379): xn = global_norm(x) and _get_stats_(xn) (:360-371) from ONE pass over the block
// input; backward: the gradients of every consumer of xn (up to CG_BIN_MAXG tensors), the gradient of the statistics and the BatchNorm
// backward in two streaming passes instead of four (statistics backward, fan-in sum, BatchNorm reduce, BatchNorm apply).
#define CG_BIN_MAXG 8
struct CgBlockInput {
  int B, C, T, V, train, ng;
  const float* x;               // (B,C,T,V) contiguous block input
  CgTailBN bn;                  // global_norm; bn.stats: [CG_STAT_REPLICAS][C][2] f64 sums of x (train); bn.save [2][C] mean / rstd
  float* xn;                    // (B,C,T,V) normalised input
  float* rm; float* rq;         // (B,C,T) mean / centred sum of squares of every row of V joints of xn (kept for the backward)
  float* out;                   // (B, 2 + 2T) block statistics of xn
  // backward
  const float* g[CG_BIN_MAXG];  // gradients with respect to xn from its consumers, (B,C,T,V) contiguous, ng of them (null entries skipped)
  const float* dout[2];         // gradients of `out` from its (up to two) consumers, (B, 2 + 2T); null entries skipped
  long long dout_ld[2];         // their row strides in floats (column slices of a wider tensor are taken as they are)
  float* pq;                    // (B,C,T,2) scratch: slope / offset of the statistics' gradient per row
  float* gsum;                  // (B,C,T,V) scratch: the summed gradient in front of the BatchNorm (eval mode: may alias dx)
  double* red;                  // [CG_STAT_REPLICAS][C][2] f64, zero on entry
  float* dx; float* dgamma; float* dbeta;
};
