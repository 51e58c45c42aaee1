// Argument block of the context-head kernels (context_heads.hip); mirrored by include/cistgcn_hip.h and cistgcn_amd/_lib.py.
#pragma once
#include "dstd_tail.h"        // CgTailBN

// ContextLayer heads 1 and 3 (reference CISTGCN.py:408-418 with :465 / :467):
//   head 0: y[b,c] = max_p  PReLU(BN(w0[c] * x[b,p]))   (context_conv1, `.max(-1)[0].max(-1)[0]`, first arg-max)
//   head 1: y[b,c] = mean_p PReLU(BN(w1[c] * x[b,p]))   (context_conv3, `.mean((2, 3))`)
// x (B,P): the (B,1,T_out,3V) view of the cumulated displacement sequence, one input channel; the convolutions are 1 -> C without bias,
// so the (B,C,P) activations are functions of x[b,p] and per-channel constants and are never stored.
#define CG_CTX_MAXC 64
struct CgCtxHeads {
  int B, P, C, train;
  const float* x;
  const double* xstats;         // train: [CG_STAT_REPLICAS][1][2] f64 {sum x, sum x^2} over batch and positions (cg_chan_stats_many)
  const float* w[2]; CgTailBN bn[2]; const float* alpha[2];      // bn.save: [2][C] mean / rstd of the channel (written by the forward)
  float* y[2];                  // (B,C)
  int32_t* arg;                 // (B,C) first arg-max position of head 0
  float* xsave;                 // [2] mean and (biased) variance of x the forward used (train)
  float* tap[2];                // optional (B,C,P): PReLU outputs (diagnostics / branch records)
  // backward
  const float* dy[2];           // (B,C)
  double* red;                  // [CG_STAT_REPLICAS][2][C][3] f64, zero on entry: sum gu, sum gu * x, sum over the negative side of g * u
  float* dx;                    // (B,P)
  float* dw[2]; float* dgamma[2]; float* dbeta[2]; float* dalpha[2];
};
