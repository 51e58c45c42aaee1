// Argument block of the gate-head kernels (gate_head.hip); mirrored by include/cistgcn_hip.h and cistgcn_amd/_lib.py.
#pragma once
#include "dstd_tail.h"        // CgTailBN

// Tail of the two gate paths of a DSTD_GC block (reference CISTGCN.py:337-352 applied by :378-384), for path p in {s, t}:
//   h2 = PReLU(Dropout(BN2d(z_p)))           z_p (B,C): output of the joint-collapsing convolution conv_p[4], BatchNorm conv_p[5] over the batch
//   u  = cat(h2, stats)                      stats (B,S): the block statistics, S = 2 + 2T
//   y  = Linear(u)                           map_p[0] (C, C+S), no bias
//   h3 = PReLU(Dropout(BN1d(y)))             map_p[1], map_p[3]
//   w  = Linear(h3)                          map_p[4] (C, C): the gate w1 / w2 (B,C)
// Every tensor is (B, <= C+S): one workgroup per path holds the whole batch, the three batch statistics are workgroup reductions.
#define CG_GATE_MAXC 64
#define CG_GATE_MAXS 192
struct CgGatePath {
  const float* z;               // (B,C) contiguous
  const float* stats; long long stats_ld;      // (B,S), row stride in floats
  CgTailBN bn2; const float* alpha2;           // bn.stats unused (the kernel reduces over the batch itself); bn.save [2][C]
  const float* Wl;              // (C, C+S)
  CgTailBN bn3; const float* alpha3;
  const float* W2;              // (C, C)
  unsigned int salt2, salt3;    // dropout site ids of the two Dropout layers
  float* y;                     // (B,C) Linear output in front of bn3 (kept for the backward)
  float* w;                     // (B,C) result
  float* tap2; float* tap3;     // optional (B,C): the two PReLU outputs (diagnostics / branch records)
  // backward
  const float* dw;              // (B,C)
  float* dz;                    // (B,C)
  float* dstats;                // (B,S) contiguous
  float* dWl; float* dW2;
  float* dgamma2; float* dbeta2; float* dalpha2; float* dgamma3; float* dbeta3; float* dalpha3;
  float* scratch;               // backward: cg_gate_head_scratch_floats(B, C, S) floats (the gradients in front of the two BatchNorms)
  double* red;                  // backward: two f64 words, zero on entry (slope sums)
};
struct CgGateHead {
  int B, C, S, train, n, pad;   // n paths (1 or 2)
  float drop_p; int pad2; const unsigned long long* seed;
  CgGatePath p[2];
};
