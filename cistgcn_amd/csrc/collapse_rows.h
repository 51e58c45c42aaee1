// Argument block of the frame-collapsing convolution (collapse_rows.hip); mirrored by include/cistgcn_hip.h and cistgcn_amd/_lib.py.
#pragma once
#include "dstd_tail.h"        // CgTailBN

// y[b,o,v] = sum_{c,t} W[o,c,t] x[b,c,t,v]: nn.Conv2d(C, O, (T, 1)) without bias.  x (B,C,T,V) contiguous, W (O, C*T), y (B,O,V);
// V <= 32, O <= 64, C*T % 4 == 0.
struct CgRowsConv {
  int B, C, T, V, O, pad;
  const float* x; const float* W;
  float* y; double* stats;          // stats: optional [CG_STAT_REPLICAS][O][2] f64 sums of y, zero on entry
  // backward
  const float* dy; float* dx; float* dW;
  float* ws;                        // cg_collapse_rows_ws_floats(C, T, O) zeroed floats
  // optional transform of the input on load (in_on != 0): x' = PReLU(BatchNorm2d(x)) with the BatchNorm `in_bn` over the C input channels
  // and the shared slope in_alpha[0] - the first level of a Map2Adj tower (CISTGCN.py:138-141 / :156-158) folded into the load path
  // of its collapsing convolution, so that the activated tensor is never stored.  Forward: in_bn.stats holds the f64 channel sums
  // of x (train; workgroup 0 writes in_bn.save and the running statistics), eval: running statistics.  Backward: in_bn.save;
  // dx is then the gradient with respect to x' (the BatchNorm / PReLU backward belongs to the producer of x: cg_pointwise_maps_bwd
  // with `yraw`).
  int in_on, in_train;
  CgTailBN in_bn;
  const float* in_alpha;
  float* in_tap;                    // forward, optional (with in_on): the activated input x' (B,C,T,V), written as it is formed - for the PReLU branch
                                    // records of the parity tests (production never asks for it: not storing x' is the point)
  double* in_red;                   // backward, optional (with in_on): [2 C + CG_ALPHA_SLOTS] f64, zero on entry: sums of g = dx' PReLU'(u) and g * xhat per
                                    // input channel and the slope-gradient partial sums - what cg_norm_act_bwd_reduce would compute from (dx', x) in a pass of its own
};
