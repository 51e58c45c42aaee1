// Argument block of the stacked pointwise maps (tower_maps.hip); mirrored by include/cistgcn_hip.h and cistgcn_amd/_lib.py.
#pragma once

#define CG_PWM_MAXN 4        // maps per launch
#define CG_PWM_MAXROWS 128   // sum of the maps' output channels, each rounded up to 16

// n pointwise (1x1) maps of ONE input: y_i[b,o,p] = sum_c W_i[o,c] x[b,c,p], x (B,Cin,P) contiguous, P % 2 == 0, Cin <= 128,
// M_i <= 64, sum of ceil16(M_i) <= 128.  Forward reads x once for all maps; backward reads x and every dy once and produces the
// input gradient of all maps (summed) and every weight gradient.
struct CgPwMaps {
  int B, Cin, P, n;
  const float* x;
  const float* W[CG_PWM_MAXN]; int M[CG_PWM_MAXN];
  float* y[CG_PWM_MAXN];            // (B, M_i, P)
  double* stats[CG_PWM_MAXN];       // optional [CG_STAT_REPLICAS][M_i][2] f64 sums of y_i, zero on entry
  // backward
  const float* dy[CG_PWM_MAXN];
  float* dx;                        // (B, Cin, P): sum over the maps of W_i^T dy_i
  float* dW[CG_PWM_MAXN];           // (M_i, Cin)
  float* dW_ws;                     // cg_pointwise_maps_ws_floats(Cin) zeroed floats (replicated accumulators)
  const float* bias[CG_PWM_MAXN];   // optional (M_i): y_i += bias_i (the residual convolutions, CISTGCN.py:246-254, :357-365)
  float* db[CG_PWM_MAXN];           // backward, optional (M_i): sum over (b, p) of dy_i
  // backward, optional (all maps or none): the maps are followed by BatchNorm2d + PReLU (the Map2Adj towers, CISTGCN.py:138-141) and dy_i is
  // the gradient BEHIND them.  The kernel undoes both while it loads dy_i: it needs the raw map outputs, the BatchNorm's saved mean / rstd
  // ([2][M_i]), gamma, beta, the reduced sums `red` of cg_norm_act_bwd_reduce_many ([M_i][2] f64: sum g, sum g xhat) and the slope.
  const float* yraw[CG_PWM_MAXN];
  const float* bn_save[CG_PWM_MAXN]; const float* bn_gamma[CG_PWM_MAXN]; const float* bn_beta[CG_PWM_MAXN];
  const double* bn_red[CG_PWM_MAXN]; const float* prelu[CG_PWM_MAXN];
  int bn_train, pad;                // 1: batch statistics (the sums enter dy), 0: running statistics
};
