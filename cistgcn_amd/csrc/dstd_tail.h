// Argument block of the DSTD_GC tail kernels (dstd_tail.hip); mirrored by include/cistgcn_hip.h and cistgcn_amd/_lib.py.
#pragma once

#define CG_TAIL_MAXW 8       // dWc register tiles per wave: (C/16) * (2C/16) / 4 waves, C <= 64

// one BatchNorm of the chain.  `stats`: replicated f64 {sum, sum of squares} of ITS INPUT over batch and positions
// ([CG_STAT_REPLICAS][C][2], zero before the producing phase; train mode only).  `save`: [2][C] mean / rstd actually used
// by the forward (written by the first phase that needs them, read by the later phases and the backward).
struct CgTailBN {
  double* stats;
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; long long* num_batches_tracked;
  float momentum, eps;
  float* save;
};

struct CgDstdTail {
  int B, C, T, V, train, pad0;
  const float* y[2];            // tcn outputs of the space / time Domain_GCNN layer (B,C,T,V), pre-BatchNorm
  const float* r[2];            // their residual addends (B,C,T,V): the block input or the residual conv + BN
  const float* w[2];            // gates w1 / w2 (B,C)
  CgTailBN bn_t[2]; const float* alpha_d[2];      // tcn.1 + layer PReLU
  CgTailBN bn_p[2]; const float* alpha_p[2];      // prelu1 / prelu2 (BatchNorm + PReLU)
  const float* Wc; CgTailBN bn_c; const float* alpha_c;   // compressor conv (C, 2C), BN, PReLU
  const float* gate; const float* bres;           // SE gate (B,C), block residual (B,C,T,V)
  float drop_p; unsigned int salt[2]; int pad1; const unsigned long long* seed;
  float* h0; float* pooled; float* out; double* ostats;      // forward results (ostats optional: sums of out)
  float* tap_x[2]; float* tap_a[2]; float* tap_h;            // optional taps (B,C,T,V): outputs of the layer PReLUs, of prelu1/2, of the
                                                             // compressor PReLU - not needed by the computation (diagnostics / branch records)
  // backward
  const float* dout; const float* dpooled; float* dgate;
  double* red_c;                // [2C + CG_ALPHA_SLOTS] f64, zero on entry: channel sums, then the spread partial sums of d alpha
  float* gp[2]; double* red_p[2];   // gradient in front of prelu1/2's BatchNorm (B,C,T,V); [2C + CG_ALPHA_SLOTS] f64 each, zero on entry
  float* dWc_ws; float* dWc;    // cg_dstd_tail_ws_floats(C) zeroed scratch; (C, 2C)
  float* dr[2]; float* dw[2]; double* red_t[2]; float* dy[2];
  float* dgamma_t[2]; float* dbeta_t[2]; float* dalpha_d[2];
  float* dgamma_p[2]; float* dbeta_p[2]; float* dalpha_p[2];
  float* dgamma_c; float* dbeta_c; float* dalpha_c;
};
