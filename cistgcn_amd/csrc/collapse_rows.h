// Argument block of the frame-collapsing convolution (collapse_rows.hip); mirrored by include/cistgcn_hip.h and cistgcn_amd/_lib.py.
#pragma once

// y[b,o,v] = sum_{c,t} W[o,c,t] x[b,c,t,v]: nn.Conv2d(C, O, (T, 1)) without bias.  x (B,C,T,V) contiguous, W (O, C*T), y (B,O,V);
// V <= 32, O <= 64, C*T % 4 == 0.
struct CgRowsConv {
  int B, C, T, V, O, pad;
  const float* x; const float* W;
  float* y; double* stats;          // stats: optional [CG_STAT_REPLICAS][O][2] f64 sums of y, zero on entry
  // backward
  const float* dy; float* dx; float* dW;
  float* ws;                        // cg_collapse_rows_ws_floats(C, T, O) zeroed floats
};
