// Argument block of the time extrapolator's dilated convolutions (fpn_conv.hip); mirrored by include/cistgcn_hip.h and _lib.py.
#pragma once

// n (<= 3) 3x3 convolutions of one input with padding = dilation = dil[i] (1..3): y_i = W_i (*) x + bias_i.
// x (B,C,H,W) with unit stride along W (a permuted view is fine), y_i (B,O,H,W) contiguous; H * W % 4 == 0, H * W <= 256, C <= 64, O <= 32 and the sample's halo image plus one
// weight matrix must fit in LDS (cg_fpn_conv_supported).
struct CgFpnConv {
  int B, C, O, H, W, n;
  int dil[3]; int pad;
  const float* x; long long xs[3];              // element strides of x: batch, channel, row (the last axis is contiguous)
  const float* w[3]; const float* bias[3];      // (O,C,3,3); bias may be null
  float* y[3];
  // backward
  const float* dy[3];
  float* dx;                                    // optional: sum over the convolutions
  float* dw[3]; float* db[3];                   // optional each
  float* ws;                                    // cg_fpn_conv_ws_floats(B, C, O, n) floats of scratch
};
