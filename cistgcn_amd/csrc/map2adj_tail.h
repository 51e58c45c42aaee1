// Argument block of the Map2Adj tail kernels (map2adj_tail.hip); mirrored by include/cistgcn_hip.h and cistgcn_amd/_lib.py.
#pragma once
#include "dstd_tail.h"

#define CG_ADJ_MAXW 4        // weight-gradient register tiles per wave: (Kc/16)^2 / 4 waves, Kc <= 64

// One tower pair's tail: rank-1 seed of s (B,V,T) and q (B,T,V) -> expansor (conv Kc x Kc, BatchNorm, Dropout, PReLU, conv Kc x Kc).
// domain 0 (space): slab axis Kc = V, maps J x J = T x T;   domain 1 (time): Kc = T, J = V.
struct CgAdjTail {
  int B, Kc, J, domain, train, pad0;
  const float* s; const float* q;
  const float* W0; CgTailBN bn; const float* alpha; const float* W4;
  float drop_p; unsigned int salt; const unsigned long long* seed;
  float* e; float* adj;                  // (B,Kc,J,J): first conv output (kept for the backward), the adjacency
  float* tap;                            // optional (B,Kc,J,J): output of the PReLU (diagnostics / branch records)
  // backward
  const float* dadj; float* g; double* red;       // g (B,Kc,J,J) scratch; red: cg_map2adj_tail_red_doubles(Kc) f64 words, zero on entry
  float* ds; float* dq; float* part;     // part: cg_map2adj_tail_part_floats(B, Kc, J) scratch
  float* dW0_ws; float* dW4_ws;          // cg_map2adj_tail_ws_floats(Kc) / 2 zeroed floats each
  float* dW0; float* dW4; float* dgamma; float* dbeta; float* dalpha;
};
// launch geometry of one tower (internal): padded slab count, LDS strides, tile width (KcM * PT = 4096), chunking, magic divisors
struct CgAdjGeom { int KcM, WS, JS, Pn, PT, PS, NP, lgq, ntiles, nch; unsigned magicJ, magicKc, magicPad; int tpw; };
struct CgAdjTailPair { int n, nch_max, dbg, pad; CgAdjGeom g[2]; CgAdjTail t[2]; };   // dbg: tuning aid (CG_ADJ_DBG), phases of the backward kernels skipped
