/* cistgcn_hip.h — C ABI of libcistgcn_hip.so, the MI355X (gfx950) kernel library behind the
 * CIST-GCN forward/backward hot path.
 *
 * The reference (QualityMinds/cistgcn) is pure Python on stock PyTorch ops: it has no FFI for this
 * path (SURVEY.md §2 "Native / kernel / collective inventory: none").  The boundary a maintainer
 * binds is therefore the set of aten-op groups inside
 *   human_motion_prediction/models/CISTGCN/CISTGCN.py   (CISTGCN.forward, :567-597)
 *   human_motion_prediction/models/layers/SE.py
 *   human_motion_prediction/losses/losses.py:50-61      (mpjpe)
 * Each entry point below names the reference lines it replaces.  INTEGRATION.md shows the ctypes
 * binding that goes into the reference's model file.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 data unless the type says otherwise; the caller
 *    (PyTorch in the shipped host code) owns all memory, nothing is allocated or freed here;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *    with the host, so every call is capturable into a hipGraph;
 *  - return value: 0 = ok, > 0 = hipError_t of the failed runtime call / launch,
 *    -1 = bad argument (null pointer, option without its buffer), -2 = unsupported shape;
 *  - re-entrant: no global state.
 */
#ifndef CISTGCN_HIP_H
#define CISTGCN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* BatchNorm channel sums ("stats" / "ystats" below) are [CG_STAT_REPLICAS][C][2] f64 buffers, zero on entry:
 * every workgroup adds {sum, sum of squares} into replica (block id mod CG_STAT_REPLICAS) so that f64 atomics do
 * not serialise on one address; cg_norm_act_fwd sums the replicas. */
#define CG_STAT_REPLICAS 16
/* The gradient of a shared PReLU slope (one scalar per tensor) is accumulated as CG_ALPHA_SLOTS partial sums behind the channel
 * sums of a `red` buffer: same-address f64 atomics serialise, thousands of workgroups on one word cost more than the kernel. */
#define CG_ALPHA_SLOTS 64

/* 4-D strided view: n[0] batch, n[1] channel, n[2] x n[3] positions; s[] in elements.
 * NCTV, NTCV (CISTGCN.py:582) and (N,3,V,T) (:592) views of one buffer differ only in s[]. */
typedef struct CgView4 {
  long long n[4];
  long long s[4];
} CgView4;

/* ---- generic strided contraction --------------------------------------------------------------
 * Y[g,m,n] = sum_k A[g,m,k] X[g,k,n] (+ bias[m]); g,m,n,k are composite indices given as int32
 * element-offset tables, concatenated in `tables` in this order:
 *   A_g[G] X_g[G] Y_g[G] | A_m[M] Y_m[M] bias_m[M] | X_n[N] Y_n[N] | A_k[K] X_k[K]
 * Replaces: nn.Conv2d 1x1 / (T,1) / (1,V) / dilated 3x3 (CISTGCN.py:54-72,138-153,165-170,229-234,
 * 305,323-340,408-418,454-458,541-545), nn.Linear (:341-352,421-440, SE.py), the adjacency
 * products torch.einsum (:122-124) and torch.matmul / bmm outer products (:187,:471), and the
 * weight / input gradients of all of them.  splitk > 1 accumulates with fp32 atomics into a dense
 * output of y_dense_numel = G*M*N elements which is zeroed here first.  `stats` (optional, f64
 * [channels][2], zero on entry, splitk == 1 only) receives per-channel sum / sum of squares of Y,
 * channel = bias_m[m]: the reduction half of the BatchNorm that follows the convolution. */
int cg_contract(const float* A, const float* X, float* Y, const float* bias, double* stats, const int32_t* tables,
                int G, int M, int N, int K, int splitk, int a_kfast, int x_kfast,
                long long y_dense_numel, void* stream);

/* Horizontal fusion: up to 16 independent contractions in ONE launch (same-depth maps of the parallel branches
 * of a DSTD_GC block - gate s/t, the four Map2Adj towers, residual maps - and, in backward, every dA / dX / bias
 * sum of such a stage).  Split-K and accumulate outputs must be zero on entry (input gradients of maps that
 * share one input are accumulated into ONE buffer instead of being summed by extra kernels). */
typedef struct CgContractDesc {
  const float* A; const float* X; float* Y; const float* bias; double* stats; const int32_t* tab;
  int G, M, N, K, splitk, kchunk /* set by the library */, a_kfast, x_kfast;
  int accumulate;                /* 1: fp32 atomic adds into a zeroed Y shared by several problems */
  int x_vec;                     /* 1: X contiguous + 16-byte aligned along n in groups of four (float4 loads) */
  int stat_ch;                   /* channels of `stats`: it holds CG_STAT_REPLICAS x stat_ch x 2 doubles */
  int mode;                      /* 0: tiled kernel.  1: streaming kernel for pointwise maps over a long position axis:
                                    requires x_vec, K <= 128, splitk 1, and Y contiguous along n in 16-byte aligned groups of
                                    four (the n offsets of Y as those of X).  2: K-reduction kernel for weight gradients (few outputs, K = batch x
                                    positions): requires K % 4 == 0, A and X contiguous along k in 16-byte aligned groups of
                                    four, no `stats`, K / splitk <= 4080, and `ws` */
  long long block0;              /* set by the library */
  float* ws;                     /* mode 2: ZEROED scratch of cg_contract_kred_ws_floats(G, M, N) floats */
  int chain;                     /* mode 1: 1 + index (in `descs`) of another mode-1 problem with the same G, M, N whose product is
                                    added to this one's output in registers (the input gradient of a tensor that feeds several
                                    pointwise maps); the chained problem writes nothing itself (no bias / stats).  0 = none */
  int pad2;
} CgContractDesc;
int cg_contract_many(const CgContractDesc* descs, int n, void* stream);
long long cg_contract_kred_ws_floats(int G, int M, int N);

/* ---- per-channel statistics and the fused BatchNorm / Dropout / PReLU row kernel ---------------
 * stats[c] = { sum, sum of squares } over batch and positions of x*pre (f64, must be zero on entry).
 * Replaces the reduction half of nn.BatchNorm{1,2}d in train mode. */
int cg_chan_stats(const float* x, const CgView4* xv, const float* pre, double* stats, void* stream);
typedef struct CgStatsArgs { const float* x; CgView4 xv; const float* pre; double* stats; } CgStatsArgs;
int cg_chan_stats_many(const CgStatsArgs* items, int n, void* stream);   /* up to 6 tensors per launch */
/* out[c] += sum over batch and positions (bias gradients of the convolutions); `out` (C floats) zero on entry. */
int cg_chan_sum(const float* x, const CgView4* xv, float* out, void* stream);

/* y = PReLU( Dropout( (x*pre)*scale + shift ) [+ add] ) [+ add if add_post]
 * with (scale, shift) from batch statistics (bn_mode 1, also updates running stats exactly like
 * nn.BatchNorm: momentum, unbiased running_var), running statistics (bn_mode 2) or identity (0).
 * Replaces every Conv->BN->[Dropout]->[PReLU] tail, `tcn + residual -> PReLU` (CISTGCN.py:266-269),
 * `PReLU(BN(w*x))` (:388), the SE channel scale (SE.py:20,41) and the residual sums (:390,:586). */
typedef struct CgNormAct {
  const float* x;  CgView4 xv;
  float* y;        CgView4 yv;
  const float* pre;              /* (B,C) per-sample channel gate or NULL */
  const float* add; CgView4 av;  /* addend or NULL */
  int add_post;                  /* 0: before the PReLU, 1: after it */
  int bn_mode;                   /* 0 none, 1 batch statistics, 2 running statistics */
  const double* stats;           /* [C][2], bn_mode 1 */
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; long long* num_batches_tracked;
  float momentum, eps;
  float* save_mean; float* save_rstd;   /* [C] written by fwd, read by bwd */
  float drop_p; const unsigned long long* seed; unsigned int salt;
  const float* alpha; int alpha_n;      /* PReLU slope(s) (1 or C) or NULL */
  /* backward only */
  const float* dy; CgView4 dyv;
  float* dx; CgView4 dxv;
  float* dadd; CgView4 dav;
  float* dpre;
  double* red;                   /* [2C + (alpha_n == 1 ? CG_ALPHA_SLOTS : alpha_n)] f64 scratch, zero on entry */
  float* dgamma; float* dbeta; float* dalpha;
  double* ystats;                /* forward, optional: [C][2] f64 sums of y (zero on entry) for the BatchNorm consuming y */
} CgNormAct;
int cg_norm_act_fwd(const CgNormAct* a, void* stream);
int cg_norm_act_bwd(const CgNormAct* a, int need_reduce, void* stream);
/* the same for up to 6 independent row problems per launch (branches of one block at the same depth) */
int cg_norm_act_fwd_many(const CgNormAct* arr, int n, void* stream);
int cg_norm_act_bwd_many(const CgNormAct* arr, const int* need_reduce, int n, void* stream);
/* pass 1 of the backward alone (channel sums, slope gradient) + dgamma / dbeta / dalpha: for a consumer that applies the BatchNorm / PReLU
 * backward itself while loading the gradient (cg_pointwise_maps_bwd with yraw) */
int cg_norm_act_bwd_reduce_many(const CgNormAct* arr, int n, void* stream);
/* dgamma / dbeta / dalpha from `red` alone: the reduction was done by the kernel that produced the gradient (cg_collapse_rows_bwd /
 * cg_collapse_cols_bwd with in_red); reads xv.n[1] (channels), bn_mode, red, alpha / alpha_n, dgamma / dbeta / dalpha of each problem */
int cg_norm_act_params_many(const CgNormAct* arr, int n, void* stream);

/* per-(b,c) mean (kind 0), max with first arg-max (kind 1) or sum (kind 2) over the positions; adjoints of 0/1.
 * Replaces AdaptiveAvgPool (SE.py:8,27; CISTGCN.py:69,76), .max(-1)[0] chains and .mean((2,3))
 * (CISTGCN.py:465-467). */
int cg_reduce_bc(const float* x, const CgView4* xv, int kind, float* out, int32_t* arg, void* stream);
int cg_reduce_bc_bwd(const float* dout, const int32_t* arg, int kind, float* dx, const CgView4* dxv, void* stream);

/* y = a (+ b) (+ c) on strided views: torch.cat slices (CISTGCN.py:77,378-380,388,468), halo
 * padding for the dilated FPN convolutions (:54-68), F.interpolate broadcast (:76), the tail
 * x[:, -1:] + x8^T + act (:595-597). */
/* y = a[0] + ... + a[n-1] (n <= 8), same logical shape, any strides: sums the gradients of a multi-consumer tensor in one
 * launch (replaces autograd's one accumulation kernel per extra consumer) */
typedef struct CgSumItem { const float* a; CgView4 av; } CgSumItem;
int cg_sum_many(float* y, const CgView4* yv, const CgSumItem* items, int n, void* stream);
int cg_add3(float* y, const CgView4* yv, const float* a, const CgView4* av, const float* b, const CgView4* bv,
            const float* c, const CgView4* cv, void* stream);
typedef struct CgCopyItem { float* y; CgView4 yv; const float* a; CgView4 av; } CgCopyItem;
int cg_copy_many(const CgCopyItem* items, int n, void* stream);   /* up to 4 strided copies (cat slices) per launch */
int cg_zero(void* p, long long bytes, void* stream);

/* ---- stage kernels --------------------------------------------------------------------------------
 * feature lift, CISTGCN.py:568-577: x (B,T,V,3) -> (B,10,T,V) = [x, acc, vel, |vel|] and its adjoint */
int cg_feature_lift_fwd(const float* x, float* f, long long B, long long T, long long V, void* stream);
int cg_feature_lift_bwd(const float* x, const float* df, float* dx, long long B, long long T, long long V, void* stream);
/* DSTD_GC._get_stats_, CISTGCN.py:360-371: contiguous (B,C,T,V) -> (B, 2+2T) */
int cg_dstd_stats_fwd(const float* x, float* out, int B, int C, int T, int V, void* stream);
int cg_dstd_stats_bwd(const float* x, const float* dout, float* dx, int B, int C, int T, int V, void* stream);
/* SELayer excitation, SE.py:9-14,30-35: gate = sigmoid(W2 relu(W1 pooled)); W1 (H,C), W2 (C,H) */
int cg_se_gate_fwd(const float* pooled, const float* W1, const float* W2, float* gate, int B, int C, int H, void* stream);
int cg_se_gate_bwd(const float* pooled, const float* W1, const float* W2, const float* gate, const float* dgate,
                   float* dpooled, float* dW1, float* dW2, int B, int C, int H, int prezeroed, void* stream);
/* prezeroed != 0: dW1/dW2 are already zero (slices of the per-step zero pool), no memset is issued */
/* rank-1 adjacency seed of Map2Adj, CISTGCN.py:183-189 (`torch.matmul` of the joint and time summaries): s (B,V,T) and
 * q (B,T,V) contiguous; domain 0: o[b,v,t,u] = s[b,v,t]*q[b,u,v] (B,V,T,T); domain 1: o[b,t,v,w] = s[b,v,t]*q[b,t,w]
 * (B,T,V,V).  Up to two problems (the space and time tower of one block) per launch; backward reads d o once and writes
 * both d s and d q.  T, V <= 64; o / dout 16-byte aligned. */
typedef struct CgRank1 { const float* s; const float* q; float* o; const float* dout; float* ds; float* dq; int domain; int pad; } CgRank1;
int cg_rank1_adj_fwd(const CgRank1* items, int n, int B, int T, int V, void* stream);
int cg_rank1_adj_bwd(const CgRank1* items, int n, int B, int T, int V, void* stream);
/* cumsum over axis 1 of a strided 4-D view (B,L,R1,R2), CISTGCN.py:589 (reverse = adjoint) */
int cg_cumsum(const float* x, const CgView4* xv, float* y, const CgView4* yv, int reverse, void* stream);
/* MPJPE, losses/losses.py:50-61 (reduce_axis=[]): pred/target contiguous (N,3); loss is one float */
int cg_mpjpe_fwd(const float* pred, const float* tgt, float* loss, long long N, void* stream);
int cg_mpjpe_bwd(const float* pred, const float* tgt, const float* gloss, float* dpred, long long N, void* stream);
/* advances the device-resident dropout seed (one word) once per training step */
int cg_seed_bump(unsigned long long* seed, void* stream);

/* ---- fused ST-GCN stage (the dominant kernel) ---------------------------------------------------
 * Domain_GCNN_layer core, CISTGCN.py:265-266 with :122-124:
 *   y[b,co,.,.] = bias[co] + sum_ci W[co,ci] * G[b,ci,.,.],
 *   G = einsum('nctv,nvtq->ncqv', x, Adj)  (domain 0, Adj (B,V,T,T))   or
 *       einsum('nctv,ntvw->nctw', x, Adj)  (domain 1, Adj (B,T,V,V)).
 * x, y contiguous (B,C,T,V).  Adjacency is staged in LDS, the graph product and the channel mix
 * run back to back without the intermediate G touching HBM.  Optional per-channel f64 sums of y
 * (ystats, [Cout][2], zero on entry) feed the train-mode BatchNorm that follows (:235).
 * Backward: dx, dAdj, dW, db from dy.  dW/db partial sums go through `ws`, a caller-owned scratch of
 * cg_stgcn_domain_bwd_ws_floats(Cin, Cout) floats (zeroed here unless ws_prezeroed != 0; replicated accumulators
 * keep the fp32 atomics off a single address), and are folded into dW/db by a second tiny kernel.
 * Wide layers (16+ channels) at batch sizes that fill the chip run as plane kernels (csrc/stgcn_domain_planes.hip: the channel
 * mix first, on whole plane rows; LDS transposes to the group-major order of the graph product); smaller launches use the tile
 * kernels.  cg_stgcn_domain_planes_min_workgroups(n) moves that switch (n workgroups; returns the previous value, n < 0 only
 * reads it): both generations compute the same function, tests pin either one. */
long long cg_stgcn_domain_planes_min_workgroups(long long n);
int cg_stgcn_domain_fwd(const float* x, const float* adj, const float* W, const float* bias, float* y, double* ystats,
                        int B, int Cin, int Cout, int T, int V, int domain, void* stream);
long long cg_stgcn_domain_bwd_ws_floats(int Cin, int Cout);
int cg_stgcn_domain_bwd(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj,
                        float* dW, float* dbias, float* ws, int B, int Cin, int Cout, int T, int V, int domain,
                        int ws_prezeroed, void* stream);

/* ---- tail of a DSTD_GC block as phase kernels (SURVEY 8b `dstd_combine`) ------------------------------------------------
 * CISTGCN.py:266-269 (`tcn` BatchNorm + Dropout + residual + PReLU of both Domain_GCNN layers), :388 (PReLU(BN(w * x)) of both
 * branches + cat), :305-309 (compressor 1x1 conv + BN + PReLU + SELayer2d), :390 (block residual).  In train mode every
 * BatchNorm needs the batch statistics of its input first, so the chain is cut at those barriers and nowhere else:
 *   forward  phase 1: sums of z_i = w_i * PReLU(Dropout(BN(y_i)) + r_i) | 2: h0 = Wc [a_1; a_2] on the matrix cores, a_i =
 *            PReLU(BN(z_i)) rebuilt per tile, + sums of h0 | 3: pooled = mean PReLU(BN(h0)) | (cg_se_gate_fwd) | 4: out = h * gate
 *            + block residual (+ sums of out for the next block's BatchNorm);
 *   backward phase 1: dgate | (cg_se_gate_bwd) | 2: sums at the compressor BatchNorm | 3: dh0, d a = Wc^T dh0, dWc on the matrix
 *            cores, gradient in front of prelu1/2 + its sums | 4: gate gradients dw, gradient of the residual addends, sums of
 *            the tcn BatchNorm | 5: dy_i and every per-channel parameter gradient.
 * All tensors contiguous (B,C,T,V) / (B,C); C <= 64.  `stats` / `red_*` / `dWc_ws` are zeroed by the caller (slices of the
 * per-step arenas); BatchNorm bookkeeping (save mean / rstd, running statistics) as cg_norm_act. */
typedef struct CgTailBN {
  double* stats;                 /* [CG_STAT_REPLICAS][C][2] sums of this BatchNorm's input (train mode) */
  const float* gamma; const float* beta;
  float* running_mean; float* running_var; long long* num_batches_tracked;
  float momentum, eps;
  float* save;                   /* [2][C] mean, rstd used by the forward */
} CgTailBN;
typedef struct CgDstdTail {
  int B, C, T, V, train, pad0;
  const float* y[2]; const float* r[2]; const float* w[2];
  CgTailBN bn_t[2]; const float* alpha_d[2];
  CgTailBN bn_p[2]; const float* alpha_p[2];
  const float* Wc; CgTailBN bn_c; const float* alpha_c;
  const float* gate; const float* bres;
  float drop_p; unsigned int salt[2]; int pad1; const unsigned long long* seed;
  float* h0; float* pooled; float* out; double* ostats;
  float* tap_x[2]; float* tap_a[2]; float* tap_h;      /* optional: outputs of the five PReLUs (diagnostics) */
  const float* dout; const float* dpooled; float* dgate;
  double* red_c;
  float* gp[2]; double* red_p[2];
  float* dWc_ws; float* dWc;
  float* dr[2]; float* dw[2]; double* red_t[2]; float* dy[2];
  float* dgamma_t[2]; float* dbeta_t[2]; float* dalpha_d[2];
  float* dgamma_p[2]; float* dbeta_p[2]; float* dalpha_p[2];
  float* dgamma_c; float* dbeta_c; float* dalpha_c;
} CgDstdTail;
int cg_dstd_tail_fwd(const CgDstdTail* t, int phase, void* stream);
int cg_dstd_tail_bwd(const CgDstdTail* t, int phase, void* stream);
long long cg_dstd_tail_ws_floats(int C);

/* ---- frame-collapsing convolution nn.Conv2d(C, O, (T, 1)) (no bias): first convolution of the gate paths CISTGCN.py:331-336 and
 * second convolution of Map2Adj.time_compress :138-150.  y[b,o,v] = sum_{c,t} W[o,c,t] x[b,c,t,v]; x (B,C,T,V) contiguous, W (O, C*T),
 * y (B,O,V); V <= 32, O <= 64, C*T % 4 == 0 (else CG_ESHAPE: cg_contract_many).  Forward: one workgroup per sample; backward:
 * dx and dW from one pass (workgroup = 64 rows of W x a slice of the samples). */
typedef struct CgRowsConv {
  int B, C, T, V, O, pad;
  const float* x; const float* W;
  float* y; double* stats;          /* stats: optional [CG_STAT_REPLICAS][O][2] f64 sums of y, zero on entry */
  const float* dy; float* dx; float* dW;
  float* ws;                        /* cg_collapse_rows_ws_floats(C, T, O) zeroed floats */
  /* optional transform of the input on load (in_on != 0): x' = PReLU(BatchNorm2d(x)) over the C input channels with the shared slope
   * in_alpha[0]: the first level of a Map2Adj tower (CISTGCN.py:138-141 / :156-158) folded into the load path of its collapsing
   * convolution - the activated tensor is never stored.  Forward: in_bn.stats = f64 channel sums of x (train; workgroup 0 writes
   * in_bn.save and the running statistics), eval: running statistics.  Backward: in_bn.save; dx is then the gradient with respect to
   * x' (the BatchNorm / PReLU backward is applied by the producer of x: cg_pointwise_maps_bwd with `yraw`). */
  int in_on, in_train;
  CgTailBN in_bn;
  const float* in_alpha;
  float* in_tap;                    /* forward, optional (with in_on): the activated input x' (B,C,T,V), written as it is formed (branch records of the parity tests) */
  double* in_red;                   /* backward, optional (with in_on): [2 C + CG_ALPHA_SLOTS] f64, zero on entry: sums of g = dx' PReLU'(u) and g * xhat per input
                                     * channel + the slope-gradient partial sums (the reduction pass of the BatchNorm / PReLU backward, done where dx' is produced) */
} CgRowsConv;
int cg_collapse_rows_fwd(const CgRowsConv* t, void* stream);
int cg_collapse_rows_bwd(const CgRowsConv* t, void* stream);
long long cg_collapse_rows_ws_floats(int C, int T, int O);
/* The same for the joint axis: nn.Conv2d(C, O, (1, V)) (no bias), second convolution of Map2Adj.joint_compress, CISTGCN.py:152-163.
 * y[b,o,t] = sum_{c,v} W[o,c,v] x[b,c,t,v]; same argument block with W (O, C*V), y (B,O,T), dy (B,O,T), ws of
 * cg_collapse_cols_ws_floats(C, V, O) zeroed floats; T <= 64, O <= 64, C*V % 4 == 0 (cg_collapse_cols_supported; else cg_contract_many). */
int cg_collapse_cols_fwd(const CgRowsConv* t, void* stream);
int cg_collapse_cols_bwd(const CgRowsConv* t, void* stream);
int cg_collapse_cols_supported(int C, int T, int V, int O);
long long cg_collapse_cols_ws_floats(int C, int V, int O);

/* ---- dilated 3x3 convolutions of the time extrapolator, FPN CISTGCN.py:54-79: n <= 3 convolutions with padding = dilation =
 * dil[i] of ONE input (B,C,H,W) = (batch, frames, channels, joints).  A sample fits in LDS with its halo: forward, input gradient
 * (summed over the convolutions) and weight / bias gradients each walk whole samples.  Shapes outside cg_fpn_conv_supported:
 * CG_ESHAPE (the caller uses cg_contract_many). */
typedef struct CgFpnConv {
  int B, C, O, H, W, n;
  int dil[3]; int pad;
  const float* x; long long xs[3];     /* element strides of x: batch, channel, row; unit stride along W */
  const float* w[3]; const float* bias[3];
  float* y[3];
  const float* dy[3];
  float* dx;
  float* dw[3]; float* db[3];
  float* ws;                     /* cg_fpn_conv_ws_floats(B, C, O, n) floats of scratch */
} CgFpnConv;
int cg_fpn_conv_fwd(const CgFpnConv* t, void* stream);
int cg_fpn_conv_bwd(const CgFpnConv* t, void* stream);
int cg_fpn_conv_supported(int B, int C, int O, int H, int W);
long long cg_fpn_conv_ws_floats(int B, int C, int O, int n);

/* ---- stacked pointwise maps of one input: the first convolutions of the Map2Adj towers of a block, CISTGCN.py:138-163 applied
 * to the normalised block input by :183-186 (up to four 1x1 convolutions of the same (B,C,T,V) tensor).  Forward: every y_i =
 * W_i x from one read of x, with the f64 channel sums of y_i (train-mode BatchNorm behind it).  Backward: dx = sum_i W_i^T dy_i and
 * every dW_i = dy_i x^T from one read of x and of each dy_i.  x (B,Cin,P) contiguous, P = T*V with P % 2 == 0 (rows of 16- or 8-byte alignment), Cin <= 128,
 * M_i <= 64, sum of ceil16(M_i) <= 128, (sum of ceil16(M_i) / 16) * ceil(Cin / 16) <= 32; other shapes: CG_ESHAPE (the caller uses cg_contract_many). */
#define CG_PWM_MAXN 4
typedef struct CgPwMaps {
  int B, Cin, P, n;
  const float* x;
  const float* W[CG_PWM_MAXN]; int M[CG_PWM_MAXN];       /* (M_i, Cin) */
  float* y[CG_PWM_MAXN];                                 /* (B, M_i, P) */
  double* stats[CG_PWM_MAXN];                            /* all null, or [CG_STAT_REPLICAS][M_i][2] each, zero on entry */
  const float* dy[CG_PWM_MAXN];
  float* dx;                                             /* (B, Cin, P) */
  float* dW[CG_PWM_MAXN];
  float* dW_ws;                                          /* cg_pointwise_maps_ws_floats(Cin) zeroed floats */
  const float* bias[CG_PWM_MAXN];                        /* optional (M_i): y_i = W_i x + bias_i (nn.Conv2d(..., 1) with bias: the residual maps) */
  float* db[CG_PWM_MAXN];                                /* backward, optional (M_i): bias gradient, sum of dy_i over batch and positions */
  /* backward, optional (all maps or none): BatchNorm2d + PReLU follow the maps (the Map2Adj towers, CISTGCN.py:138-141) and dy_i is the
   * gradient BEHIND them; the kernel undoes both while it loads dy_i (raw outputs yraw, saved mean / rstd [2][M_i], gamma, beta, the sums
   * `bn_red` [M_i][2] f64 of cg_norm_act_bwd_reduce_many, the slope) */
  const float* yraw[CG_PWM_MAXN];
  const float* bn_save[CG_PWM_MAXN]; const float* bn_gamma[CG_PWM_MAXN]; const float* bn_beta[CG_PWM_MAXN];
  const double* bn_red[CG_PWM_MAXN]; const float* prelu[CG_PWM_MAXN];
  int bn_train, pad;
} CgPwMaps;
int cg_pointwise_maps_fwd(const CgPwMaps* t, void* stream);
int cg_pointwise_maps_bwd(const CgPwMaps* t, void* stream);
long long cg_pointwise_maps_ws_floats(int Cin);

/* ---- tail of the interpretability map, Map2Adj.forward CISTGCN.py:183-189 (expansor :165-170) ------------------------
 * s (B,V,T), q (B,T,V): outputs of the joint / time towers.  Seed o = s (x) q (space: o[b,v,t,u] = s[b,v,t] q[b,u,v];
 * time: o[b,t,v,w] = s[b,v,t] q[b,t,w]) -> conv W0 over the slab axis -> BatchNorm -> Dropout -> PReLU -> conv W4 = Adj.
 * The seed is generated inside the kernels, never stored.  `items`: the towers of a block sharing a launch (n <= 2, same B).
 *   forward phase 1: e = W0 o + f64 channel sums | 2: Adj = W4 PReLU(Dropout(BN(e)))
 *   backward phase 1: g = gradient in front of the BatchNorm, its sums, d alpha, dW4 | 2: d e, ds, dq, dW0, d gamma / d beta
 * domain 0 (space): Kc = V, J = T; domain 1 (time): Kc = T, J = V; Kc, J <= 64.  `bn.stats`, `red`, `dW*_ws` zeroed by the caller. */
typedef struct CgAdjTail {
  int B, Kc, J, domain, train, pad0;
  const float* s; const float* q;
  const float* W0; CgTailBN bn; const float* alpha; const float* W4;
  float drop_p; unsigned int salt; const unsigned long long* seed;
  float* e; float* adj;          /* (B,Kc,J,J) */
  float* tap;                    /* optional (B,Kc,J,J): PReLU output (diagnostics) */
  const float* dadj; float* g; double* red;       /* g (B,Kc,J,J) scratch; red: cg_map2adj_tail_red_doubles(Kc) f64 words */
  float* ds; float* dq; float* part;   /* part: cg_map2adj_tail_part_floats(B, Kc, J) floats of scratch */
  float* dW0_ws; float* dW4_ws;  /* cg_map2adj_tail_ws_floats(Kc) / 2 floats each */
  float* dW0; float* dW4; float* dgamma; float* dbeta; float* dalpha;
} CgAdjTail;
int cg_map2adj_tail_fwd(const CgAdjTail* items, int n, int phase, void* stream);
int cg_map2adj_tail_bwd(const CgAdjTail* items, int n, int phase, void* stream);
long long cg_map2adj_tail_ws_floats(int Kc);
long long cg_map2adj_tail_part_floats(int B, int Kc, int J);
long long cg_map2adj_tail_red_doubles(int Kc);

/* ---- head of a DSTD_GC block, DSTD_GC.forward CISTGCN.py:375-379 with _get_stats_ :360-371 ---------------
 * xn = global_norm(x) (BatchNorm2d) and the block statistics of xn from one pass over x; backward: the gradients of all consumers of
 * xn (g[0..ng-1], null entries skipped), the gradient of the statistics (dout[0] + dout[1]) and the BatchNorm backward in two
 * streaming passes.  x (B,C,T,V) contiguous with T*V <= 4608 (cg_block_input_supported; else CG_ESHAPE: cg_norm_act + cg_dstd_stats).
 * Forward (two launches) writes xn, rm / rq ((B,C,T) row means and centred sums of squares, read again by the backward), out
 * (B, 2+2T) and bn.save; train mode needs bn.stats = f64 sums of x.  Backward (three launches) needs pq (B,C,T,2) and gsum (B,C,T,V)
 * scratch and red = [CG_STAT_REPLICAS][C][2] zeroed f64 words; writes dx, dgamma, dbeta. */
#define CG_BIN_MAXG 8
typedef struct CgBlockInput {
  int B, C, T, V, train, ng;
  const float* x;
  CgTailBN bn;
  float* xn;
  float* rm; float* rq;
  float* out;
  const float* g[CG_BIN_MAXG];
  const float* dout[2];
  long long dout_ld[2];         /* row strides of dout[i] in floats */
  float* pq;
  float* gsum;
  double* red;
  float* dx; float* dgamma; float* dbeta;
} CgBlockInput;
int cg_block_input_fwd(const CgBlockInput* t, void* stream);
int cg_block_input_bwd(const CgBlockInput* t, void* stream);
int cg_block_input_supported(int B, int C, int T, int V);

/* ---- tail of the gate paths of a DSTD_GC block, conv_s / conv_t slots 5-7 and map_s / map_t, CISTGCN.py:337-352 / :378-384 ------
 * per path: z (B,C) -> BatchNorm2d -> Dropout -> PReLU -> cat with the block statistics (B,S) -> Linear (C, C+S) -> BatchNorm1d -> Dropout ->
 * PReLU -> Linear (C,C) = the gate (B,C).  A workgroup owns sixteen samples;
 * two launches forward, three backward, for both paths (cut only at the batch statistics, which every workgroup computes itself).  C <= 64,
 * S <= 192 (cg_gate_head_supported; else CG_ESHAPE: cg_norm_act_* +
 * cg_copy_many + cg_contract_many).  Forward writes y (B,C: the first Linear's output, kept for the backward), w, both bn.save;
 * backward needs `scratch` = cg_gate_head_scratch_floats(B, C, S) floats and `red` = two zeroed f64 words per path, dWl / dW2 ZERO on entry
 * (accumulated with float atomics), and writes dz, dstats (B,S) and every parameter gradient. */
typedef struct CgGatePath {
  const float* z;
  const float* stats; long long stats_ld;
  CgTailBN bn2; const float* alpha2;
  const float* Wl;
  CgTailBN bn3; const float* alpha3;
  const float* W2;
  unsigned int salt2, salt3;
  float* y;
  float* w;
  float* tap2; float* tap3;
  const float* dw;
  float* dz;
  float* dstats;
  float* dWl; float* dW2;
  float* dgamma2; float* dbeta2; float* dalpha2; float* dgamma3; float* dbeta3; float* dalpha3;
  float* scratch;
  double* red;
} CgGatePath;
typedef struct CgGateHead {
  int B, C, S, train, n, pad;
  float drop_p; int pad2; const unsigned long long* seed;
  CgGatePath p[2];
} CgGateHead;
int cg_gate_head_fwd(const CgGateHead* t, void* stream);
int cg_gate_head_bwd(const CgGateHead* t, void* stream);
int cg_gate_head_supported(int B, int C, int S);
long long cg_gate_head_scratch_floats(int B, int C, int S);

/* ---- ContextLayer heads 1 and 3, CISTGCN.py:408-418 with :465 / :467 ---------------------------------
 * Conv2d(1, C, 1, bias=False) -> BatchNorm2d(C) -> PReLU of the one-channel tensor x (B,1,T_out,3V), reduced over the positions:
 *   head 0 (context_conv1): y[0][b,c] = max_p z (first arg-max in `arg`), head 1 (context_conv3): y[1][b,c] = mean_p z.
 * The (B,C,P) activations are functions of x[b,p] and per-channel constants (batch statistics of w[c] x are w[c] mean(x),
 * w[c]^2 var(x)); they are never stored.  x (B,P) contiguous, C <= 64, P <= 16384 (else CG_ESHAPE: the caller composes the heads
 * from cg_pointwise_maps / cg_norm_act / cg_reduce_bc).  Train mode needs `xstats` = [CG_STAT_REPLICAS][1][2] f64 sums of x
 * (cg_chan_stats_many on the (B,1,P) view).  bn[h].save ([2][C]) and xsave ([2]) are written by the forward, read by the backward.
 * Backward: two launches (per-channel sums, then dx and the parameter gradients in closed form); `red`:
 * cg_context_heads_red_doubles(C) f64 words, zero on entry. */
typedef struct CgCtxHeads {
  int B, P, C, train;
  const float* x;
  const double* xstats;
  const float* w[2]; CgTailBN bn[2]; const float* alpha[2];
  float* y[2];
  int32_t* arg;
  float* xsave;
  float* tap[2];                /* optional (B,C,P): the PReLU outputs (diagnostics / branch records) */
  const float* dy[2];
  double* red;
  float* dx;
  float* dw[2]; float* dgamma[2]; float* dbeta[2]; float* dalpha[2];
} CgCtxHeads;
int cg_context_heads_fwd(const CgCtxHeads* t, void* stream);
int cg_context_heads_bwd(const CgCtxHeads* t, void* stream);
long long cg_context_heads_red_doubles(int C);

/* ---- evaluation harness counterpart, environment/test.py:97-132 ----------------------
 * y[r,k,:] = x[r,idx[k],:] : `inputs[:, :, dim_used]` (32 -> 22 joints); x (rows,Jin,3), y (rows,Jout,3) contiguous */
int cg_gather_joints(const float* x, float* y, const int32_t* idx, long long rows, int Jin, int Jout, void* stream);
/* out = target with the prediction scattered back (`mygt[:, :, dim_used] = outputs`, repeated joints copied, test.py:121-127)
 * and frame_err[t] = mean over samples and joints of |out - target| (losses.mpjpe, reduce_axis (0,2), losses.py:50-61).
 * pred (B,To,J22,3), target/out (B,To,J32,3); src[j] = prediction joint taken by skeleton joint j, or -1 (ground truth kept) */
int cg_eval_scatter_mpjpe(const float* pred, const float* target, float* out, float* frame_err, const int32_t* src,
                          int B, int To, int J32, int J22, void* stream);

/* ---- on-device input pipeline -----------------------------------------------------------------
 * The training augmentations of environment/custom_transforms.py (RandomFlip :243-298, RandomRotation :10-84, RandomScale
 * :87-161, RandomNoise :350-400, RandomTranslation :164-240, RandomPoseInvers :301-347; order of loaders/loader.py:42-130) and
 * the per-item tensors of loaders/h36m_motion_3d.py:94-108 for a whole batch in one launch.  raw (B,L,J,3); params (B,24) = per
 * sequence [flip x,y,z | rotate? | R 3x3 row-major (p' = (p-c) R + c) | scale x,y,z | translation rate x,y,z | noise amplitude
 * (0: off) | invert? | pad], drawn on the host in the reference's order; noise_tab (B,J,3) the U(-1,1) draws of RandomNoise or
 * NULL; perm (J) the joint permutation the pair swaps of RandomPoseInvers compose to, or NULL; outputs sample (B,input_n,J,3),
 * target (B,L-input_n,J,3), target_vel (same shape, cumulative frame differences from frame input_n-1 on), target_gvel
 * (B,L-input_n,J,1) cumulative speeds, processed (B,L,J,3) or NULL, sample_vel (B,input_n,J,3) = frame differences of the input
 * frames or NULL (the same dictionary comes out of loaders/amass_motion_3d.py:76-91). */
int cg_augment_sequences(const float* raw, const float* params, float* sample, float* target, float* target_vel,
                         float* target_gvel, float* processed, float* sample_vel, const float* noise_tab, const int32_t* perm,
                         int B, int L, int J, int input_n, void* stream);

/* ---- optimizer on the flat parameter buffer -----------------------------------
 * torch.optim.Adam semantics (environment/utils.py:53-57): L2 weight decay added to the gradient,
 * bias-corrected moments; optional clip_grad_value_ (environment/train.py:97-98) and gradient
 * pre-scale (1/world_size after the RCCL all-reduce).  step_count is the 1-based step index. */
int cg_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, float grad_scale, float clip_value, long long step_count,
                 void* stream);

/* Gather (direction 0) / scatter (direction 1) between the 698 per-tensor gradient buffers and the
 * flat fp32 buffer that RCCL all-reduces: ptrs[t] <-> flat + flat_off[t], work pre-chunked
 * on the host (chunk i = chunk_len[i] elements of tensor chunk_tensor[i] from element chunk_begin[i]). */
int cg_multi_copy(void* ptrs, const long long* flat_off, const int32_t* chunk_tensor, const int32_t* chunk_begin,
                  const int32_t* chunk_len, int n_chunks, float* flat, int direction, float scale, void* stream);
/* `scale` multiplies the gathered gradients (direction 0): the replica weight B_r * world / sum B of a data-parallel step
 * with unequal per-GPU batches (BASELINE configs[4]); passing a sub-range of the chunk arrays gathers one bucket.
 * p[i] *= s: the 1/world of the gradient mean when no optimizer kernel follows to absorb it (cg_adam_flat's grad_scale). */
int cg_scale(float* p, long long n, float s, void* stream);

#ifdef __cplusplus
}
#endif
#endif
