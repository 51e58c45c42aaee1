// Shared between the two translation units of the fused ST-GCN stage (stgcn_domain.hip: VALU kernels for narrow layers
// and the C ABI; stgcn_domain_mfma.hip: matrix-core kernels for wide layers).
#pragma once
#include "cg_common.h"

// geometry of the matrix-core kernels (see the header comment of stgcn_domain_mfma.hip)
struct CgDomM {
  int B, Cin, Cout, T, V;
  int NG, J;                 // groups per sample (joints | frames), contraction length (frames | joints)
  int GT, ntiles;            // groups per tile, tiles per sample
  int total, per;            // tiles, tiles per workgroup (contiguous range)
  int Js, RS;                // column stride of a group inside a row of the [channel][position] images, row stride (== 4 mod 8)
  int Jr, Jsa;               // adjacency slab: rows per group (J up to 16), row stride (Jr + 4)
  int CiM, CoM, WS;          // channels up to 16, row stride of the weight image
  int dbg, w_global;         // (unused) | 1: the backward's dG product reads W from HBM/L2 instead of an LDS image
  unsigned magicJ, magicGT, magicJs, magicRun, magicNP, magicMTi, magicNT, magicNPp;  // ceil(2^32 / d) for the index divisions by J, GT and Js (0: d = 1)
};

int cg_domm_geom(CgDomM& g, int B, int Cin, int Cout, int T, int V, int domain, bool bwd);
size_t cg_domm_lds_bytes(const CgDomM& g, bool bwd);
int cg_domm_bwd_launch(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj, float* ws,
                       int replicas, int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream);
int cg_domm_fwd_launch(const float* x, const float* adj, const float* W, const float* bias, float* y, double* ystats,
                       int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream);

// geometry of the plane kernels (stgcn_domain_planes.hip)
struct CgDomP {
  int B, Cin, Cout, T, V, TV;
  int NG, J;                 // groups per sample (joints | frames), contraction length (frames | joints)
  int JS, GSTR;              // LDS image sZ[group][16][JS]: row stride (== 2 mod 4), group stride
  int NOC, KS, WS;           // chunks of 16 output channels, MFMA steps over the input channels, row stride of the weight image
  int JSTEPS, zfloats;       // MFMA steps over the contraction axis, floats of sZ
  int VW, VWB, NL;           // floats per lane of a plane access | of an adjacency row access, adjacency accesses per step
  unsigned magicV, magicNQ;  // ceil(2^32 / d) for the divisions by V and by T*V / VW
  // backward of the space domain
  int NTC, TC;               // chunks of frames per sample, frames per chunk (<= 16)
  int CinR, XS;              // rows / row stride of the x piece sX[CinR][XS] (and of sdZ[16][XS])
  int GZ, GY, YS;            // sZ[V][16][16] group stride; sY[V][16][YS] group / row stride
  int zfl, yfl, dzfl, bwd_floats;  // floats of sZ, sY, sdZ (= second dY image), the whole LDS image
  int VWP, VWY, VWA;         // floats per lane: x piece / dx rows, dY pieces, slab rows
};
int cg_domp_geom(CgDomP& g, int B, int Cin, int Cout, int T, int V, int domain);
int cg_domp_bwd_launch(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj, float* ws,
                       int replicas, int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream);
int cg_domp_bwd_time_launch(const float* x, const float* adj, const float* W, const float* dy, float* dx, float* dadj, float* ws,
                            int replicas, int B, int Cin, int Cout, int T, int V, hipStream_t stream);
int cg_domp_fwd_launch(const float* x, const float* adj, const float* W, const float* bias, float* y, double* ystats,
                       int B, int Cin, int Cout, int T, int V, int domain, hipStream_t stream);
