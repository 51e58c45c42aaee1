// On-device input pipeline: the training augmentations of the reference
// (environment/custom_transforms.py: RandomFlip :243-298, RandomRotation :10-84, RandomScale :87-161, RandomNoise :350-400,
// RandomTranslation :164-240, RandomPoseInvers :301-347, composed in the order of loaders/loader.py:42-130) and the per-item tensors of
// loaders/h36m_motion_3d.py:94-108 (sample / target split, velocities, cumulative target velocities and speeds), for a
// whole batch in one launch: one workgroup per sequence, the sequence lives in LDS between the steps.
// The random draws are made on the host in the reference's order (environment/input_pipeline.py) and arrive as one
// parameter row per sequence; everything that depends on the data (centroids, per-axis extent) is computed here.
#include "cg_common.h"

#define CG_AUG_NPAR 24      // flip x,y,z | rot_on | R[3][3] (row vector times matrix) | scale x,y,z | translation rate x,y,z | noise | invert | pad

HIP_DYNAMIC_SHARED(unsigned char, cg_dyn_lds)

// mean over all points of the sequence, per axis (f64 accumulation, result valid in every thread)
__device__ __forceinline__ void cg_aug_centroid(const float* s, int n_pts, double* red, float c[3]) {
  double a[3] = {0.0, 0.0, 0.0};
  for (int p = threadIdx.x; p < n_pts; p += blockDim.x) { a[0] += s[3 * p]; a[1] += s[3 * p + 1]; a[2] += s[3 * p + 2]; }
  __shared__ double out[3];
  for (int k = 0; k < 3; ++k) {
    const double t = cg_block_sum(a[k], red);
    if (threadIdx.x == 0) out[k] = t / (double)n_pts;
  }
  __syncthreads();
  c[0] = (float)out[0]; c[1] = (float)out[1]; c[2] = (float)out[2];
  __syncthreads();
}

__global__ void cg_augment_sequences_kernel(const float* __restrict__ raw, const float* __restrict__ params, float* __restrict__ sample,
                                            float* __restrict__ target, float* __restrict__ target_vel, float* __restrict__ target_gvel,
                                            float* __restrict__ processed, float* __restrict__ sample_vel,
                                            const float* __restrict__ noise_tab, const int* __restrict__ perm, int L, int J, int input_n) {
  float* s = reinterpret_cast<float*>(cg_dyn_lds);
  __shared__ double red[16];
  __shared__ float ext[6];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int n_pts = L * J, n = 3 * n_pts;
  const float* par = params + (long long)b * CG_AUG_NPAR;
  const float* src = raw + (long long)b * n;
  for (int e = tid; e < n; e += nt) s[e] = src[e];
  __syncthreads();
  float c[3];
  // RandomFlip: mirror about the centroid of the incoming sequence (one centroid for all axes)
  if (par[0] != 0.f || par[1] != 0.f || par[2] != 0.f) {
    cg_aug_centroid(s, n_pts, red, c);
    for (int e = tid; e < n; e += nt) {
      const int a = e % 3;
      if (par[a] != 0.f) s[e] = c[a] - (s[e] - c[a]);
    }
    __syncthreads();
  }
  // RandomRotation: (p - centroid) R + centroid, R from the rotation vector (host)
  if (par[3] != 0.f) {
    cg_aug_centroid(s, n_pts, red, c);
    for (int p = tid; p < n_pts; p += nt) {
      const float x = s[3 * p] - c[0], y = s[3 * p + 1] - c[1], z = s[3 * p + 2] - c[2];
      s[3 * p] = (x * par[4] + y * par[7] + z * par[10]) + c[0];
      s[3 * p + 1] = (x * par[5] + y * par[8] + z * par[11]) + c[1];
      s[3 * p + 2] = (x * par[6] + y * par[9] + z * par[12]) + c[2];
    }
    __syncthreads();
  }
  // RandomScale (about the origin, as the reference)
  for (int e = tid; e < n; e += nt) s[e] *= par[13 + e % 3];
  __syncthreads();
  // per-axis extent (max - min over the whole sequence) of the current data -> ext[0..2]
  auto extent = [&]() {
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (int p = tid; p < n_pts; p += nt)
      for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], s[3 * p + a]); mx[a] = fmaxf(mx[a], s[3 * p + a]); }
    float* part = reinterpret_cast<float*>(red);            // one slot per wave: 4 waves x 6 values fit the 16 doubles
    for (int a = 0; a < 3; ++a)
      for (int off = 32; off > 0; off >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64)); }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) for (int a = 0; a < 3; ++a) { part[wave * 6 + a] = mn[a]; part[wave * 6 + 3 + a] = mx[a]; }
    __syncthreads();
    if (tid < 3) {
      float lo = part[tid], hi = part[3 + tid];
      for (int w = 1; w < (nt >> 6); ++w) { lo = fminf(lo, part[w * 6 + tid]); hi = fmaxf(hi, part[w * 6 + 3 + tid]); }
      ext[tid] = hi - lo;
    }
    __syncthreads();
  };
  // RandomNoise (custom_transforms.py:368-396, whole sequence, constant amplitude): data += noise * u[joint][axis] * extent[axis],
  // u ~ U(-1, 1) drawn per (joint, axis) on the host
  if (par[19] != 0.f && noise_tab) {
    extent();
    const float* u = noise_tab + (long long)b * J * 3;
    for (int e = tid; e < n; e += nt) { const int a = e % 3, j = (e / 3) % J; s[e] += par[19] * u[3 * j + a] * ext[a]; }
    __syncthreads();
  }
  // RandomTranslation: rate x per-axis extent of the (scaled) sequence
  if (par[16] != 0.f || par[17] != 0.f || par[18] != 0.f) {
    extent();
    for (int e = tid; e < n; e += nt) { const int a = e % 3; s[e] += par[16 + a] * ext[a]; }
    __syncthreads();
  }
  // RandomPoseInvers (custom_transforms.py:323-344, whole sequence): left and right joints trade places.  The pair swaps of the
  // reference compose to a joint permutation (host): output joint j shows joint perm[j]; applied while the results are written
  const bool inv = par[20] != 0.f && perm;
  auto at = [&](int e) {                                  // element e of the (L, J, 3) result
    if (!inv) return s[e];
    const int a = e % 3, p = e / 3, j = p % J;
    return s[(p - j + perm[j]) * 3 + a];
  };
  // outputs of H36m_Motion3D.__getitem__ (h36m_motion_3d.py:94-108)
  const int n_in = input_n * J * 3, n_out = (L - input_n) * J * 3;
  if (processed) for (int e = tid; e < n; e += nt) processed[(long long)b * n + e] = at(e);
  for (int e = tid; e < n_in; e += nt) sample[(long long)b * n_in + e] = at(e);
  for (int e = tid; e < n_out; e += nt) target[(long long)b * n_out + e] = at(n_in + e);
  if (sample_vel)                                          // "sample_vel": velocities[:input_n] (h36m_motion_3d.py:101)
    for (int e = tid; e < n_in; e += nt) sample_vel[(long long)b * n_in + e] = at(e + 3 * J) - at(e);
  // velocities[t] = s[t+1] - s[t]; target_vel = cumsum_t velocities[input_n-1:], target_gvel = cumsum_t |velocities[input_n-1:]|
  const int To = L - input_n;
  for (int j = tid; j < J; j += nt) {
    float acc[3] = {0.f, 0.f, 0.f}, gacc = 0.f;
    for (int k = 0; k < To; ++k) {
      const int t = input_n - 1 + k;
      float v[3];
      for (int a = 0; a < 3; ++a) { v[a] = at(((t + 1) * J + j) * 3 + a) - at((t * J + j) * 3 + a); acc[a] += v[a]; }
      gacc += sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      float* tv = target_vel + (((long long)b * To + k) * J + j) * 3;
      tv[0] = acc[0]; tv[1] = acc[1]; tv[2] = acc[2];
      target_gvel[((long long)b * To + k) * J + j] = gacc;
    }
  }
}

// include/cistgcn_hip.h : cg_augment_sequences
extern "C" int cg_augment_sequences(const float* raw, const float* params, float* sample, float* target, float* target_vel,
                                    float* target_gvel, float* processed, float* sample_vel, const float* noise_tab, const int* perm,
                                    int B, int L, int J, int input_n, void* stream_) {
  if (!raw || !params || !sample || !target || !target_vel || !target_gvel) return CG_EARG;
  if (B <= 0 || L <= 1 || J <= 0 || input_n <= 0 || input_n >= L) return CG_ESHAPE;
  const size_t lds = (size_t)L * J * 3 * sizeof(float);
  if (lds > 150 * 1024) return CG_ESHAPE;
  if (lds > 48 * 1024) {
    hipError_t e = cg_lds_limit((const void*)cg_augment_sequences_kernel, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(cg_augment_sequences_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream_, raw, params, sample, target,
                     target_vel, target_gvel, processed, sample_vel, noise_tab, perm, L, J, input_n);
  return cg_launch_status();
}
