// Adam on the flat fp32 parameter buffer: one launch for all 698 tensors.
// torch.optim.Adam semantics as configured by environment/utils.py:53-57 (weight_decay is L2:
// added to the gradient; bias-corrected first/second moments; eps added after the sqrt), with the
// optional clip_grad_value_ of environment/train.py:97-98 and a gradient pre-scale that folds the
// 1/world_size of the data-parallel mean into the update.
#include "cg_common.h"
#include <math.h>

__global__ void cg_adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                    long long n, float lr, float b1, float b2, float eps, float wd, float gscale, float clip,
                                    float bc1, float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    const float pi = p[i];
    gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

extern "C" int cg_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float grad_scale, float clip_value, long long step_count,
                            void* stream_) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return CG_EARG;
  if (n <= 0 || step_count <= 0) return CG_ESHAPE;
  const float bc1 = 1.f - powf(beta1, (float)step_count);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step_count));
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cg_adam_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, param, grad, exp_avg, exp_avg_sq,
                     n, lr, beta1, beta2, eps, weight_decay, grad_scale, clip_value, bc1, bc2_sqrt);
  return cg_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Gather the per-tensor gradients (698 tensors) into one flat fp32 buffer (direction 0), or scatter
// the flat buffer back (direction 1), in ONE launch (the gather multiplies by `scale`: the replica weight B_r*world/sum B
// of a data-parallel step with unequal per-GPU batches): the flat buffer is what RCCL all-reduces and
// what cg_adam_flat consumes.  Work is pre-chunked on the host: chunk i copies `chunk_len[i]`
// elements of tensor `chunk_tensor[i]` starting at element `chunk_begin[i]`; tensor t lives at
// ptrs[t] and at flat + flat_off[t].
// ---------------------------------------------------------------------------------------------
__global__ void cg_multi_copy_kernel(float* const* __restrict__ ptrs, const long long* __restrict__ flat_off,
                                     const int32_t* __restrict__ chunk_tensor, const int32_t* __restrict__ chunk_begin,
                                     const int32_t* __restrict__ chunk_len, float* __restrict__ flat, int direction, float scale) {
  const int ch = blockIdx.x;
  const int t = chunk_tensor[ch];
  const long long b = chunk_begin[ch];
  const int n = chunk_len[ch];
  float* p = ptrs[t] + b;
  float* f = flat + flat_off[t] + b;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (direction == 0) f[i] = p[i] * scale; else p[i] = f[i];
  }
}

extern "C" int cg_multi_copy(void* ptrs, const long long* flat_off, const int32_t* chunk_tensor, const int32_t* chunk_begin,
                             const int32_t* chunk_len, int n_chunks, float* flat, int direction, float scale, void* stream_) {
  if (!ptrs || !flat_off || !chunk_tensor || !chunk_begin || !chunk_len || !flat) return CG_EARG;
  if (n_chunks <= 0) return CG_ESHAPE;
  hipLaunchKernelGGL(cg_multi_copy_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream_,
                     (float* const*)ptrs, flat_off, chunk_tensor, chunk_begin, chunk_len, flat, direction, scale);
  return cg_launch_status();
}

// p[i] *= s : the 1/world of a data-parallel gradient mean when no optimizer kernel follows to absorb it
__global__ void cg_scale_kernel(float* __restrict__ p, long long n, float s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] *= s;
}

extern "C" int cg_scale(float* p, long long n, float s, void* stream_) {
  if (!p) return CG_EARG;
  if (n <= 0) return CG_ESHAPE;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cg_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, p, n, s);
  return cg_launch_status();
}
