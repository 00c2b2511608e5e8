// Per-sample ops: occupancy-mask lookup (K8) and the density activation (K9, K10).
// Reference semantics: /root/reference/lib/cuda/render_utils_kernel.cu:300-428.
#include "common.h"

// K8 :310-318.  i = (int)roundf(x*scale + shift) (contracted; half away from zero).
__global__ void __launch_bounds__(DVGO_BLOCK)
maskcache_lookup_kernel(const uint8_t* __restrict__ world, const float* __restrict__ xyz,
                        const float* __restrict__ scale, const float* __restrict__ shift,
                        int sz_i, int sz_j, int sz_k, int64_t n_pts, uint8_t* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  const int i = (int)roundf(fmaf(xyz[3 * p + 0], scale[0], shift[0]));
  const int j = (int)roundf(fmaf(xyz[3 * p + 1], scale[1], shift[1]));
  const int k = (int)roundf(fmaf(xyz[3 * p + 2], scale[2], shift[2]));
  uint8_t v = 0;
  if (0 <= i && i < sz_i && 0 <= j && j < sz_j && 0 <= k && k < sz_k)
    v = world[(int64_t)i * sz_j * sz_k + (int64_t)j * sz_k + k];
  out[p] = v;
}

// K9 :364-369.  Streaming: 4 B in, 8 B out per sample; 4 samples per lane (16-B accesses)
// in the body, scalar tail.
__device__ __forceinline__ void raw2alpha_one(float d, float shift, float interval, float& e, float& a) {
  e = expf(d + shift);                 // may be +inf
  a = 1.0f - powf(1.0f + e, -interval);
}

__global__ void __launch_bounds__(DVGO_BLOCK)
raw2alpha_kernel(const float* __restrict__ density, float shift, float interval, int64_t n,
                 float* __restrict__ exp_d, float* __restrict__ alpha) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t * 4;
  if (i + 3 < n) {
    const float4 d = *reinterpret_cast<const float4*>(density + i);
    float4 e, a;
    raw2alpha_one(d.x, shift, interval, e.x, a.x);
    raw2alpha_one(d.y, shift, interval, e.y, a.y);
    raw2alpha_one(d.z, shift, interval, e.z, a.z);
    raw2alpha_one(d.w, shift, interval, e.w, a.w);
    *reinterpret_cast<float4*>(exp_d + i) = e;
    *reinterpret_cast<float4*>(alpha + i) = a;
  } else {
    for (int64_t k = i; k < n; ++k) {
      float e, a;
      raw2alpha_one(density[k], shift, interval, e, a);
      exp_d[k] = e;
      alpha[k] = a;
    }
  }
}

// K10 :402-405.  (float)( min((double)e, 1e10) * (double)powf(1+e, -interval-1) * interval * g )
__device__ __forceinline__ float raw2alpha_bwd_one(float e, float g, float interval) {
  double v = fmin((double)e, 1e10) * (double)powf(1.0f + e, -interval - 1.0f);
  v = v * (double)interval;
  v = v * (double)g;
  return (float)v;
}

__global__ void __launch_bounds__(DVGO_BLOCK)
raw2alpha_backward_kernel(const float* __restrict__ exp_d, const float* __restrict__ grad_back,
                          float interval, int64_t n, float* __restrict__ grad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t * 4;
  if (i + 3 < n) {
    const float4 e = *reinterpret_cast<const float4*>(exp_d + i);
    const float4 g = *reinterpret_cast<const float4*>(grad_back + i);
    float4 o;
    o.x = raw2alpha_bwd_one(e.x, g.x, interval);
    o.y = raw2alpha_bwd_one(e.y, g.y, interval);
    o.z = raw2alpha_bwd_one(e.z, g.z, interval);
    o.w = raw2alpha_bwd_one(e.w, g.w, interval);
    *reinterpret_cast<float4*>(grad + i) = o;
  } else {
    for (int64_t k = i; k < n; ++k) grad[k] = raw2alpha_bwd_one(exp_d[k], grad_back[k], interval);
  }
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// scalar fallbacks for buffers that are not 16-byte aligned (e.g. sliced tensors)
__global__ void __launch_bounds__(DVGO_BLOCK)
raw2alpha_scalar_kernel(const float* __restrict__ density, float shift, float interval, int64_t n,
                        float* __restrict__ exp_d, float* __restrict__ alpha) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  float e, a;
  raw2alpha_one(density[k], shift, interval, e, a);
  exp_d[k] = e;
  alpha[k] = a;
}
__global__ void __launch_bounds__(DVGO_BLOCK)
raw2alpha_backward_scalar_kernel(const float* __restrict__ exp_d, const float* __restrict__ grad_back,
                                 float interval, int64_t n, float* __restrict__ grad) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  grad[k] = raw2alpha_bwd_one(exp_d[k], grad_back[k], interval);
}

extern "C" {

int dvgo_maskcache_lookup(const uint8_t* world, const float* xyz, const float* xyz2ijk_scale,
                          const float* xyz2ijk_shift, int sz_i, int sz_j, int sz_k, int64_t n_pts,
                          uint8_t* out, void* stream) {
  if (n_pts < 0 || sz_i < 0 || sz_j < 0 || sz_k < 0) return DVGO_EINVAL;
  if (n_pts == 0) return 0;
  if (!world || !xyz || !xyz2ijk_scale || !xyz2ijk_shift || !out) return DVGO_EINVAL;
  if (!dvgo_fits(n_pts)) return DVGO_ERANGE;
  maskcache_lookup_kernel<<<dvgo_blocks(n_pts, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      world, xyz, xyz2ijk_scale, xyz2ijk_shift, sz_i, sz_j, sz_k, n_pts, out);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_raw2alpha(const float* density, float shift, float interval, int64_t n_pts, float* exp_d,
                   float* alpha, void* stream) {
  if (n_pts < 0) return DVGO_EINVAL;
  if (n_pts == 0) return 0;
  if (!density || !exp_d || !alpha) return DVGO_EINVAL;
  if (!dvgo_fits(n_pts)) return DVGO_ERANGE;
  if (aligned16(density) && aligned16(exp_d) && aligned16(alpha)) {
    raw2alpha_kernel<<<dvgo_blocks((n_pts + 3) / 4, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
        density, shift, interval, n_pts, exp_d, alpha);
  } else {
    raw2alpha_scalar_kernel<<<dvgo_blocks(n_pts, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
        density, shift, interval, n_pts, exp_d, alpha);
  }
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_raw2alpha_backward(const float* exp_d, const float* grad_back, float interval,
                            int64_t n_pts, float* grad, void* stream) {
  if (n_pts < 0) return DVGO_EINVAL;
  if (n_pts == 0) return 0;
  if (!exp_d || !grad_back || !grad) return DVGO_EINVAL;
  if (!dvgo_fits(n_pts)) return DVGO_ERANGE;
  if (aligned16(exp_d) && aligned16(grad_back) && aligned16(grad)) {
    raw2alpha_backward_kernel<<<dvgo_blocks((n_pts + 3) / 4, DVGO_BLOCK), DVGO_BLOCK, 0,
                                (hipStream_t)stream>>>(exp_d, grad_back, interval, n_pts, grad);
  } else {
    raw2alpha_backward_scalar_kernel<<<dvgo_blocks(n_pts, DVGO_BLOCK), DVGO_BLOCK, 0,
                                       (hipStream_t)stream>>>(exp_d, grad_back, interval, n_pts, grad);
  }
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
